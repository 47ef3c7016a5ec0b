"""Helpers for the -m gpu tests: device buffers via torch (plumbing only) + librxgpu calls."""
import ctypes as C

import numpy as np

import rx_tools_amd as R
from support import oracle, oracle_fm_state, ptr16, ptr32, ptr64, FmState, PowerCfg  # noqa: F401


def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "no GPU visible"
    return torch


def to_dev(a):
    torch = torch_cuda()
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def carry_from_oracle_state(st):
    c = R.FmCarry()
    for f in ("now_r", "now_j", "prev_index", "pre_r", "pre_j", "deemph_avg", "now_lpr", "prev_lpr_index", "squelch_hits", "dc_avg", "dc_avgI", "dc_avgQ"):
        setattr(c, f, getattr(st, f))
    C.memmove(C.addressof(c.lp_i_hist), C.addressof(st.lp_i_hist), C.sizeof(c.lp_i_hist))
    C.memmove(C.addressof(c.lp_q_hist), C.addressof(st.lp_q_hist), C.sizeof(c.lp_q_hist))
    C.memmove(C.addressof(c.droop_i_hist), C.addressof(st.droop_i_hist), C.sizeof(c.droop_i_hist))
    C.memmove(C.addressof(c.droop_q_hist), C.addressof(st.droop_q_hist), C.sizeof(c.droop_q_hist))
    return c


def carry_tuple(c):
    return (c.now_r, c.now_j, c.prev_index, c.pre_r, c.pre_j, c.deemph_avg, c.now_lpr, c.prev_lpr_index,
            bytes(c.lp_i_hist), bytes(c.lp_q_hist), bytes(c.droop_i_hist), bytes(c.droop_q_hist))


def gpu_fm_stream(iq, block_len, n_runs=1, carry=None, pipelined=False, **params):
    """Run iq (int16 numpy) through rxgpu_fm_stream_run in n_runs consecutive calls.
    Returns (out int16, per-block lens, final carry)."""
    torch = torch_cuda()
    p = R.FmParams.wbfm()
    for k, v in params.items():
        setattr(p, k, v)
    n_blocks = len(iq) // block_len
    per = (n_blocks + n_runs - 1) // n_runs
    s = R.FmStream(p, per, block_len)
    if carry is None and "squelch_hits" in params:
        carry = R.FmCarry()
    if carry is not None:
        s.set_carry(carry)
    d_iq = to_dev(iq)
    d_out = torch.zeros(len(iq) // 2 + 64, dtype=torch.int16, device="cuda")
    outs, lens = [], []
    b = 0
    pos = 0
    while b < n_blocks:
        nb = min(per, n_blocks - b)
        if pipelined:
            # enqueue everything back to back (run r+1's decimator overlaps run r's audio stages), then wait once
            n, bl = s.run_async(d_iq.data_ptr() + b * block_len * 2, nb, block_len, d_out.data_ptr() + 2 * pos,
                                d_out.numel() - pos, True)
            pos += n
        else:
            n, bl = s.run(d_iq.data_ptr() + b * block_len * 2, nb, block_len, d_out.data_ptr(), d_out.numel(), True)
            outs.append(d_out[:n].cpu().numpy().copy())
        lens += bl
        b += nb
    if pipelined:
        s.wait()
        outs.append(d_out[:pos].cpu().numpy().copy())
    c = s.get_carry()
    fix = s.host_fixups
    s.close()
    return np.concatenate(outs) if outs else np.zeros(0, np.int16), np.array(lens, dtype=np.int32), c, fix
