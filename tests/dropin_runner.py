"""Runs the reference's own rx_fm main() (oracle/_ref/libref_fm.so = rtl_fm.c compiled unmodified)
on a synthetic cs16 capture served by the fake SoapySDR device, untouched: the reference's threads call the reference's
full_demod / scanner.  (The GPU side of the end-to-end tests runs the product's own executables, dropin/Makefile.)
Usage: python dropin_runner.py cpu <iq.npy> <out.raw> [rx_fm args...]
       python dropin_runner.py power-cpu <iq.npy> <out.csv> [rx_power args...]
       python dropin_runner.py plan <out.json> [rx_fm args...]
(plan: the reference's main() parses the flags and derives its parameters itself -- getopt, `rate_in *= post_downsample`,
optimal_settings via the controller thread, deemph_a -- on a one-block capture; when the fake device runs dry the
globals demod/dongle are dumped as JSON.  Pins rxgpu_fm_params_init / rxgpu_fm_plan_settings.)
(rx_power: the capture holds P passes x tunes x buf_len int16; once it is exhausted scanner() adds nothing
more, so the first report after -i 1 holds exactly P passes.)
Separate process per run: the reference keeps its exit flag in a file-static."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref")


def main_power(mode, iq, out_path, extra):
    ref = C.CDLL(os.path.join(REF, "libref_power.so"), mode=C.RTLD_GLOBAL)
    ref.soapy_fake_set_source.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    # scanner() asks for buf_len complex elements but consumes buf_len int16: hand out buf_len/2 per read
    buf_len = int(extra[extra.index("--buf-len") + 1])
    del extra[extra.index("--buf-len"):extra.index("--buf-len") + 2]
    ref.soapy_fake_set_source(iq.ctypes.data, len(iq) // 2, buf_len // 2)
    args = [b"rx_power"] + [a.encode() for a in extra] + [out_path.encode()]
    argv = (C.c_char_p * (len(args) + 1))(*args, None)
    rc = ref.rx_power_main(len(args), argv)
    os._exit(rc)


def main_plan(out_json, extra):
    import json
    import signal
    import time
    from rx_tools_amd.structs import DemodState, DongleState
    ref = C.CDLL(os.path.join(REF, "libref_fm.so"), mode=C.RTLD_GLOBAL)
    ref.soapy_fake_set_source.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    ref.ref_fm_demod.restype = C.c_void_p
    ref.ref_fm_dongle.restype = C.c_void_p
    iq = np.zeros(2 * 131072, np.int16)
    ref.soapy_fake_set_source(iq.ctypes.data, len(iq) // 2, 0)
    PACE = C.CFUNCTYPE(None)

    def eos():
        time.sleep(0.2)                    # let the controller thread finish optimal_settings (rtl_fm.c:1012-1020)
        d = DemodState.from_address(ref.ref_fm_demod())
        g = DongleState.from_address(ref.ref_fm_dongle())
        ref.ref_fm_fn.restype = C.c_void_p
        fn = {ref.ref_fm_fn(i): i for i in range(5)}
        row = {k: int(getattr(d, k)) for k in ("rate_in", "rate_out", "rate_out2", "downsample", "post_downsample", "output_scale",
                                                "downsample_passes", "comp_fir_size", "custom_atan", "deemph", "deemph_a",
                                                "squelch_level", "dc_block_audio", "dc_block_raw", "adc_block_const", "rdc_block_const")}
        row["mode_fn"] = fn.get(d.mode_demod, -1)      # ref_fm_fn numbering: 0 fm, 1 raw, 2 am, 3 usb, 4 lsb
        row["capture_freq"], row["capture_rate"], row["offset_tuning"] = int(g.freq), int(g.rate), int(g.offset_tuning)
        with open(out_json, "w") as f:
            json.dump(row, f)
        os.kill(os.getpid(), signal.SIGINT)
    keep = (PACE(lambda: time.sleep(0.01)), PACE(eos))
    ref.soapy_fake_set_pace_hook(keep[0])
    ref.soapy_fake_set_eos_hook(keep[1])
    args = [b"rx_fm"] + [a.encode() for a in extra] + [b"/dev/null"]
    argv = (C.c_char_p * (len(args) + 1))(*args, None)
    rc = ref.rx_fm_main(len(args), argv)
    os._exit(rc)


def main():
    if sys.argv[1] == "plan":
        return main_plan(sys.argv[2], sys.argv[3:])
    mode, iq_path, out_path = sys.argv[1:4]
    extra = sys.argv[4:]
    iq = np.load(iq_path)
    if mode.startswith("power"):
        return main_power(mode, iq, out_path, extra)
    ref = C.CDLL(os.path.join(REF, "libref_fm.so"), mode=C.RTLD_GLOBAL)
    ref.soapy_fake_set_source.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    ref.soapy_fake_set_source(iq.ctypes.data, len(iq) // 2, 0)
    # pacing and shutdown from Python callbacks
    import signal
    import time
    PACE = C.CFUNCTYPE(None)

    pace_s = float(os.environ.get("DROPIN_PACE", "0.02"))

    def pace():
        # wait until the demod thread has consumed the previous block: a fixed sleep well above one block's CPU time
        # (the reference's hand-off is a single lossy slot, rtl_fm.c:858-862: on a loaded host the caller raises $DROPIN_PACE)
        time.sleep(pace_s)

    def eos():
        time.sleep(0.1)
        os.kill(os.getpid(), signal.SIGINT)
    keep = (PACE(pace), PACE(eos))
    ref.soapy_fake_set_pace_hook(keep[0])
    ref.soapy_fake_set_eos_hook(keep[1])
    args = [b"rx_fm"] + [a.encode() for a in extra] + [out_path.encode()]
    argv = (C.c_char_p * (len(args) + 1))(*args, None)
    rc = ref.rx_fm_main(len(args), argv)
    os._exit(rc)


if __name__ == "__main__":
    main()
