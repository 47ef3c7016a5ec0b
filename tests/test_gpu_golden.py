"""-m gpu: the HIP path against the golden vectors produced by the reference's own code
(tests/golden, generated in the build container by oracle/gen_golden.py)."""
import os
import sys

import numpy as np
import pytest

import rx_tools_amd as R
from support import GOLDEN_DIR

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))


def test_fm_golden():
    from gen_golden import FM_CASES
    from gpu_support import gpu_fm_stream
    z = np.load(os.path.join(GOLDEN_DIR, "fm_cases.npz"))
    for name, kind, n_blocks, block_len, params in FM_CASES:
        got, lens, c, _ = gpu_fm_stream(z[name + "__iq"], block_len, **params)
        assert np.array_equal(got, z[name + "__out"]), name
        assert np.array_equal(lens, z[name + "__lens"]), name
        want = z[name + "__carry"]
        assert [c.now_r, c.now_j, c.prev_index, c.pre_r, c.pre_j, c.now_lpr, c.prev_lpr_index] == list(want[:7]), name
        hist = np.concatenate([np.ctypeslib.as_array(c.lp_i_hist).ravel(), np.ctypeslib.as_array(c.lp_q_hist).ravel(),
                               np.ctypeslib.as_array(c.droop_i_hist), np.ctypeslib.as_array(c.droop_q_hist)])
        p = params.get("downsample_passes", 0)
        gh = z[name + "__hist"]
        assert np.array_equal(hist[:6 * p], gh[:6 * p]) and np.array_equal(hist[60:60 + 6 * p], gh[60:60 + 6 * p]), name
        if params.get("comp_fir_size") == 9:
            assert np.array_equal(hist[120:], gh[120:]), name


def test_power_golden():
    from gen_golden import POWER_CASES
    from test_gpu_power import gpu_scan
    z = np.load(os.path.join(GOLDEN_DIR, "power_cases.npz"))
    for name, rng, crop, window, flags, amp, passes, max_tunes in POWER_CASES:
        plan = R.plan_range(rng, crop, flags[0])
        n = 1 << plan.bin_e
        data = z[name + "__in"]
        tunes = data.shape[1]
        wc = R.window_coefs(window, n)
        assert np.array_equal(wc, z[name + "__window"])
        avg, samples = gpu_scan(np.ascontiguousarray(data).ravel(), passes, tunes, plan, wc, R.sine_table(plan.bin_e), *flags)
        assert np.array_equal(avg, z[name + "__avg"]), name
        assert np.array_equal(samples, z[name + "__samples"]), name
