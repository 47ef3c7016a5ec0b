"""rx_sdr output conversions on the MI355X against the oracle and the golden vectors (every int16 value),
through the C ABI (rxgpu_sdr_convert / rxgpu_sdr_convert_host)."""
import os

import numpy as np
import pytest

import rx_tools_amd as R
import support
from gpu_support import to_dev, torch_cuda

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "sdr_cases.npz"))
ALL16 = np.arange(-32768, 32768, dtype=np.int16)


def _dev(fmt, data):
    out = R.sdr_convert(fmt, to_dev(data))
    R.check(R.lib().rxgpu_sync())
    return out.cpu().numpy()


@pytest.mark.parametrize("fmt", ["CU8", "CS8", "CF32"])
def test_every_int16_matches_reference(fmt):
    got = _dev(fmt, ALL16)
    want = GOLD["all_" + fmt]
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


def test_cs12_golden():
    assert np.array_equal(_dev("CS16", GOLD["cs12_in"][: 3 * 4096]), GOLD["cs12_out"][: 2 * 4096])
    assert np.array_equal(R.sdr_convert_host("CS16", GOLD["cs12_in"]), GOLD["cs12_out"])


@pytest.mark.parametrize("fmt", ["CU8", "CS8", "CF32", "CS16"])
@pytest.mark.parametrize("n_elems", [0, 1, 7, 8, 9, 4099, 131072, 1 << 20 | 5])
def test_ragged_sizes_vs_oracle(fmt, n_elems):
    rng = np.random.default_rng(n_elems + 17)
    if fmt == "CS16":
        data = rng.integers(0, 256, size=3 * n_elems, dtype=np.uint8)
    else:
        data = rng.integers(-32768, 32768, size=2 * n_elems, dtype=np.int16)
    want = support.oracle_sdr_convert(fmt, data)
    got_host = R.sdr_convert_host(fmt, data)
    assert np.array_equal(got_host.view(np.uint8), want.view(np.uint8))
    if n_elems:
        got = _dev(fmt, data)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


def test_full_size_properties():
    """2^28 elements (1 GiB of CS16): each output is a pure function of its input, so the converted stream must
    equal the 65536-entry table of the exhaustive test gathered by the input values."""
    torch = torch_cuda()
    n16 = 1 << 29
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randint(-32768, 32768, (n16,), dtype=torch.int16, device="cuda", generator=g)
    torch.cuda.synchronize()           # torch fills on its own stream, librxgpu launches on rxgpu_stream()
    for fmt in ("CU8", "CS8", "CF32"):
        out = R.sdr_convert(fmt, x)
        R.check(R.lib().rxgpu_sync())
        table = torch.from_numpy(GOLD["all_" + fmt].copy()).cuda()
        want = table[(x.to(torch.int64) + 32768)]
        assert torch.equal(out.view(torch.uint8), want.view(torch.uint8))
        del out, want


def test_bad_arguments():
    L = R.lib()
    assert L.rxgpu_sdr_convert(9, 16, 1, 16) == -2
    assert L.rxgpu_sdr_convert_host(-1, 16, 1, 16) == -2
    assert L.rxgpu_sdr_out_bytes(0, 10) == 20 and L.rxgpu_sdr_out_bytes(2, 10) == 80 and L.rxgpu_sdr_in_bytes(3, 10) == 30
    x = to_dev(np.zeros(64, dtype=np.int16))
    assert L.rxgpu_sdr_convert(0, x.data_ptr() + 2, 4, x.data_ptr()) == -2     # misaligned
