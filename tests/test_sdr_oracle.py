"""rx_sdr output formats + WAV header: the CPU restatement against golden vectors made by the reference's own
main() (oracle/gen_golden.py: sdr_cases), and against the reference itself when it is present."""
import ctypes as C
import os

import numpy as np
import pytest

import support

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "sdr_cases.npz"))
ALL16 = np.arange(-32768, 32768, dtype=np.int16)


@pytest.mark.parametrize("fmt", ["CU8", "CS8", "CF32"])
def test_oracle_every_int16(fmt):
    got = support.oracle_sdr_convert(fmt, ALL16)
    want = GOLD["all_" + fmt]
    assert got.dtype == want.dtype
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))     # bit-exact, CF32 included


def test_oracle_cs12():
    assert np.array_equal(support.oracle_sdr_convert("CS16", GOLD["cs12_in"]), GOLD["cs12_out"])


def test_known_answers():
    # spot values written out so the fixture itself is pinned: rtl_sdr.c:368-386
    cs8, cu8, cf = GOLD["all_CS8"], GOLD["all_CU8"], GOLD["all_CF32"]
    at = lambda a, x: a[x + 32768]
    assert [int(at(cs8, x)) for x in (-32768, -103, -102, 0, 153, 154, 32664, 32665, 32767)] == [-127, 0, 0, 0, 0, 1, 127, -128, -128]
    assert [int(at(cu8, x)) for x in (-32768, -32767, -103, -102, 0, 153, 154, 32767)] == [0, 0, 126, 127, 127, 127, 128, 255]
    assert float(at(cf, 32767)) == 1.0 and float(at(cf, -32767)) == -1.0 and float(at(cf, 0)) == 0.0
    assert at(cf, -32768) == np.float32(-32768.0) / np.float32(32767.0)


def test_oracle_wav_header():
    L = support.oracle()
    for (rate, raw), want in zip(GOLD["wav_args"], GOLD["wav_headers"]):
        buf = (C.c_uint8 * 44)()
        L.rxo_wav_header(int(rate), int(raw), buf)
        assert bytes(buf) == want.tobytes()
    assert GOLD["wav_headers"][0][:4].tobytes() == b"RIFF" and GOLD["wav_headers"][0][36:40].tobytes() == b"data"


def test_product_wav_header_is_host_only():
    # rxgpu_wav_header needs no device: same bytes as the reference's generate_header
    import rx_tools_amd as R
    for (rate, raw), want in zip(GOLD["wav_args"], GOLD["wav_headers"]):
        assert R.wav_header(int(rate), bool(raw)) == want.tobytes()


@pytest.mark.ref
@pytest.mark.parametrize("fmt", ["CU8", "CS8", "CF32"])
def test_oracle_vs_reference_main(fmt):
    if not support.have_ref():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(42)
    x = rng.integers(-32768, 32768, size=2 * 70001, dtype=np.int16)
    for chunk in (0, 4096, 1000):
        want = support.ref_sdr_convert(fmt, x, chunk=chunk)
        got = support.oracle_sdr_convert(fmt, x)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


@pytest.mark.ref
def test_oracle_cs12_vs_reference_main():
    if not support.have_ref():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(43)
    b = rng.integers(0, 256, size=3 * 50021, dtype=np.uint8)
    assert np.array_equal(support.oracle_sdr_convert("CS16", b), support.ref_sdr_convert("CS16", b, chunk=5000))
