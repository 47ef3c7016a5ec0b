"""-m "not gpu": the product's drop-in executables BUILD from an unmodified rx_tools checkout (dropin/Makefile) and bind what they should.
Runs where /root/reference (the checkout) and librxgpu.so exist; the executables themselves run in tests/test_dropin_e2e.py on the GPU box."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
LIB = os.path.join(ROOT, "rx_tools_amd", "librxgpu.so")

pytestmark = pytest.mark.skipif(not (os.path.isdir(os.path.join(REF, "src")) and os.path.exists(LIB) and shutil.which("gcc") and shutil.which("make")),
                                reason="needs the rx_tools checkout, librxgpu.so, gcc and make")


def undefined(path):
    out = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip()}


@pytest.mark.parametrize("patch", [0, 1])
def test_dropin_makefile_builds_from_the_unmodified_checkout(tmp_path, patch):
    """make -C dropin REF=... [PATCH=1]: the checkout's rtl_fm.c / rtl_power.c compiled where they lie; full_demod, scanner and csv_dbm land in
    librxgpu at link time; with PATCH=1 exactly one line of a scratch copy is rewritten (the make fails otherwise, and deletes the copy) and the
    callback goes to the library too, through the wrapper that page-locks the read buffer"""
    before = {f: os.path.getmtime(os.path.join(REF, "src", f)) for f in ("rtl_fm.c", "rtl_power.c")}
    out = str(tmp_path / "bin")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "dropin"), "REF=" + REF, "OUT=" + out, "SOAPY=stub"] + (["PATCH=1"] if patch else []),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    fm, pw = os.path.join(out, "rx_fm"), os.path.join(out, "rx_power")
    assert os.access(fm, os.X_OK) and os.access(pw, os.X_OK)
    u_fm, u_pw = undefined(fm), undefined(pw)
    assert {"rxgpu_full_demod", "rxgpu_set_demod_functions", "rxgpu_dropin_pin", "rxgpu_dropin_block_rms"} <= u_fm
    assert {"rxgpu_scan", "rxgpu_scan_sync"} <= u_pw          # csv_dbm stays the file's own, behind the sync
    assert ("rxgpu_callback" in u_fm) == bool(patch) and ("rxgpu_pin" in u_fm) == bool(patch)
    # nothing of the checkout was edited, no scratch copy is left behind
    assert before == {f: os.path.getmtime(os.path.join(REF, "src", f)) for f in before}
    assert not [f for f in os.listdir(out) if f.endswith(".c")]
