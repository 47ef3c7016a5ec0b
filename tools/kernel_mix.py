#!/usr/bin/env python3
"""tools/kernel_mix.py OUT.json -- the static VALU opcode mix of the kernels whose roofline is VALU issue (hipcc -S of the sources, no GPU
needed): per kernel {opcode: count}.  bench.py weights the measured per-opcode issue rates (profiles/rNN_valu_issue.json) with it to get
the kernel's VALU ceiling.  Static counts of straight-line, fully unrolled transform code: the hot loop IS most of the listing."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rx_tools_amd", "csrc")
WANT = {"power_kernels.hip": ["k_pw_fft4096ILi2ELb0ELb1E", "k_pwm_tailILi14ELb0ELi1E", "k_pwm_tailILi18ELb0ELi2E", "k_pwm_headILi18E"],
        "fm_kernels.hip": ["k_ch_fftRILi10ELb1ELb1ELi16E", "k_fm_decimate_laneILb1ELi6ELi0E", "k_fm_fifth_regnILb1ELi4E", "k_ch_ncoILi0E"]}


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for src, names in WANT.items():
            asm = os.path.join(tmp, src + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                                   "-I" + CSRC, "-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
            cur = None
            for line in open(asm):
                m = re.match(r"^(_Z\w+):", line)
                if m:
                    cur = next((n for n in names if n in m.group(1)), None)
                    if cur:
                        out[cur] = collections.Counter()
                    continue
                if ".end_amdhsa_kernel" in line or line.startswith(".Lfunc_end"):
                    cur = None
                if cur:
                    m = re.match(r"^\t(v_\w+)", line)
                    if m:
                        op = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", m.group(1))
                        out[cur][op] += 1
    res = {k: dict(sorted(v.items(), key=lambda kv: -kv[1])) for k, v in out.items()}
    json.dump(res, open(sys.argv[1], "w"), indent=1)
    for k, v in res.items():
        print(k, sum(v.values()), list(v.items())[:8])


if __name__ == "__main__":
    main()
