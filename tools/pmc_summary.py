#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter CSVs into profiles/rNN_pmc_summary.json (per kernel, averaged over its launches).

    python tools/pmc_summary.py OUT.json FETCH.csv WRITE.csv [POWER_VALU.csv]

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes, so it is doubled
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported.
"""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "")
    return n.split("<")[0].strip()


def fold(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main():
    out, fetch, write = sys.argv[1], fold(sys.argv[2]), fold(sys.argv[3])
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f = fetch.get(k, {}).get("FETCH_SIZE", 0.0)
        w = write.get(k, {}).get("WRITE_SIZE", 0.0)
        res[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                  "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncorrected"}
    res["_command"] = ("rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python bench.py "
                       "--steps 2 --warmup 1 --cpu-seconds 0 --workload rx_fm --blocks 8192  (8192 blocks = 4 GiB per launch)")
    if len(sys.argv) > 4:
        pw = fold(sys.argv[4])
        for k, d in pw.items():
            if not k.startswith("k_pw_fft"):
                continue
            e = dict(d)
            if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_INSTS_VALU"):
                cyc = d["GRBM_GUI_ACTIVE"] / 8.0                       # summed over 8 XCDs
                e["shader_cycles_per_xcd"] = cyc
                e["valu_wave_instr_per_simd_cycle"] = d["SQ_INSTS_VALU"] / (cyc * 1024.0)
                e["valu_issue_fraction_of_peak_4cyc_per_wave64_instr"] = 4.0 * d["SQ_INSTS_VALU"] / (cyc * 1024.0)
            if d.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_conflict_fraction"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
            res[k] = e
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k in ("k_fm_decimate", "k_pw_fft4096")}, indent=1))


if __name__ == "__main__":
    main()
