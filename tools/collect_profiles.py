#!/usr/bin/env python3
"""Copy the rocprofv3 evidence of gpurun_out/prof_TAG (tools/profile_round.sh) into profiles/ under round names and fold the
PMC passes into profiles/rNN_pmc_summary.json (per kernel, per launch shape).

    python tools/collect_profiles.py r02 gpurun_out/prof_r02b

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes, so it is doubled
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported.  Launches of the same kernel with different grids (the
drop-in's 1 MiB blocks beside the 4 GiB launches of the bench) are kept apart by grid size; the summary names the largest.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n.strip()


def fold(path):
    """{kernel: {grid: {counter: mean}}} with the number of dispatches"""
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(list)))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, grids in acc.items():
        if not k.startswith("k_"):
            continue
        g = max(grids)                                        # the bench's launch shape
        out[k] = {c: sum(v) / len(v) for c, v in grids[g].items()}
        out[k]["_grid_size"] = g
        out[k]["_dispatches"] = len(next(iter(grids[g].values())))
    return out


def one(pattern):
    hits = glob.glob(pattern)
    return hits[0] if hits else None


def main():
    tag, src = sys.argv[1], sys.argv[2]
    dst = os.path.join(ROOT, "profiles")
    copies = [("bench_n1.json", "%s_bench_n1.json"), ("trace_bench.json", "%s_bench_profiled_run.json"),
              ("trace/runc/*_kernel_stats.csv", "%s_bench_kernel_stats.csv"),
              ("trace_variants/runc/*_kernel_stats.csv", "%s_variants_kernel_stats.csv"),
              ("pmc_FETCH_SIZE/runc/*_counter_collection.csv", "%s_rx_fm_pmc_FETCH_SIZE.csv"),
              ("pmc_WRITE_SIZE/runc/*_counter_collection.csv", "%s_rx_fm_pmc_WRITE_SIZE.csv"),
              ("pmc_valu_power/runc/*_counter_collection.csv", "%s_rx_power_pmc_valu_lds.csv"),
              ("pmc_valu_fm/runc/*_counter_collection.csv", "%s_rx_fm_pmc_valu.csv")]
    for pat, name in copies:
        f = one(os.path.join(src, pat))
        if not f:
            continue
        if "counter_collection" in f:
            # keep this library's kernels at the bench's launch shapes; the drop-in latency loop adds thousands of 1 MiB launches
            rows = list(csv.DictReader(open(f)))
            keep, seen = [], defaultdict(int)
            for r in rows:                                      # at most 6 dispatches per kernel, launch shape and counter
                if not (short(r["Kernel_Name"]).startswith("k_") and int(r["Grid_Size"]) >= (1 << 16)):
                    continue
                key = (r["Kernel_Name"], r["Grid_Size"], r["Counter_Name"])
                seen[key] += 1
                if seen[key] <= 6:
                    keep.append(r)
            with open(os.path.join(dst, name % tag), "w", newline="") as o:
                w = csv.DictWriter(o, fieldnames=list(rows[0].keys()))
                w.writeheader()
                w.writerows(keep)
        else:
            shutil.copy(f, os.path.join(dst, name % tag))
    # the per-dispatch trace is large: keep only our kernels' rows
    f = one(os.path.join(src, "trace/runc/*_kernel_trace.csv"))
    if f:
        rows = list(csv.DictReader(open(f)))
        keep = [r for r in rows if "k_" in r["Kernel_Name"].split("(")[0]]
        with open(os.path.join(dst, "%s_bench_kernel_trace.csv" % tag), "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(keep)
    res = {}
    fetch = fold(os.path.join(dst, "%s_rx_fm_pmc_FETCH_SIZE.csv" % tag))
    write = fold(os.path.join(dst, "%s_rx_fm_pmc_WRITE_SIZE.csv" % tag))
    for k in sorted(set(fetch) | set(write)):
        fk, wk = fetch.get(k, {}), write.get(k, {})
        f_, w_ = fk.get("FETCH_SIZE", 0.0), wk.get("WRITE_SIZE", 0.0)
        res[k] = {"FETCH_SIZE_KiB": f_, "WRITE_SIZE_KiB": w_, "hbm_bytes_per_launch": (2.0 * f_ + w_) * 1024.0,
                  "grid_size": fk.get("_grid_size", wk.get("_grid_size")), "dispatches": fk.get("_dispatches"),
                  "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncorrected"}
    res["_command"] = ("rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python bench.py --steps 2 "
                       "--warmup 1 --cpu-seconds 0 --workload rx_fm --variants none --no-parity --blocks 8192  (8192 blocks = 4 GiB per launch)")
    for name, key in (("%s_rx_power_pmc_valu_lds.csv" % tag, "rx_power"), ("%s_rx_fm_pmc_valu.csv" % tag, "rx_fm")):
        p = os.path.join(dst, name)
        if not os.path.exists(p):
            continue
        for k, d in fold(p).items():
            e = res.setdefault(k, {})
            e.update({c: v for c, v in d.items() if not c.startswith("_")})
            e["valu_grid_size"] = d["_grid_size"]
            if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_INSTS_VALU"):
                cyc = d["GRBM_GUI_ACTIVE"] / 8.0                       # summed over 8 XCDs
                e["shader_cycles_per_xcd"] = cyc
                e["valu_wave_instr_per_simd_cycle"] = d["SQ_INSTS_VALU"] / (cyc * 1024.0)
                e["valu_issue_fraction_of_peak_4cyc_per_wave64_instr"] = 4.0 * d["SQ_INSTS_VALU"] / (cyc * 1024.0)
            if d.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_conflict_fraction"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    json.dump(res, open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w"), indent=1)
    for k in ("k_fm_decimate<false, true, true, true>", "k_pw_fft4096<2, false>"):
        print(k, json.dumps(res.get(k), indent=1)[:900])


if __name__ == "__main__":
    main()
