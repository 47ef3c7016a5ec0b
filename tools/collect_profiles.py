#!/usr/bin/env python3
"""Fold gpurun_out/prof_TAG (tools/profile_round.sh TAG) into profiles/ under round names:

    python tools/collect_profiles.py r05 gpurun_out/prof_r05

  rNN_bench_n1.json, rNN_bench_full.json                   the driver's bench command: the compact stdout line and the full record
  rNN_bench_profiled_run.json                               the same command's full record under --kernel-trace
  rNN_bench_kernel_stats.csv, rNN_bench_kernel_trace.csv   rocprofv3 --kernel-trace --stats of the headline command (our kernels' rows)
  rNN_variants_kernel_stats.csv                            the same for the rx_fm variants
  rNN_chan_audio_kernel_stats.csv, rNN_chan_audio.txt      the channeliser's per-channel audio stages (tools/chan_audio_time.py under the trace)
  rNN_pmc_chains.json     per rx_fm chain / rx_power / channeliser: per kernel {launches, VALU wave-instructions, shader cycles, waves waiting,
                          FETCH_SIZE, WRITE_SIZE, LDS conflict fraction} and the chain's HBM bytes per step beside its algorithmic bytes
  rNN_pmc_summary.json    the per-kernel entries bench.py reads (k_fm_decimate traffic, k_pw_fft4096 / k_ch_fftR instruction counts, per-chain traffic)
  rNN_pmc_chan_modes.json the channeliser's other modes (-A std, audio stages, NCO): per kernel VALU wave-instructions and VALU-active quad-cycles per run
                          -- the measured VALU busy time behind those legs' `valu` bound in the bench line
  rNN_pmc_power_legs.json the other rx_power geometries of the bench line: per kernel and per launch VALU wave-instructions, HBM bytes, shader cycles
  rNN_dropin_latency.txt  rxgpu_callback + rxgpu_full_demod per 1 MiB block, by phase
  rNN_scan_latency.txt    rxgpu_scan / rxgpu_scan_sync on the configs[2] sweep, by phase, beside one hipMemcpy of the same bytes (tools/scan_latency.py)
  rNN_pw_big_time.txt     G bins/s of the large-N rx_power geometries, one tune each (tools/pw_big_time.py)
  rNN_valu_issue.json/.txt, rNN_rwmix.json/.txt   (section `probes`) wave64 instructions per SIMD-cycle per opcode; what HBM gives a read stream with writes mixed in

Only what the profile directory holds is written: a section that was not run leaves the round without that file and bench.py falls back to the newest
round that has it.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes, so it is doubled
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def one(pattern):
    """the NEWEST match: gpurun merges a run's files into what earlier runs left in the same directory"""
    hits = glob.glob(pattern)
    return max(hits, key=os.path.getmtime) if hits else None


def per_kernel(path, min_grid=0):
    """{kernel: {counter: SUM over its dispatches, '_dispatches': n}} for our kernels"""
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_") or k.startswith("k_diag_") or int(r["Grid_Size"]) < min_grid:   # k_diag_*: bench.py's box-ceiling probe, no leg's kernel
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    out = {}
    for k, d in acc.items():
        out[k] = dict(d)
        out[k]["_dispatches"] = max(cnt[k].values())
    return out


def chain_entry(src, prefix, steps, algorithmic_bytes_per_step, what):
    merged = {}
    for pass_dir in sorted(glob.glob(os.path.join(src, prefix + "_p*"))):
        f = one(os.path.join(pass_dir, "*", "*_counter_collection.csv"))
        if not f:
            continue
        for k, d in per_kernel(f).items():
            e = merged.setdefault(k, {})
            for c, v in d.items():
                e[c] = v if c != "_dispatches" else max(v, e.get(c, 0))
    if not merged:
        return None
    kernels, total = {}, 0.0
    for k, d in sorted(merged.items()):
        n = d.get("_dispatches", 1)
        e = {"dispatches_in_run": n}
        if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
            e["hbm_bytes_per_step"] = (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0 / steps
            e["fetch_bytes_per_step"] = 2.0 * d.get("FETCH_SIZE", 0.0) * 1024.0 / steps
            e["write_bytes_per_step"] = d.get("WRITE_SIZE", 0.0) * 1024.0 / steps
            total += e["hbm_bytes_per_step"]
        if d.get("SQ_INSTS_VALU") is not None:
            e["valu_wave_instr_per_step"] = d["SQ_INSTS_VALU"] / steps
        if d.get("SQ_ACTIVE_INST_VALU") is not None:
            # quad-cycles in which a SIMD's VALU was executing an instruction of this kernel (a multi-pass fp64 op counts every pass)
            e["valu_active_quad_cycles_per_step"] = d["SQ_ACTIVE_INST_VALU"] / steps
        if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_INSTS_VALU") is not None:
            cyc = d["GRBM_GUI_ACTIVE"] / 8.0                                    # summed over the 8 XCDs
            e["shader_cycles_per_step"] = cyc / steps
            e["valu_wave_instr_per_simd_cycle"] = d["SQ_INSTS_VALU"] / (cyc * 1024.0)
        if d.get("SQ_WAVE_CYCLES"):
            e["wave_time_waiting"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
        if d.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_fraction"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        kernels[k] = e
    return {"what": what, "steps_in_run": steps, "algorithmic_bytes_per_step": algorithmic_bytes_per_step,
            "hbm_bytes_per_step": total, "traffic_over_algorithmic": total / algorithmic_bytes_per_step if algorithmic_bytes_per_step else None,
            "kernels": kernels}



LEGS = {  # name -> (bench label prefix, passes of the profiled launch, runs in the profiled process)
    "n14_fir9": ("-f 100M:100.1M:10 -F 9", 4096, 2),
    "n14_box28": ("-f 100M:100.1M:10 (boxcar ds=28)", 4096, 2),
    "n15_box14": ("-f 100M:100.2M:10 (boxcar ds=14)", 2048, 2),
    "n18": ("-f 100M:102.8M:20", 256, 2),
}


def main():
    tag, src = sys.argv[1], sys.argv[2]
    dst = os.path.join(ROOT, "profiles")
    for pat, name in (("bench_n1.json", "%s_bench_n1.json"), ("bench_full.json", "%s_bench_full.json"), ("trace_bench_full.json", "%s_bench_profiled_run.json"),
                      ("trace/*/*_kernel_stats.csv", "%s_bench_kernel_stats.csv"), ("trace_variants/*/*_kernel_stats.csv", "%s_variants_kernel_stats.csv"),
                      ("trace_chan_audio/*/*_kernel_stats.csv", "%s_chan_audio_kernel_stats.csv"), ("chan_audio.txt", "%s_chan_audio.txt"),
                      ("dropin_latency.txt", "%s_dropin_latency.txt"), ("scan_latency.txt", "%s_scan_latency.txt"), ("pw_big_time.txt", "%s_pw_big_time.txt"),
                      ("valu_issue.json", "%s_valu_issue.json"), ("valu_issue.txt", "%s_valu_issue.txt"),
                      ("rwmix.json", "%s_rwmix.json"), ("rwmix.txt", "%s_rwmix.txt")):
        f = one(os.path.join(src, pat))
        if f and os.path.getsize(f):
            shutil.copy(f, os.path.join(dst, name % tag))
    f = one(os.path.join(src, "trace/*/*_kernel_trace.csv"))
    if f:
        rows = list(csv.DictReader(open(f)))
        keep = [r for r in rows if "k_" in r["Kernel_Name"].split("(")[0]]
        with open(os.path.join(dst, "%s_bench_kernel_trace.csv" % tag), "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(keep)
    chains = {}
    T4 = 8192 * 131072                                                           # samples per chain_once.py run
    for ds, label in ((118, "headline: low_pass ds=118"), (6, "-M wbfm default, downsample=6"), (5, "configs[0] geometry: ds=5, 240 kHz"),
                      (-7, "-F cascade, 7 passes (ds=128)"), (-39, "-M wbfm -F 9: 3 passes + droop FIR")):
        e = chain_entry(src, "chain_%d" % ds, 2, 4.0 * T4,
                        "tools/chain_once.py 8192 %d 2: two pipelined runs of 8192 blocks (4 GiB each); counters summed over every kernel of the chain, per run" % ds)
        if e:
            chains[label] = e
    e = chain_entry(src, "rx_power", 3, 4.0 * 512 * 599 * 8192, "bench.py --workload rx_power --steps 2 --warmup 1: three 512-pass launches of the configs[2] geometry")
    if e:
        chains["rx_power configs[2]"] = e
    e = chain_entry(src, "chan", 6, 4.0 * 2048 * 131072, "tools/chan_once.py: six runs of the bench shape, 1 GiB of capture each")
    if e:
        chains["channeliser"] = e
    modes = {}
    for mode, samples in (("std", 2048 * 131072), ("audio", 2048 * 131072), ("nco", 256 * 131072)):
        e = chain_entry(src, "chanmode_" + mode, 6, 4.0 * samples, "tools/chan_once.py %s: six runs of the bench leg's shape" % mode)
        if e:
            modes[mode] = {"samples_per_run": samples, "what": e["what"], "kernels": e["kernels"],
                           "valu_wave_instr_per_run": sum(k.get("valu_wave_instr_per_step", 0.0) for k in e["kernels"].values()),
                           "valu_active_quad_cycles_per_run": sum(k.get("valu_active_quad_cycles_per_step", 0.0) for k in e["kernels"].values())}
    if modes:
        json.dump(modes, open(os.path.join(dst, "%s_pmc_chan_modes.json" % tag), "w"), indent=1)
    if chains:
        json.dump(chains, open(os.path.join(dst, "%s_pmc_chains.json" % tag), "w"), indent=1)
        # what bench.py reads
        summ = {}
        hd = chains.get("headline: low_pass ds=118", {}).get("kernels", {})
        for k, v in hd.items():
            if k.startswith("k_fm_decimate<") and "hbm_bytes_per_step" in v:
                summ[k] = {"hbm_bytes_per_launch": v["hbm_bytes_per_step"], "launch_samples": T4,
                           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncorrected"}
        for src_label, prefix in (("rx_power configs[2]", "k_pw_fft4096"), ("channeliser", "k_ch_fft")):
            for k, v in chains.get(src_label, {}).get("kernels", {}).items():
                if k.startswith(prefix) and "valu_wave_instr_per_step" in v:
                    summ[k] = {"SQ_INSTS_VALU": v["valu_wave_instr_per_step"], "hbm_bytes_per_launch": v.get("hbm_bytes_per_step"),
                               "valu_wave_instr_per_simd_cycle": v.get("valu_wave_instr_per_simd_cycle"), "wave_time_waiting": v.get("wave_time_waiting"),
                               "lds_conflict_fraction": v.get("lds_conflict_fraction")}
        summ["_chains"] = {k: {"hbm_bytes_per_step": v["hbm_bytes_per_step"], "algorithmic_bytes_per_step": v["algorithmic_bytes_per_step"],
                               "traffic_over_algorithmic": v["traffic_over_algorithmic"]} for k, v in chains.items()}
        json.dump(summ, open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w"), indent=1)
    legs = {}
    for name, (label, passes, runs) in LEGS.items():
        e = chain_entry(src, "leg_" + name, runs, 0.0, "tools/pw_big_once.py, %d passes per launch, %d launches in the profiled process" % (passes, runs))
        if not e:
            continue
        tot_valu = sum(k.get("valu_wave_instr_per_step", 0.0) for k in e["kernels"].values())
        tot_cyc = sum(k.get("shader_cycles_per_step", 0.0) for k in e["kernels"].values())
        legs[label] = {"passes": passes, "valu_wave_instr_per_launch": tot_valu, "hbm_bytes_per_launch": e["hbm_bytes_per_step"],
                       "shader_cycles_per_launch_sum_of_kernels": tot_cyc, "kernels": e["kernels"], "what": e["what"]}
    if legs:
        json.dump(legs, open(os.path.join(dst, "%s_pmc_power_legs.json" % tag), "w"), indent=1)
    for k, v in chains.items():
        print("%-44s HBM bytes/step %.3f GB = %.3f x algorithmic" % (k, v["hbm_bytes_per_step"] / 1e9, v["traffic_over_algorithmic"] or 0))
        for kk, e in v["kernels"].items():
            if e.get("hbm_bytes_per_step", 0) > 5e7 or e.get("valu_wave_instr_per_step", 0) > 1e7:
                print("      %-56s %8.3f GB  valu %7.1f M  %.3f/SIMD-cycle  waiting %.2f  lds-conflict %s" % (
                    kk[:56], e.get("hbm_bytes_per_step", 0) / 1e9, e.get("valu_wave_instr_per_step", 0) / 1e6, e.get("valu_wave_instr_per_simd_cycle", 0) or 0,
                    e.get("wave_time_waiting", 0) or 0, ("%.2f" % e["lds_conflict_fraction"]) if e.get("lds_conflict_fraction") is not None else "-"))
    for label, v in legs.items():
        print("%-40s valu %.1f M wave-instr  HBM %.3f GB per launch" % (label, v["valu_wave_instr_per_launch"] / 1e6, v["hbm_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
