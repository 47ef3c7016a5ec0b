#!/usr/bin/env python3
"""Fold gpurun_out/prof_TAG (tools/profile_round4.sh) into profiles/:

    python tools/collect_round4.py r04 gpurun_out/prof_r04

Everything tools/collect_round3.py writes (bench lines, kernel trace + stats, per-chain counters, the summary bench.py reads, the VALU issue table,
the mixed-traffic ceilings), plus
  rNN_pmc_power_legs.json   the other rx_power geometries of the bench line (N = 2^14 with -F 9 / boxcar, 2^15, 2^18): per kernel and per launch
                            VALU wave-instructions, fetched + written HBM bytes, shader cycles -- what bench.py turns into each leg's roofline
  rNN_pmc_chains.json       gains "rx_power configs[2], twiddles through the vector cache" (the A/B of the LDS table in counters)
  rNN_ab_*.txt, rNN_cndmask_probe.txt   the in-process A/B runs and the cndmask probe as printed
"""
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import collect_round3 as c3  # noqa: E402

ROOT = c3.ROOT
LEGS = {  # name -> (bench label prefix, passes of the profiled launch, runs in the profiled process)
    "n14_fir9": ("-f 100M:100.1M:10 -F 9", 4096, 2),
    "n14_box28": ("-f 100M:100.1M:10 (boxcar ds=28)", 4096, 2),
    "n15_box14": ("-f 100M:100.2M:10 (boxcar ds=14)", 2048, 2),
    "n18": ("-f 100M:102.8M:20", 256, 2),
}


def main():
    tag, src = sys.argv[1], sys.argv[2]
    sys.argv = [sys.argv[0], tag, src]
    c3.main()
    dst = os.path.join(ROOT, "profiles")
    for name in ("ab_power_tw.txt", "ab_chan_tw.txt", "ab_power_big.txt", "cndmask_probe.txt"):
        f = os.path.join(src, name)
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, name)))
    legs = {}
    for name, (label, passes, runs) in LEGS.items():
        e = c3.chain_entry(src, "leg_" + name, runs, 0.0, "tools/pw_big_once.py, %d passes per launch, %d launches in the profiled process" % (passes, runs))
        if not e:
            continue
        tot_valu = sum(k.get("valu_wave_instr_per_step", 0.0) for k in e["kernels"].values())
        tot_cyc = sum(k.get("shader_cycles_per_step", 0.0) for k in e["kernels"].values())
        legs[label] = {"passes": passes, "valu_wave_instr_per_launch": tot_valu, "hbm_bytes_per_launch": e["hbm_bytes_per_step"],
                       "shader_cycles_per_launch_sum_of_kernels": tot_cyc, "kernels": e["kernels"], "what": e["what"]}
    json.dump(legs, open(os.path.join(dst, "%s_pmc_power_legs.json" % tag), "w"), indent=1)
    chains_path = os.path.join(dst, "%s_pmc_chains.json" % tag)
    chains = json.load(open(chains_path))
    e = c3.chain_entry(src, "rx_power_twglobal", 3, 4.0 * 512 * 599 * 8192,
                       "RXGPU_FFT_TW=global bench.py --workload rx_power --steps 2 --warmup 1: the configs[2] launches with the stage 4-11 twiddles through the vector cache (rounds 1-3)")
    if e:
        chains["rx_power configs[2], twiddles through the vector cache (A/B)"] = e
        json.dump(chains, open(chains_path, "w"), indent=1)
    for label, v in legs.items():
        print("%-40s valu %.1f M wave-instr  HBM %.3f GB per launch" % (label, v["valu_wave_instr_per_launch"] / 1e6, v["hbm_bytes_per_launch"] / 1e9))
        for kk, k in v["kernels"].items():
            if k.get("valu_wave_instr_per_step", 0) > 1e5:
                print("      %-50s %8.3f GB  valu %7.1f M  %.3f/SIMD-cycle  waiting %.2f" % (kk[:50], k.get("hbm_bytes_per_step", 0) / 1e9, k.get("valu_wave_instr_per_step", 0) / 1e6,
                                                                                            k.get("valu_wave_instr_per_simd_cycle", 0) or 0, k.get("wave_time_waiting", 0) or 0))


if __name__ == "__main__":
    main()
