"""tools/bench_show.py FILE -- the parity verdicts and the main figures of a bench.py JSON line"""
import json, sys
d = json.load(open(sys.argv[1]))
print("headline", round(d["value"] / 1e6, 3), "TS/s  frac", round(d["roofline"]["frac"], 3), " parity_ok", d.get("parity_ok"), d.get("parity_all_legs"))
print("fm legs", d.get("parity_legs"), "cpu s", d.get("parity_cpu_seconds"), "wall", d.get("parity_wall_seconds_all_legs"), d.get("parity_how"))
for k, v in d.get("rx_fm_variants", {}).items():
    p = v.get("parity", {})
    print("  %-45s %.3f TS/s frac %.3f parity %s cpu %.1fs fix %s %s" % (k[:45], v["value"] / 1e6, v["frac_of_hbm_peak"], p.get("parity_ok"), p.get("parity_cpu_seconds", 0), p.get("parity_host_fixups"), v["stage_us_per_step"]))
hf = d.get("host_fed", {})
print("host_fed", {k: round(v["GS/s"], 2) for k, v in hf.get("legs", {}).items()}, hf.get("parity"), hf.get("dropin_block_us"))
pw = d.get("rx_power", d if "Mbins" in d.get("unit", "") else {})
if pw:
    print("rx_power", round(pw["value"] / 1e3, 1), "Gbins/s", pw["roofline"].get("frac"), pw["roofline"].get("hbm_frac"), pw.get("parity"))
    for k, v in pw.get("other_geometries", {}).items():
        print("  %-40s %.1f Gbins/s %s" % (k[:40], v["Mbins/s"] / 1e3, v.get("parity")))
ch = d.get("channeliser")
if ch:
    print("chan", round(ch["value"] / 1e3, 1), "GS/s", ch.get("parity"))
print("sdr", {k: round(v["GB/s"]) for k, v in d.get("sdr_convert", {}).get("legs", {}).items()}, d.get("sdr_convert", {}).get("parity_ok"))
