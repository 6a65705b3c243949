"""Rewrite the measured numbers in DESIGN.md / README.md / profiles/README.md from profiles/r01_*.json and csv."""
import csv, json, os, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(R, "profiles", "r01_bench.json")))
st = json.load(open(os.path.join(R, "profiles", "r01_stats.log")))
t = json.load(open(os.path.join(R, "profiles", "r01_traffic.json")))
rows = list(csv.DictReader(open(os.path.join(R, "profiles", "r01_kernel_stats.csv"))))
k = next(r for r in rows if "ndt_align_kernel" in r["Name"])
prof_avg, prof_calls = float(k["AverageNs"]) / 1e3, int(k["Calls"])
cb = d["cpu_baseline"]
p = os.path.join(R, "DESIGN.md"); s = open(p).read()
a = s.index("Round-1 numbers (1 × MI355X, `profiles/r01_bench.json`):"); b = s.index("History of the round (same workload):")
s = s[:a] + f"""Round-1 numbers (1 × MI355X, `profiles/r01_bench.json`):

| | |
|---|---|
| value | {d['value']:,.0f} matches/s ({d['ms_per_step']:.3f} ms per step of 256) |
| `ndt_align_kernel` | {d['roofline']['kernel_ms']:.3f} ms per batch launch (HIP events); rocprofv3 under the profiler: {prof_avg:.1f} µs average over {prof_calls} launches vs {st['roofline']['kernel_ms']*1e3:.1f} µs from the events of that run (`r01_kernel_stats.csv`, `r01_stats.log`) |
| map build | {d['map_build_ms']:.3f} ms |
| one scan (configs[1]) | {d['single_scan_ms']:.3f} ms |
| roofline | {d['roofline']['achieved']:.0f} GB/s algorithmic, frac {d['roofline']['frac']:.3f}; measured traffic {t['bytes_per_launch']/1e9:.2f} GB per launch (FETCH + WRITE, raw) |
| source pre-filter (row f1) | 256 scans × 30k raw points in {d['prefilter']['ms']:.2f} ms ({d['prefilter']['raw_points_per_s']/1e9:.1f} G points/s) |
| whole front-end step | predict → pre-filter → map rebuild → match → fuse for 256 raw scans of 30k points, all on the device: {d['front_end_step']['ms']:.2f} ms ({d['front_end_step']['scans_per_s']/1e3:.0f}k scans/s) |
| local-map assembly (row f3) | `Submap::makeMap` of {d['local_map']['scans']} scans × {d['local_map']['points']//d['local_map']['scans']} points with moving-object removal: {d['local_map']['ms']:.2f} ms (oracle's pointer octree on one host core: {d['local_map']['cpu_ms_1core']:.0f} ms; identical cloud: {d['local_map']['identical']}) |
| CPU baseline | oracle (port), 1 thread: {cb['value']:.1f} matches/s; {cb['all_cores']['cores']} threads: {cb['all_cores']['value']:.0f} matches/s |
| parity on the bench sample | max |Δpos| = {d['parity']['max_dpos_m']} m, max |Δyaw| = {d['parity']['max_dyaw_rad']} rad, identical iteration counts ({d['parity']['sample']} scans) |

""" + s[b:]
s = re.sub(r"so the GPU/CPU ratio \(≈ \d+× one core,\n≈ \d+× \d+ threads\)", f"so the GPU/CPU ratio (≈ {d['value']/cb['value']:.0f}× one core,\n≈ {d['value']/cb['all_cores']['value']:.0f}× {cb['all_cores']['cores']} threads)", s)
s = re.sub(r"(### 4\.1 Map build \(a2\): `ndt_map_build_dev` — )[\d.]+( ms per 1M-point map)", r"\g<1>%.2f\g<2>" % d["map_build_ms"], s)
s = re.sub(r"(### 4\.2 Match kernel \(a3–a9\): `ndt_align_kernel<SSE, INCL>` — )[\d.]+( ms per 256 matches)", r"\g<1>%.2f\g<2>" % d["roofline"]["kernel_ms"], s)
s = re.sub(r"\*\*1\.50 GB algorithmic per launch\*\*; kernel [\d.]+ ms ⇒ \*\*\d+ GB/s = [\d.]+ of the 8 TB/s HBM peak\*\*",
           "**1.50 GB algorithmic per launch**; kernel %.3f ms ⇒ **%.0f GB/s = %.3f of the 8 TB/s HBM peak**" % (d["roofline"]["kernel_ms"], d["roofline"]["achieved"], d["roofline"]["frac"]), s)
s = re.sub(r"show FETCH_SIZE = \d+ MB and WRITE_SIZE = \d+ MB per launch", "show FETCH_SIZE = %.0f MB and WRITE_SIZE = %.0f MB per launch" % (t["fetch_size_kb"] * 1024 / 1e6, t["write_size_kb"] * 1024 / 1e6), s)
s = re.sub(r"be ≤ \d+ MB\)", "be ≤ %.0f MB)" % (2 * t["fetch_size_kb"] * 1024 / 1e6), s)
s = re.sub(r"the measured [\d.]+ ms is the critical path", "the measured %.2f ms is the critical path" % d["roofline"]["kernel_ms"], s)
open(p, "w").write(s)
p = os.path.join(R, "README.md"); s = open(p).read()
s = re.sub(r"Round-1 result on one MI355X: \d+k scan-matches/s \([\d.]+ ms per batch of 256 including the map\nrebuild\)",
           "Round-1 result on one MI355X: %dk scan-matches/s (%.2f ms per batch of 256 including the map\nrebuild)" % (round(d["value"] / 1000), d["ms_per_step"]), s)
open(p, "w").write(s)
p = os.path.join(R, "profiles", "README.md"); s = open(p).read()
s = re.sub(r"\([\d.]+ us here vs [\d.]+ us from the HIP events of the same run, `r01_stats.log`; [\d.]+ us un-profiled\)",
           "(%.1f us here vs %.1f us from the HIP events of the same run, `r01_stats.log`; %.1f us un-profiled)" % (prof_avg, st["roofline"]["kernel_ms"] * 1e3, d["roofline"]["kernel_ms"] * 1e3), s)
open(p, "w").write(s)
print("synced: value %.0f kernel %.3f ms frac %.3f" % (d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
