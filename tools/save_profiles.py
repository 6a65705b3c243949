"""Copy the rocprofv3 outputs of a gpurun call into profiles/ (tracked) and derive r01_traffic.json.
Expects gpurun_out/{r1stats,r1fetch,r1write,r1lm,r1def}/run_*.csv and gpurun_out/{r1stats,r1fetch,r1write,bench_r1}.log."""
import csv, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
os.makedirs(P, exist_ok=True)
for a, b in [("r1stats/run_kernel_stats.csv", "r01_kernel_stats.csv"), ("r1stats/run_domain_stats.csv", "r01_domain_stats.csv"),
             ("r1stats/run_kernel_trace.csv", "r01_kernel_trace.csv"), ("r1lm/run_kernel_stats.csv", "r01_localmap_kernel_stats.csv")]:
    shutil.copy(os.path.join(G, a), os.path.join(P, b))
keep = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
        "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
vals = {}
for src, dst, name in [("r1fetch/run_counter_collection.csv", "r01_pmc_fetch_size.csv", "FETCH_SIZE"),
                       ("r1write/run_counter_collection.csv", "r01_pmc_write_size.csv", "WRITE_SIZE")]:
    rows = list(csv.DictReader(open(os.path.join(G, src))))
    with open(os.path.join(P, dst), "w", newline="") as f:
        w = csv.DictWriter(f, keep); w.writeheader()
        for r in rows:
            r = {k: r[k] for k in keep}; r["Kernel_Name"] = r["Kernel_Name"].split("(")[0][-60:]; w.writerow(r)
    al = [float(r["Counter_Value"]) for r in rows if "ndt_align_kernel" in r["Kernel_Name"]]
    vals[name] = sum(al) / len(al)
for n, out in [("r1stats.log", "r01_stats.log"), ("r1fetch.log", "r01_fetch.log"), ("r1write.log", "r01_write.log"),
               ("bench_r1.log", "r01_bench.json")]:
    js = [l for l in open(os.path.join(G, n)) if l.startswith("{")]
    open(os.path.join(P, out), "w").write(js[-1])
traffic = {
    "kernel": "ndt_align_kernel", "workload": "bench.py default (256 scans x 10k pts vs 1M-pt map)",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, mean over the batch launches of "
              "`bench.py --no-single-scan --no-cpu-baseline --steps 8` (profiles/r01_pmc_*.csv)",
    "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
    "bytes_per_launch": (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
    "note": "raw counters x 1024 (rocprofv3 reports KB).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads half the bytes "
            "of wide coalesced streaming reads; this kernel's reads are mostly 8-byte gathers and record loads, a pattern "
            "the guide leaves uncalibrated, so the figure is not doubled (upper bound with doubling: fetch x 2)."}
json.dump(traffic, open(os.path.join(P, "r01_traffic.json"), "w"), indent=1)
import re
st = {}
for r in csv.DictReader(open(os.path.join(P, "r01_kernel_stats.csv"))):
    m = re.search(r"(ndt_\w+|map_\w+|scan_\w+|fill_\w+|prefilter_\w+)", r["Name"])
    st[m.group(1) if m else r["Name"][:40]] = r
k = st["ndt_align_kernel"]
print("align kernel: calls %s avg %.1f us | fetch %.0f MB write %.0f MB | traffic %.0f MB" % (
    k["Calls"], float(k["AverageNs"]) / 1e3, vals["FETCH_SIZE"] * 1024 / 1e6, vals["WRITE_SIZE"] * 1024 / 1e6, traffic["bytes_per_launch"] / 1e6))
for n in ("r01_stats.log", "r01_bench.json"):
    d = json.load(open(os.path.join(P, n)))
    print(n, round(d["value"]), "ms/step %.4f kernel_ms %.4f frac %.4f map %.4f single %s" % (
        d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["map_build_ms"], d.get("single_scan_ms")))
for name, r in st.items():
    if name.startswith("map_") or name.startswith("scan_") or name.startswith("fill_"):
        print("  %-28s %6.1f us" % (name, float(r["AverageNs"]) / 1e3))

# the default command itself (gpurun_out/r1def): summary + the match-kernel dispatches labelled by launch kind
dd = os.path.join(G, "r1def")
if os.path.exists(os.path.join(dd, "run_kernel_trace.csv")):
    shutil.copy(os.path.join(dd, "run_kernel_stats.csv"), os.path.join(P, "r01_default_cmd_kernel_stats.csv"))
    rows = sorted(csv.DictReader(open(os.path.join(dd, "run_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
    al = [r for r in rows if "ndt_align" in r["Kernel_Name"]]
    line = [l for l in open(os.path.join(G, "r1def.log")) if l.startswith("{")][-1]
    d = json.loads(line)
    kinds = (["warm-up"] * d["warmup"] + ["timed step"] * d["steps"] + ["single scan (configs[1])"] * 5 +
             ["front-end step (filtered raw scans)"] * 5)
    assert len(al) == len(kinds), (len(al), len(kinds))
    with open(os.path.join(P, "r01_default_cmd_align_dispatches.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel", "Workgroup_Size", "Grid_Size", "Start_Timestamp", "End_Timestamp", "Duration_us", "launch"])
        for r, k in zip(al, kinds):
            w.writerow([r["Dispatch_Id"], "ndt_align_kernel", r["Workgroup_Size_X"], r["Grid_Size_X"], r["Start_Timestamp"],
                        r["End_Timestamp"], "%.3f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3), k])
    open(os.path.join(P, "r01_default_cmd.log"), "w").write(line)
    timed = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r, k in zip(al, kinds) if k == "timed step"]
    print("default command: timed steps average %.1f us in the trace, %.1f us from its HIP events" % (
        sum(timed) / len(timed), d["roofline"]["kernel_ms"] * 1e3))

