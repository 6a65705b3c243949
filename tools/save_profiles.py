"""Copy the rocprofv3 outputs of a `tools/gpu_profile.sh <tag>` call into profiles/ (tracked) and derive the
summaries bench.py reads: python tools/save_profiles.py <tag> <round>, e.g. `r2z/prof 02`.
Expects gpurun_out/<tag>/{bench.log,stats.log,stats/,pmc_fetch/,pmc_write/,pmc_valu/,pmc_lds/}."""
import collections
import csv
import json
import os
import re
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
G, P = os.path.join(R, "gpurun_out", tag), os.path.join(R, "profiles")
pre = "r%s_" % rnd
os.makedirs(P, exist_ok=True)


def kname(full):
    m = re.search(r"(ndt_\w+|fitness_\w+|map_\w+|scan_\w+|fill_\w+|prefilter_\w+|make_map_\w+|predict_\w+|fuse_\w+|repack_\w+|__amd_rocclr_\w+)", full)
    return m.group(1) if m else full[:48]


def line_of(path):
    js = [l for l in open(path) if l.startswith("{")]
    return json.loads(js[-1])


# ---- kernel-trace summary
for a, b in [("stats/run_kernel_stats.csv", "kernel_stats.csv"), ("stats/run_domain_stats.csv", "domain_stats.csv")]:
    shutil.copy(os.path.join(G, a), os.path.join(P, pre + b))
stats = {}
for r in csv.DictReader(open(os.path.join(P, pre + "kernel_stats.csv"))):
    stats[kname(r["Name"])] = r
rows = sorted(csv.DictReader(open(os.path.join(G, "stats/run_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
with open(os.path.join(P, pre + "kernel_trace_match_and_fitness.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Dispatch_Id", "Kernel", "Grid_Size_X", "Workgroup_Size_X", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count",
                "Start_Timestamp", "End_Timestamp", "Duration_us"])
    for r in rows:
        k = kname(r["Kernel_Name"])
        if k.startswith("ndt_align") or k.startswith("fitness_"):
            w.writerow([r["Dispatch_Id"], k, r["Grid_Size_X"], r["Workgroup_Size_X"], r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""),
                        r.get("VGPR_Count", ""), r.get("SGPR_Count", ""), r["Start_Timestamp"], r["End_Timestamp"],
                        "%.3f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)])

# ---- counters: mean over the match kernel's dispatches of every counter
keep = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
        "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
vals = collections.defaultdict(list)
for d in ("pmc_fetch", "pmc_write", "pmc_valu", "pmc_lds"):
    src = os.path.join(G, d, "run_counter_collection.csv")
    if not os.path.exists(src):
        continue
    rws = list(csv.DictReader(open(src)))
    with open(os.path.join(P, pre + d + ".csv"), "w", newline="") as f:
        w = csv.DictWriter(f, keep); w.writeheader()
        for r in rws:
            k = kname(r["Kernel_Name"])
            if not (k.startswith("ndt_align") or k.startswith("fitness_")):
                continue
            r = {c: r[c] for c in keep}; r["Kernel_Name"] = k; w.writerow(r)
            if k.startswith("ndt_align"):
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in vals.items()}

for n, out in [("stats.log", "stats.log"), ("bench.log", "bench.json")] + [(d + ".log", d + ".log") for d in ("pmc_fetch", "pmc_write", "pmc_valu", "pmc_lds")]:
    if os.path.exists(os.path.join(G, n)):
        open(os.path.join(P, pre + out), "w").write(json.dumps(line_of(os.path.join(G, n))) + "\n")
bench = line_of(os.path.join(G, "bench.log"))
workload = bench["config"]["workload"]

traffic = {
    "kernel": "ndt_align_kernel", "workload": workload,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, mean over the match kernel's launches of "
              "`bench.py --no-single-scan --no-cpu-baseline --steps 8` (profiles/%spmc_fetch.csv, %spmc_write.csv)" % (pre, pre),
    "fetch_size_kb": mean["FETCH_SIZE"], "write_size_kb": mean["WRITE_SIZE"],
    "bytes_per_launch": (2.0 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0,
    "note": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (rocprofv3 reports KB).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads half "
            "the bytes of wide coalesced streaming reads and other widths are to be calibrated: tools/repro/calib_fetch.hip "
            "(profiles/r02_calib_fetch.txt) reads a 64 MiB buffer with this kernel's access widths -- 4, 8, 16 bytes per lane, "
            "48-byte records, 8-byte gathers one line apart -- and FETCH_SIZE is 0.500 x the bytes of the 128-byte lines "
            "touched in every case, so the fetch counter is doubled; WRITE_SIZE is exact per the guide."}
json.dump(traffic, open(os.path.join(P, pre + "traffic.json"), "w"), indent=1)

if "SQ_INSTS_VALU" in mean:
    pe = bench["roofline"]["valu"]["point_evals_per_launch"]
    simd_quads = mean["SQ_WAVE_CYCLES"] / 4.0             # 4 waves per SIMD: wave-resident quad-cycles -> SIMD quad-cycles
    valu = {
        "kernel": "ndt_align_kernel", "workload": workload,
        "source": "rocprofv3 --pmc (two passes: profiles/%spmc_valu.csv, %spmc_lds.csv), mean over the match kernel's launches; "
                  "SQ cycle counters are in quad-cycles" % (pre, pre),
        "counters": {k: mean[k] for k in sorted(mean) if k.startswith("SQ_")},
        "insts_per_point_eval": mean["SQ_INSTS_VALU"] / (pe / 64.0),     # VALU instructions a lane executes per point-evaluation
        "valu_wave_insts_per_64_points": mean["SQ_INSTS_VALU"] / (pe / 64.0),
        "salu_wave_insts_per_64_points": mean["SQ_INSTS_SALU"] / (pe / 64.0),
        "valu_util": mean["SQ_ACTIVE_INST_VALU"] / simd_quads,
        "valu_lane_util": mean["SQ_THREAD_CYCLES_VALU"] / (mean["SQ_ACTIVE_INST_VALU"] * 64.0),
        "note": "instruction counts cover the whole kernel (window staging, scan ordering, work sharing, the optimiser), divided by the "
                "point-evaluations of the launch; valu_util = SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / 4): the share of a SIMD's "
                "time (while its four waves are resident) in which the vector ALU executes; valu_lane_util = active lanes per "
                "executed VALU instruction / 64"}
    if "SQ_INSTS_LDS" in mean:
        valu["lds_wave_insts_per_64_points"] = mean["SQ_INSTS_LDS"] / (pe / 64.0)
        valu["lds_bank_conflict_share_of_lds_time"] = mean["SQ_LDS_BANK_CONFLICT"] / (mean["SQ_ACTIVE_INST_LDS"] * 4.0)
        valu["wave_time_waiting"] = mean["SQ_WAIT_ANY"] / mean["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in mean else None
    json.dump(valu, open(os.path.join(P, pre + "valu.json"), "w"), indent=1)

k = stats["ndt_align_kernel"]
print("match kernel: calls %s avg %.1f us | fetch %.0f MB write %.0f MB" % (
    k["Calls"], float(k["AverageNs"]) / 1e3, mean["FETCH_SIZE"] * 1024 / 1e6, mean["WRITE_SIZE"] * 1024 / 1e6))
for n in ("stats.log", "bench.json"):
    d = json.load(open(os.path.join(P, pre + n)))
    print(n, round(d["value"]), "ms/step %.4f match kernel_ms %.4f fitness %.4f frac %.4f map %.4f" % (
        d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["fitness"]["ms"], d["roofline"]["frac"], d["map_build_ms"]))
for name, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
    print("  %-34s calls %4s avg %8.1f us" % (name, r["Calls"], float(r["AverageNs"]) / 1e3))
