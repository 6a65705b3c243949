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

# ---- round 5: the other kernels that carry a `frac` in the bench line, profiled on their own (tools/gpu_profile.sh parts i-iii)
def mean_by_kernel(path, counter):
    """mean per dispatch of one counter for every kernel of a rocprofv3 --pmc run"""
    acc = collections.defaultdict(list)
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def traffic_json(name, kernels_note, fetch, write, per_what, src):
    kb_f = sum(fetch.values()); kb_w = sum(write.values())
    json.dump({"kernels": kernels_note, "per": per_what, "source": src,
               "fetch_size_kb": kb_f, "write_size_kb": kb_w, "bytes_per_launch": (2.0 * kb_f + kb_w) * 1024.0,
               "by_kernel_kb": {k: {"fetch": fetch.get(k, 0.0), "write": write.get(k, 0.0)} for k in sorted(set(fetch) | set(write))},
               "note": "sum over the chain's kernels of their mean per dispatch; (2 x FETCH_SIZE + WRITE_SIZE) x 1024, calib_fetch as in "
                       "%straffic.json" % pre}, open(os.path.join(P, pre + name), "w"), indent=1)


# fitness kernels: their counters come with the bench's own --pmc passes (a counter pass serialises the kernels: per-dispatch
# values are each kernel's own), their time ALONE from the synchronous loop (alone_stats)
ff = {k: v for k, v in mean_by_kernel(os.path.join(G, "pmc_fetch", "run_counter_collection.csv"), "FETCH_SIZE").items() if k.startswith("fitness_")}
fw = {k: v for k, v in mean_by_kernel(os.path.join(G, "pmc_write", "run_counter_collection.csv"), "WRITE_SIZE").items() if k.startswith("fitness_")}
if ff:
    traffic_json("fitness_traffic.json", "fitness_points_kernel + fitness_reduce_kernel", ff, fw, "launch of 256 matches",
                 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the bench (profiles/%spmc_fetch.csv, %spmc_write.csv)" % (pre, pre))
for part, out in (("alone_stats", "alone_kernel_stats.csv"), ("build_stats", "build_kernel_stats.csv"), ("c5_stats", "c5_kernel_stats.csv")):
    src = os.path.join(G, part, "run_kernel_stats.csv")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, pre + out))
bf = {k: v for k, v in mean_by_kernel(os.path.join(G, "build_fetch", "run_counter_collection.csv"), "FETCH_SIZE").items() if k.startswith(("map_", "scan_", "fill_"))}
bw = {k: v for k, v in mean_by_kernel(os.path.join(G, "build_write", "run_counter_collection.csv"), "WRITE_SIZE").items() if k.startswith(("map_", "scan_", "fill_"))}
if bf:
    traffic_json("build_traffic.json", "the chain of ndt_map_build_dev", bf, bw, "build of the 1M-point map",
                 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes -- python3 tools/prof_build.py 10 C3 (a build-only loop)")
for n in ("c5_bench.log", "c5_stats.log"):
    if os.path.exists(os.path.join(G, n)):
        open(os.path.join(P, pre + n.replace(".log", ".json")), "w").write(json.dumps(line_of(os.path.join(G, n))) + "\n")
c5f = {k: v for k, v in mean_by_kernel(os.path.join(G, "c5_fetch", "run_counter_collection.csv"), "FETCH_SIZE").items() if k.startswith("fitness_")}
c5w = {k: v for k, v in mean_by_kernel(os.path.join(G, "c5_write", "run_counter_collection.csv"), "WRITE_SIZE").items() if k.startswith("fitness_")}
if c5f:
    traffic_json("c5_fitness_traffic.json", "fitness_points_kernel + fitness_far_kernel + fitness_reduce_kernel (configs[4])", c5f, c5w,
                 "launch of 512 seeds", "rocprofv3 --pmc passes of `bench.py --config C5 --no-cpu-baseline --no-single-scan --steps 6`")
c5v = os.path.join(G, "c5_valu", "run_counter_collection.csv")
if os.path.exists(c5v):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(c5v)):
        per[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    outv = {}
    for k, c in per.items():
        if not (k.startswith("fitness_") or k.startswith("ndt_align")):
            continue
        mk = {n: sum(v) / len(v) for n, v in c.items()}
        outv[k] = {"counters": mk, "valu_lane_util": mk["SQ_THREAD_CYCLES_VALU"] / (mk["SQ_ACTIVE_INST_VALU"] * 64.0) if mk.get("SQ_ACTIVE_INST_VALU") else None,
                   "lanes_of_64": mk["SQ_THREAD_CYCLES_VALU"] / mk["SQ_ACTIVE_INST_VALU"] if mk.get("SQ_ACTIVE_INST_VALU") else None}
    json.dump({"workload": "bench.py --config C5 (512 seeds x one 10k-pt scan vs the 5M-point map)", "kernels": outv},
              open(os.path.join(P, pre + "c5_valu.json"), "w"), indent=1)
vm = os.path.join(G, "pmc_vmem", "run_counter_collection.csv")
if os.path.exists(vm):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(vm)):
        per[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    json.dump({k: {n: sum(v) / len(v) for n, v in c.items()} for k, c in per.items() if k.startswith(("ndt_align", "fitness_"))},
              open(os.path.join(P, pre + "vmem_insts.json"), "w"), indent=1)

k = stats["ndt_align_kernel"]
print("match kernel: calls %s avg %.1f us | fetch %.0f MB write %.0f MB" % (
    k["Calls"], float(k["AverageNs"]) / 1e3, mean["FETCH_SIZE"] * 1024 / 1e6, mean["WRITE_SIZE"] * 1024 / 1e6))
for n in ("stats.log", "bench.json"):
    d = json.load(open(os.path.join(P, pre + n)))
    print(n, round(d["value"]), "ms/step %.4f match kernel_ms %.4f fitness %.4f frac %.4f map %.4f" % (
        d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["fitness"]["ms"], d["roofline"]["frac"], d["map_build_ms"]))
for name, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
    print("  %-34s calls %4s avg %8.1f us" % (name, r["Calls"], float(r["AverageNs"]) / 1e3))
