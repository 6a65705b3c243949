#!/usr/bin/env python3
"""bench.py -- scan-matches/sec of the NDT hot path on MI355X (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[2] per GPU -- a batch of 256 scans (10k points
each) against a shared 1M-point NDT map at 0.5 m voxels; with --gpus N every rank holds its own
256-scan shard of configs[3]'s batch (weak scaling, no data-path collective; results are gathered
to rank 0 over RCCL).  configs[1] (one scan) is the same kernel at B = 1 and is reported as
`single_scan_ms`.

One step = the whole hot path over one batch with inputs resident in HBM: voxel
normal-distributions build of the map (once per batch; the reference rebuilds it on every
estimatePose call, src/PoseEstimator.cpp:19) + all 256 full optimisations to convergence +
fitness scores + final Hessians, then the gather of the 256 result records.

Steps are pipelined the way a caller with a stream of batches would run them: two map buffers, the
rebuild for step i + 1 is queued on a second stream and runs while the matches of step i finish
(helper workgroups that can get no more work leave their CUs), the match launches stay in order on
one stream and each waits for its own map's build.  `map_build_ms` is the build alone (un-overlapped),
`map_build_in_step_ms` what the build's stream shows inside the pipelined loop.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(res, n_pts):
    """per point-evaluation 8 B (float2 point) + Kbar x 20 B (mu + Sigma^-1 as
    float32 equivalents); per match E x N x (8 + 20 Kbar) + N x 16 for the fitness pass."""
    ev = res["evals"].astype(np.float64)
    kb = res["kbar"].astype(np.float64)
    return float(np.sum(ev * n_pts * (8.0 + 20.0 * kb) + n_pts * 16.0))


def measured_traffic():
    """HBM bytes per launch of the match kernel from the PMC passes committed under profiles/
    (FETCH_SIZE + WRITE_SIZE, collected separately with rocprofv3 -- they cannot be read live here);
    None when no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        t = json.load(f)
    return float(t["bytes_per_launch"]), os.path.relpath(files[-1], ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="scans per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-scan", action="store_true", help="skip the configs[1] latency launches (B = 1)")
    ap.add_argument("--cpu-sample", type=int, default=256, help="scans timed on the host cores")
    ap.add_argument("--cpu-reps", type=int, default=4, help="times the CPU sample is run (about 10 s of CPU work in all)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ndt_slam_amd import capi, shard, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the NDT core has no CPU fallback")
    # NDT_BENCH_REHEARSAL=1: every rank on device 0 over gloo -- only to walk the N > 1 code path on a
    # one-GPU box (RCCL refuses two ranks on one device); its numbers mean nothing.
    rehearsal = world > 1 and os.environ.get("NDT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    cfg = synth.CONFIGS["C3"]
    B, n_scan = args.batch, cfg["n_scan"]
    # synthetic inputs (no reference data exists): same map on every rank, own scan shard
    map_xy = synth.make_map(cfg["n_map"], cfg["half"])
    sf = synth.ScanFactory(map_xy, cfg["half"], n_scan)
    scans, off, truths, inits = sf.batch(rank * B, B)

    ctx = capi.Context(local_rank)
    # a dedicated (non-null) torch stream: map build, matches, events and the gather are all
    # ordered on it (the C ABI reads a NULL stream argument as "the context's own stream")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx.set_stream(stream.cuda_stream)
    prm = capi.default_params(resolution=cfg["resolution"])     # otherwise ndt_mapping.launch:32-36
    d_map = torch.from_numpy(map_xy).to(dev)
    d_scans = torch.from_numpy(scans).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_init = torch.from_numpy(inits).to(dev)
    d_res2 = [torch.zeros(B * capi.RESULT_BYTES, dtype=torch.uint8, device=dev) for _ in range(2)]
    d_res = d_res2[0]
    # the gather of step i runs on a side stream while step i + 1 is already computing
    side = torch.cuda.Stream(device=dev) if world > 1 else None
    ev_done = [torch.cuda.Event() for _ in range(2)]
    gathered = [None, None]
    torch.cuda.synchronize()
    # Two voxel grids, built by a second context on its own stream: the rebuild of step i + 1 runs while
    # the matches of step i finish (their last workgroups leave CUs free), the matches themselves stay in
    # order on `stream`.  Every step still rebuilds its map and then matches against it.
    bstream = torch.cuda.Stream(device=dev)
    bctx = capi.Context(local_rank)
    bctx.set_stream(bstream.cuda_stream)
    gmaps = [capi.Map(bctx, params=prm, dev_ptr=d_map.data_ptr(), n=len(map_xy), stride=8) for _ in range(2)]
    gmap = gmaps[0]
    torch.cuda.synchronize()
    solo_build_ms = []
    for _ in range(3):                                      # the build alone, nothing else on the GPU
        gmaps[1].rebuild(dev_ptr=d_map.data_ptr(), n=len(map_xy), stride=8)
        solo_build_ms.append(bctx.last_timing()[0])
    torch.cuda.synchronize()

    ev_a = [torch.cuda.Event(enable_timing=True) for _ in range(2 * (args.steps + args.warmup))]
    ev_m = [torch.cuda.Event(enable_timing=True) for _ in range(2 * (args.steps + args.warmup))]

    def step(i):
        gm = gmaps[i & 1]
        # a2: rebuild the voxel grid of this step in place, as soon as the matches of step i - 2 (the last
        # readers of this grid) are done
        if i >= 2:
            bstream.wait_event(ev_a[2 * (i - 2) + 1])
        ev_m[2 * i].record(bstream)
        gm.rebuild(dev_ptr=d_map.data_ptr(), n=len(map_xy), stride=8)
        ev_m[2 * i + 1].record(bstream)
        # a3-a9 for the whole batch: one launch on `stream`, after this step's build (the library waits
        # for it too; waiting here keeps that wait out of the kernel's event interval)
        stream.wait_event(ev_m[2 * i + 1])
        out = d_res2[i & 1]
        if world > 1:
            stream.wait_event(ev_done[i & 1])          # the gather that last read this result buffer has finished
        ev_a[2 * i].record(stream)
        gm.align_batch_dev(d_scans.data_ptr(), d_off.data_ptr(), B, len(scans), d_init.data_ptr(),
                           out.data_ptr(), stream=stream.cuda_stream)
        ev_a[2 * i + 1].record(stream)                 # (also what the rebuild of step i + 2 waits for)
        if world > 1:    # gather of poses (the only collective on this path)
            side.wait_stream(stream)
            with torch.cuda.stream(side):
                gathered[i & 1] = shard.gather_results(out, dst=0)
                ev_done[i & 1].record(side)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kern_ms = [ev_a[2 * i].elapsed_time(ev_a[2 * i + 1]) for i in range(args.warmup, args.warmup + args.steps)]
    map_ms = [ev_m[2 * i].elapsed_time(ev_m[2 * i + 1]) for i in range(args.warmup, args.warmup + args.steps)]
    last = (args.warmup + args.steps - 1) & 1
    res = np.frombuffer(d_res2[last].cpu().numpy().tobytes(), dtype=capi.RESULT_DTYPE)
    assert np.all(res["status"] == 0)

    out = None
    if rank == 0:
        avg_kern_ms = float(np.mean(kern_ms))
        traffic, traffic_src = measured_traffic()
        alg_bytes = algorithmic_bytes(res, n_scan)
        achieved = alg_bytes / (avg_kern_ms * 1e-3) / 1e9
        err = res["pose"] - truths
        err[:, 2] = (err[:, 2] + math.pi) % (2 * math.pi) - math.pi
        out = {
            "metric": "scan-matches/sec (10k-pt scan vs 1M-pt NDT map)",
            "value": world * B * args.steps / elapsed,
            "unit": "matches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: batch of %d scans x %d pts vs shared %d-pt map, 0.5 m "
                                   "voxels, per GPU (configs[3] sharding at N>1); map rebuilt every step (the rebuild of step i+1 overlaps the end of step i's matches)"
                                   % (B, n_scan, cfg["n_map"]),
                       "scans_per_gpu": B, "scan_points": n_scan, "map_points": cfg["n_map"],
                       "resolution": cfg["resolution"], "parallelism": "scan-shards x%d, gather of results" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "note": "achieved = SURVEY 8d algorithmic bytes / kernel time.  The kernel is VALU-issue bound, "
                                 "not HBM bound: each voxel record is staged once per match in LDS, so the measured HBM "
                                 "traffic is below the algorithmic bytes (DESIGN.md 4.7)",
                         "kernel": "ndt_align_kernel", "kernel_ms": avg_kern_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "mean_evals": float(res["evals"].mean()), "max_evals": int(res["evals"].max()),
                         "mean_kbar": float(res["kbar"].mean())},
            "map_build_ms": float(np.median(solo_build_ms)), "map_build_in_step_ms": float(np.mean(map_ms)),
            "converged": int(res["converged"].sum()),
            "median_abs_err_m": float(np.median(np.hypot(err[:, 0], err[:, 1]))),
        }

    # configs[1]: one scan (latency of a single match, same kernel at B = 1)
    if rank == 0 and world == 1 and not args.no_single_scan:
        one = torch.zeros(capi.RESULT_BYTES, dtype=torch.uint8, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record(stream)
            gmap.align_batch_dev(d_scans.data_ptr(), d_off.data_ptr(), 1, int(off[1]), d_init.data_ptr(),
                                 one.data_ptr(), stream=stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out["single_scan_ms"] = float(np.median(ts))

    # Row f1 (source pre-filter, pcl::ApproximateVoxelGrid): raw scans 3x oversampled -> filtered scans,
    # all on the device; reported beside the headline metric, not part of it (the 10k-pt scans of the
    # metric are post-filter clouds by definition).
    if rank == 0 and world == 1 and not args.no_single_scan:
        rng = np.random.default_rng(11)
        raw = np.repeat(scans, 3, axis=0) + rng.normal(0, 0.004, (3 * len(scans), 2)).astype(np.float32)
        raw_off = (off.astype(np.int64) * 3)
        d_raw = torch.from_numpy(raw).to(dev); d_roff = torch.from_numpy(raw_off).to(dev)
        d_f = torch.empty_like(d_raw); d_foff = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record(stream)
            ctx.prefilter_batch_dev(d_raw.data_ptr(), 8, d_roff.data_ptr(), B, len(raw), 0.05, d_f.data_ptr(),
                                    d_foff.data_ptr(), stream=stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        n_out = int(d_foff[-1].item())
        ms = float(np.median(ts))
        out["prefilter"] = {"scans": B, "raw_points": int(len(raw)), "filtered_points": n_out, "leaf": 0.05, "ms": ms,
                            "raw_points_per_s": len(raw) / (ms * 1e-3),
                            "algorithmic_GBps": (len(raw) + n_out) * 8 / (ms * 1e-3) / 1e9}
        # a whole front-end step for the batch without leaving the device (rows f2 + f1 + a2 + a3-a9 + f2):
        # odometry prediction -> pre-filter -> map rebuild -> matches -> EKF fusion
        pred0 = np.column_stack([inits[:, 0], inits[:, 1], np.degrees(inits[:, 2])])
        d_last = torch.from_numpy(pred0).to(dev); d_prevo = torch.zeros(B, 3, dtype=torch.float64, device=dev)
        d_curo = torch.zeros_like(d_prevo)                       # zero odometry motion: prediction = last pose
        d_mo = torch.zeros_like(d_prevo); d_pred = torch.zeros_like(d_prevo); d_in2 = torch.zeros_like(d_prevo)
        d_lc = torch.from_numpy(np.tile(np.eye(3).ravel() * 1e-4, (B, 1))).to(dev)
        d_fu = torch.zeros_like(d_prevo); d_cv = torch.zeros(B, 9, dtype=torch.float64, device=dev)
        d_ok = torch.zeros(B, dtype=torch.int32, device=dev)
        fprm = capi.default_fuse_params(score_thre=0.5)
        ts = []
        for _ in range(5):
            e0.record(stream)
            ctx.predict_batch_dev(d_curo.data_ptr(), d_prevo.data_ptr(), d_last.data_ptr(), B, d_mo.data_ptr(),
                                  d_pred.data_ptr(), d_in2.data_ptr(), stream=stream.cuda_stream)
            ctx.prefilter_batch_dev(d_raw.data_ptr(), 8, d_roff.data_ptr(), B, len(raw), 0.05, d_f.data_ptr(),
                                    d_foff.data_ptr(), stream=stream.cuda_stream)
            gmap.rebuild(dev_ptr=d_map.data_ptr(), n=len(map_xy), stride=8)
            gmap.align_batch_dev(d_f.data_ptr(), d_foff.data_ptr(), B, len(raw), d_in2.data_ptr(), d_res2[0].data_ptr(),
                                 stream=stream.cuda_stream)
            ctx.fuse_batch_dev(d_res2[0].data_ptr(), d_pred.data_ptr(), d_mo.data_ptr(), d_last.data_ptr(), d_lc.data_ptr(), B,
                               fprm, d_fu.data_ptr(), d_cv.data_ptr(), d_ok.data_ptr(), stream=stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out["front_end_step"] = {"stages": "predict + pre-filter + map rebuild + match + fuse, all on the device",
                                 "scans": B, "raw_points_per_scan": int(len(raw) // B), "ms": float(np.median(ts)),
                                 "scans_per_s": B / (float(np.median(ts)) * 1e-3), "accepted": int(d_ok.sum().item())}

    # Row f3 (local-map assembly, Submap::makeMap with moving-object removal): a submap of 12 registered scans
    # of the metric's size (walls seen again by every scan + an object that moves, synth.submap_scans) assembled
    # on the device; the oracle's literal octree does the same on one host core (checker and CPU figure).
    # Reported beside the headline metric, not part of it.
    if rank == 0 and world == 1 and not args.no_single_scan:
        ns = 12
        reg = synth.submap_scans(ns, cfg["n_scan"])
        reg_off = np.zeros(ns + 1, np.uint64)
        reg_off[1:] = np.cumsum([len(r) for r in reg])
        d_reg = torch.from_numpy(np.concatenate(reg)).to(dev)
        d_lm = torch.empty((len(d_reg) + 1, 2), dtype=torch.float32, device=dev)
        d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(6):
            e0.record(stream)
            ctx.make_map_dev(d_reg.data_ptr(), 8, reg_off, True, True, True, 0.05, 0.1, d_lm.data_ptr(), d_cnt.data_ptr(),
                             stream=stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        n_lm = int(d_cnt.item())
        lm = {"scans": ns, "points": int(len(d_reg)), "kept": n_lm, "resol": 0.05, "thre_neighbor": 0.1,
              "ms": float(np.median(ts[1:]))}
        if not args.no_cpu_baseline:
            from oracle import ndt_oracle as O
            t = time.perf_counter()
            ref_lm = O.make_map(reg, True, True, True, 0.05, 0.1)
            lm["cpu_ms_1core"] = (time.perf_counter() - t) * 1e3
            lm["identical"] = bool(n_lm == len(ref_lm) and d_lm[:n_lm].cpu().numpy().tobytes() == ref_lm.tobytes())
        out["local_map"] = lm

    # CPU baseline: the oracle (a port -- PCL itself is absent) on this box's host cores,
    # rank 0 at N = 1 only, on a bounded sample of the same batch.
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ndt_oracle as O
        ns = min(args.cpu_sample, B)
        t = time.perf_counter()
        om = O.Map(map_xy, O.default_params(resolution=cfg["resolution"]))
        t_build = time.perf_counter() - t
        sub_off = off[:ns + 1]
        t = time.perf_counter()
        for _ in range(max(1, args.cpu_reps)):
            ref = om.align_batch(scans[:int(sub_off[-1])], sub_off, inits[:ns], nthreads=1)
        t_align = (time.perf_counter() - t) / max(1, args.cpu_reps)
        ncpu = min(16, os.cpu_count() or 1)     # the box's CPU share for one GPU
        t = time.perf_counter()
        om.align_batch(scans[:int(sub_off[-1])], sub_off, inits[:ns], nthreads=ncpu)
        t_all = time.perf_counter() - t
        d = res["pose"][:ns] - ref["pose"]
        d[:, 2] = (d[:, 2] + math.pi) % (2 * math.pi) - math.pi
        out["cpu_baseline"] = {
            "value": ns / t_align, "unit": "matches/s", "cores": 1, "kind": "port",
            "sample": "first %d of the %d scans x %d repetitions, 1 thread, map built once (amortised); oracle/ndt_oracle.c"
                      % (ns, B, max(1, args.cpu_reps)),
            "map_build_s": t_build,
            "reference_faithful_matches_per_s": 1.0 / (t_build + t_align / ns),
            "all_cores": {"value": ns / t_all, "cores": ncpu},
        }
        out["parity"] = {"max_dpos_m": float(np.abs(d[:, :2]).max()), "max_dyaw_rad": float(np.abs(d[:, 2]).max()),
                         "same_iters": bool(np.all(res["iters"][:ns] == ref["iters"])), "sample": ns}
        out["gpu_over_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
