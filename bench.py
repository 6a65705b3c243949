#!/usr/bin/env python3
"""bench.py -- scan-matches/sec of the NDT hot path on MI355X (BASELINE.json metric).

Workload (config.workload), default `--config C3`: BASELINE.json configs[2] per GPU -- a batch of 256 scans (10k
points each) against a shared 1M-point NDT map at 0.5 m voxels.  With --gpus N this is configs[3]'s sharding: rank 0
holds the N x 256-scan batch and fans the shards out over RCCL (grouped isend / irecv, device to device), every rank
matches its shard, the result records are gathered to rank 0 (weak scaling; no collective on the data path).
`--config C5`: BASELINE.json configs[4] per GPU -- multi-hypothesis relocalisation, 512 seed poses x one 10k-point
scan against the 5M-point map (x 8 GPUs = the 4096 seeds of configs[4]); the scan is broadcast, the seeds are
sharded, the best hypothesis is an arg-max over all ranks.  The default run reports C5 as a side figure
(`multi_hypothesis`), configs[1] (one scan) as `single_scan_ms`.

One step = the whole hot path over one batch with inputs resident in HBM: voxel normal-distributions build of the
map (the reference rebuilds it on every estimatePose call, src/PoseEstimator.cpp:19) + all full optimisations to
convergence + fitness scores + final Hessians, then the gather of the result records.

Steps are pipelined the way a caller with a stream of batches would run them: two map buffers, the rebuild for step
i + 1 is queued on a second stream and runs while the matches of step i finish; with `--inflight 2` consecutive match
launches alternate between two contexts / streams, so the owners of step i + 1 start on the CUs the helpers of step i
have left.  `roofline.kernel_ms` is always the duration of ONE launch (HIP events on that launch's stream).

`python bench.py --gpus N` is one command: without WORLD_SIZE in the environment the process starts the N ranks itself
(`python -m torch.distributed.run`, one rank per GPU) before it has touched the GPU, relays rank 0's line and exits with
the ranks' status; under a launcher (WORLD_SIZE set) it is a rank.

Layout of this file: parse_args / spawn_ranks (the launcher) -- init_env, make_inputs, Pipeline (set-up) -- run_timed,
kernel_figures (the timed region and what the library's events say about it) -- leg_* (side figures behind the timed
region) -- headline (the JSON line) -- main.
"""
import argparse
import datetime
import json
import math
import os
import platform
import sys
import time
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6   # half the 157.3 TF fp32 vector rate (MI355X_MICROARCH.md chip table)
SEED_SHARDS = 8           # configs[4]: 4096 seeds = 8 shards of 512; rank r takes seeds[r::8]


# ------------------------------------------------------------------------------------------------ bookkeeping helpers
def algorithmic_bytes(res, n_pts, fitness=True):
    """per point-evaluation 8 B (float2 point) + Kbar x 20 B (mu + Sigma^-1 as
    float32 equivalents); per match E x N x (8 + 20 Kbar) + N x 16 for the fitness pass.
    The fitness term belongs to the fitness kernels (fitness=False: the match kernel alone)."""
    ev = res["evals"].astype(np.float64)
    kb = res["kbar"].astype(np.float64)
    return float(np.sum(ev * n_pts * (8.0 + 20.0 * kb) + (n_pts * 16.0 if fitness else 0.0)))


def _latest(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def measured_traffic(pattern="r[0-9][0-9]_traffic.json"):
    """HBM bytes per launch of the match kernel from the PMC passes committed under profiles/
    (FETCH_SIZE + WRITE_SIZE, collected separately with rocprofv3 -- they cannot be read live here);
    None when no summary is present."""
    f = _latest(pattern)
    if not f:
        return None, None
    with open(f) as fh:
        t = json.load(fh)
    return float(t["bytes_per_launch"]), os.path.relpath(f, ROOT)


def side_traffic(kind):
    """HBM bytes per launch of the map build's / the fitness kernels' chain from profiles/r*_<kind>_traffic.json
    (tools/gpu_profile.sh: a build-only / fitness-only loop under rocprofv3 --pmc); (None, None) when absent."""
    return measured_traffic("r[0-9][0-9]_%s_traffic.json" % kind)


def measured_valu():
    """SQ counter summary of the match kernel (profiles/r*_valu.json, made by tools/save_profiles.py from
    rocprofv3 --pmc passes): instructions per point-evaluation, VALU utilisation; None when absent."""
    cands = [f for f in sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_valu.json")))
             if "_c5_" not in f and "_build_" not in f and "_fitness_" not in f]
    if not cands:
        return None
    with open(cands[-1]) as fh:
        v = json.load(fh)
    v["source"] = os.path.relpath(cands[-1], ROOT)
    return v


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


# ------------------------------------------------------------------------------------------------ arguments, launcher
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["C3", "C5"], default="C3",
                    help="C3: 256 scans x 10k vs 1M map per GPU (configs[2]/[3]); C5: 512 seeds x one scan vs 5M map per GPU (configs[4])")
    ap.add_argument("--batch", type=int, default=None, help="matches per GPU and step (default 256 for C3, 512 for C5)")
    ap.add_argument("--inflight", type=int, default=1, choices=[1, 2, 3, 4], help="match launches in flight (k: k contexts / streams in turn, k + 1 map buffers)")
    ap.add_argument("--workgroups", type=int, default=0, help="workgroups per match launch (0 = one per CU); fewer leave CUs to the map build's stream")
    ap.add_argument("--max-helpers", type=int, default=-1, help="helper workgroups per unfinished scan (default -1: the library chooses, 2 for whole-GPU batches and 8 below); fewer free CUs earlier for whatever is queued behind the launch")
    ap.add_argument("--defer-fitness", type=int, default=0, choices=[0, 1], help="NDT_OPT_DEFER_FITNESS on the match contexts (and one more map buffer): the fitness kernels of a launch on the context's own stream, beside the next launch's match kernel.  An experiment, not the headline: LOG R5.14")
    ap.add_argument("--prepare", choices=["off", "build-stream", "own-stream"], default="off",
                    help="prepare batch i ahead of its launch (ndt_align_batch_prepare_dev: optimiser start, window geometry and voxel order as a kernel of its own): behind the step's rebuild on the build stream, or on a stream of its own queued first.  Off by default: the match kernel is 18 us shorter with it (roofline.frac 0.30 -> 0.32) but the step is not -- the order kernel's 1024-thread workgroups need whole CUs and only get them when the fitness kernels of the step before are through (LOG R5.2)")
    ap.add_argument("--time-builds", action="store_true", help="extra events around the map build and around the whole launch inside the step loop (launch_interval_ms, map_build_in_step_ms)")
    ap.add_argument("--no-scatter", action="store_true", help="N > 1: every rank generates its own shard instead of receiving it from rank 0")
    ap.add_argument("--map-from-rank0", action="store_true", help="N > 1: only rank 0 generates the target cloud, the others build their map from shard.broadcast_map's copy (by default every rank generates it AND the broadcast is timed and checked against it: comm.broadcast_map_ms)")
    ap.add_argument("--moving-steps", type=int, default=48, help="steps of the moving-local-map side leg (0: skip)")
    ap.add_argument("--moving-margin", type=int, default=8, help="ndt_params::grid_margin of the second moving-local-map leg (voxels)")
    ap.add_argument("--moving-every", type=int, default=8, help="moving-local-map leg: the cloud's voxel bounding box moves every k-th step")
    ap.add_argument("--steady-steps", type=int, default=200, help="steps of the steady-state side figure (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--libm-f32", type=int, default=None, choices=[0, 1], help="override ndt_params.libm_f32 of the preset (A/B of the float32 cos / sin model)")
    ap.add_argument("--no-single-scan", action="store_true", help="skip the side figures (configs[1] latency, C5 leg, rows f1-f3)")
    ap.add_argument("--cpu-sample", type=int, default=256, help="matches timed on the host cores")
    ap.add_argument("--cpu-reps", type=int, default=5, help="times the CPU sample is run (median reported)")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process -- one
    `python -m torch.distributed.run` (one rank per GPU, rendezvous on 127.0.0.1, a free port) -- relay rank 0's JSON line to
    stdout (everything else the ranks print goes to stderr) and return the children's status.  This process never imports
    torch.cuda, creates no HIP context and never exec()s: it only waits.  (src/ScanMatcher.cpp:40,45 is the loop whose
    iterations the ranks share out.)"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: what RCCL needs on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        (sys.stdout if line.startswith("{") else sys.stderr).flush()
    return proc.wait()


# ------------------------------------------------------------------------------------------------ set-up
def init_env(args):
    """Rank, device and process group of this process (one rank per GPU; backend nccl = RCCL over xGMI)."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d (a launcher started %d rank(s); run `python bench.py --gpus %d` on its own "
                         "and it starts them itself)" % (world, args.gpus, world, args.gpus))
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
        # a bounded timeout: a rank that drops out of a collective turns into an error on the others, not a hang
        tmo = datetime.timedelta(seconds=300)
        if rehearsal:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
    env = SimpleNamespace(rank=rank, local_rank=local_rank, world=world, dev=dev, rehearsal=rehearsal,
                          # point-to-point / collective payloads: device tensors over RCCL, host tensors in the gloo rehearsal
                          comm_dev=torch.device("cpu") if rehearsal else dev, comm={})

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    env.fence = fence
    return env


def fan_out_map(env, args, cfg):
    """The target cloud.  Every rank BUILDS its own cell table ("build redundantly from a broadcast of raw
    points"); where the raw points come from at N > 1: by default every rank generates the same synthetic cloud and the
    fan-out a caller with ONE PointCloudMap needs (src/ScanMatcher.cpp:40) -- shard.broadcast_map, chunked -- is timed
    and checked against it byte for byte; with --map-from-rank0 only rank 0 generates and the others use the copy."""
    import torch
    import torch.distributed as dist
    from ndt_slam_amd import shard, synth
    map_xy = synth.make_map(cfg["n_map"], cfg["half"]) if (env.rank == 0 or not args.map_from_rank0) else None
    if env.world == 1:
        return map_xy
    comm = env.comm
    # Everything that can fail LOCALLY (copies, the comparison) sits between collectives that every rank reaches: a rank's
    # own failure travels in the flag of the closing all-reduce instead of leaving the others blocked in it.  (A collective
    # that itself fails on one rank is bounded by the process group's timeout, init_env.)
    got, ok, err = None, 1, None
    try:
        env.fence()
        t0 = time.perf_counter()
        t_map = shard.broadcast_map(map_xy if env.rank == 0 else None, src=0, device=env.comm_dev)
        env.fence()
        comm["broadcast_map_ms"] = (time.perf_counter() - t0) * 1e3
        comm["broadcast_map_bytes"] = int(t_map.numel() * 4)
    except Exception as e:                                  # noqa: BLE001
        ok, err, t_map = 0, "%s: %s" % (type(e).__name__, e), None
    if ok:
        try:
            got = t_map.cpu().numpy()
            ok = 1 if (map_xy is None or got.tobytes() == map_xy.tobytes()) else 0      # (a rank without a copy of its own says "same")
        except Exception as e:                              # noqa: BLE001
            ok, err = 0, "%s: %s" % (type(e).__name__, e)
    try:
        same = torch.tensor([ok], dtype=torch.int64, device=env.comm_dev)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)         # the ONE collective behind the local work: every rank reaches it
        if err is None and not args.map_from_rank0:
            comm["broadcast_map_identical_on_every_rank"] = bool(int(same.item()))
    except Exception as e:                                  # noqa: BLE001
        err = err or "%s: %s" % (type(e).__name__, e)
    if err is not None:                                     # keep the headline figure: the cloud is generated locally
        comm["broadcast_map_error"] = err
    if map_xy is None:
        map_xy = got if (got is not None and err is None) else synth.make_map(cfg["n_map"], cfg["half"])
    return map_xy


def make_inputs(env, args):
    """Synthetic inputs (the reference ships no data), resident in HBM before anything is timed."""
    import torch
    import torch.distributed as dist
    from ndt_slam_amd import shard, synth
    c5 = args.config == "C5"
    cfg = synth.CONFIGS["C5" if c5 else "C3"]
    B = args.batch or (512 if c5 else 256)
    n_scan = cfg["n_scan"]
    rank, world, dev, comm = env.rank, env.world, env.dev, env.comm
    map_xy = fan_out_map(env, args, cfg)
    I = SimpleNamespace(c5=c5, cfg=cfg, B=B, n_scan=n_scan, map_xy=map_xy, truths=None, truth=None)
    if c5:
        # one scan, broadcast from rank 0; seeds sharded: rank r takes every 8th seed of the 4096-seed lattice
        assert world <= SEED_SHARDS and B <= cfg["seeds"] // SEED_SHARDS
        if rank == 0 or world == 1:
            sf = synth.ScanFactory(map_xy, cfg["half"], n_scan)
            scan, truth, _ = sf.make(0)
            pay = torch.from_numpy(np.concatenate([scan.ravel().astype(np.float64), truth])).to(env.comm_dev)
        else:
            pay = torch.empty(2 * n_scan + 3, dtype=torch.float64, device=env.comm_dev)
        if world > 1:
            env.fence()
            t0 = time.perf_counter()
            try:
                dist.broadcast(pay, src=0)
                env.fence()
                comm["broadcast_scan_ms"] = (time.perf_counter() - t0) * 1e3
            except Exception as e:                          # keep the headline figure: every rank generates the scan itself
                comm["broadcast_error"] = "%s: %s" % (type(e).__name__, e)
                sf = synth.ScanFactory(map_xy, cfg["half"], n_scan)
                scan, truth, _ = sf.make(0)
                pay = torch.from_numpy(np.concatenate([scan.ravel().astype(np.float64), truth]))
        pay = pay.cpu().numpy()
        scan = pay[:2 * n_scan].astype(np.float32).reshape(-1, 2)
        I.truth = pay[2 * n_scan:]
        seeds = synth.hypothesis_seeds(I.truth, cfg["seeds"])
        I.inits = np.ascontiguousarray(seeds[rank::SEED_SHARDS][:B])
        I.d_scans = torch.from_numpy(scan).to(dev)
        I.d_off = torch.tensor([0, n_scan], dtype=torch.int64, device=dev)
        I.d_init = torch.from_numpy(I.inits).to(dev)
        I.total_points = n_scan
        I.scans_host, I.off_host = scan, np.array([0, n_scan], np.uint64)
        return I

    def own_shard():
        sf = synth.ScanFactory(map_xy, cfg["half"], n_scan)
        I.scans_host, I.off_host, I.truths, I.inits = sf.batch(rank * B, B)
        I.d_scans = torch.from_numpy(I.scans_host).to(dev)
        I.d_off = torch.from_numpy(I.off_host.astype(np.int64)).to(dev)
        I.d_init = torch.from_numpy(I.inits).to(dev)
        I.total_points = len(I.scans_host)

    if world > 1 and not args.no_scatter:
        # configs[3]: rank 0 holds the whole batch (N x B scans) and fans the shards out, device to device
        if rank == 0:
            sf = synth.ScanFactory(map_xy, cfg["half"], n_scan)
            scans_all, off_all, truths_all, inits_all = sf.batch(0, world * B)
        else:
            scans_all = off_all = inits_all = None
        env.fence()
        t0 = time.perf_counter()
        try:
            d_scans, d_off, d_init = shard.scatter_batch(scans_all, off_all, inits_all, src=0, device=env.comm_dev)
            env.fence()
            comm["scatter_ms"] = (time.perf_counter() - t0) * 1e3
            comm["scatter_bytes"] = int(world * B * n_scan * 8)
            I.d_scans, I.d_off, I.d_init = d_scans.to(dev), d_off.to(dev), d_init.to(dev)
            I.total_points = int(I.d_scans.shape[0])
            I.scans_host = I.off_host = I.inits = None
            if rank == 0:
                I.scans_host, I.off_host, I.inits = shard.shard_batch(scans_all, off_all, inits_all, world, 0)
                I.truths = truths_all[:B]
        except Exception as e:                              # keep the headline figure: every rank makes its own shard
            comm["scatter_error"] = "%s: %s" % (type(e).__name__, e)
            own_shard()
    else:
        own_shard()
    return I


class Pipeline:
    """Contexts, streams and buffers of the pipelined step, and the step itself.

    match launches: `inflight` contexts, each with its own stream and scratch; map builds: one more context on a
    high-priority stream; `inflight + 1` map and result buffers (a step's buffers are free again that many steps later)."""

    def __init__(self, env, args, I, n_events):
        import torch
        from ndt_slam_amd import capi
        self.env, self.args, self.I = env, args, I
        self.torch, self.capi = torch, capi
        dev, world = env.dev, env.world
        k = args.inflight
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
        self.mctx = [capi.Context(env.local_rank) for _ in range(k)]
        for c, s in zip(self.mctx, self.streams):
            assert s.cuda_stream != 0
            c.set_stream(s.cuda_stream)
            if args.workgroups:
                c.set_option(capi.OPT_WORKGROUPS, args.workgroups)
            if args.max_helpers >= 0:
                c.set_option(capi.OPT_MAX_HELPERS, args.max_helpers)
            if args.defer_fitness:
                c.set_option(capi.OPT_DEFER_FITNESS, 1)
        self.stream, self.ctx = self.streams[0], self.mctx[0]
        torch.cuda.set_stream(self.stream)
        self.prm = capi.default_params(resolution=I.cfg["resolution"])     # PCL 1.10 preset; otherwise ndt_mapping.launch:32-36
        if args.libm_f32 is not None:
            self.prm.libm_f32 = args.libm_f32
        self.d_map = torch.from_numpy(I.map_xy).to(dev)
        self.nbuf = k + 1 + (1 if args.defer_fitness else 0)     # (deferred fitness: the launch before last may still be reading its map)
        self.d_res2 = [torch.zeros(I.B * capi.RESULT_BYTES, dtype=torch.uint8, device=dev) for _ in range(self.nbuf)]
        self.side = torch.cuda.Stream(device=dev) if world > 1 else None      # gather of step i while step i + 1 computes
        self.ev_done = [torch.cuda.Event() for _ in range(self.nbuf)]
        self.gathered = [None] * self.nbuf
        torch.cuda.synchronize()
        self.bstream = torch.cuda.Stream(device=dev, priority=-1)     # map builds: small kernels, first in line for freed CUs
        self.pstream = torch.cuda.Stream(device=dev, priority=-1) if args.prepare == "own-stream" else None   # batches prepared ahead (ndt_order_kernel)
        self.bctx = capi.Context(env.local_rank)
        self.bctx.set_stream(self.bstream.cuda_stream)
        self.gmaps = [capi.Map(self.bctx, params=self.prm, dev_ptr=self.d_map.data_ptr(), n=len(I.map_xy), stride=8)
                      for _ in range(self.nbuf)]
        self.gmap = self.gmaps[0]
        torch.cuda.synchronize()
        self.solo_build_ms = []
        for _ in range(3):                                      # the build alone, nothing else on the GPU
            self.gmaps[-1].rebuild(dev_ptr=self.d_map.data_ptr(), n=len(I.map_xy), stride=8)
            self.solo_build_ms.append(self.bctx.last_timing()[0])
        torch.cuda.synchronize()
        self.ev_a = [torch.cuda.Event(enable_timing=args.time_builds) for _ in range(2 * n_events)]
        self.ev_m = [torch.cuda.Event(enable_timing=args.time_builds) for _ in range(2 * n_events)]
        self.stats = {"rebuilt_steps": 0}
        self.cloud_of = lambda i: self.d_map                   # the cloud step i rebuilds its map from (the moving-map leg swaps this)
        self.best_log = []
        self.tp_off = capi.RESULT_DTYPE.fields["trans_prob"][1]
        self.seed_index = ((env.rank + SEED_SHARDS * torch.arange(I.B, dtype=torch.int64)).to(env.comm_dev) if I.c5 else None)
        self.open_build = []                                   # the map whose rebuild_end is still owed (one per context)
        # "step i is done" for other streams: the event the library attaches to the last kernel of the step's launch
        # (ndt_ctx_wait_launch) -- an event RECORD on the match stream is a packet of its own between two kernels, 6 us per
        # step (tools/launch_gap.py); records are only made when --time-builds wants their timestamps
        self.launch_no, self.n_launched = {}, [0] * k

    def wait_step_done(self, stream, j):
        k = self.args.inflight
        if self.args.time_builds:
            stream.wait_event(self.ev_a[2 * j + 1])
        else:
            c = j % k
            self.mctx[c].wait_launch(self.n_launched[c] - 1 - self.launch_no[j], stream.cuda_stream)

    def launch(self, i):
        """a3-a9 for the whole batch of step i: one launch behind this step's build (the library makes the match stream
        wait for the build of the map it is given: no second wait here -- every wait is a packet between two kernels),
        then the collective of the step."""
        torch, args, env, I = self.torch, self.args, self.env, self.I
        from ndt_slam_amd import shard
        k, nbuf, comm = args.inflight, self.nbuf, env.comm
        gm = self.gmaps[i % nbuf]
        st, cx = self.streams[i % k], self.mctx[i % k]
        out = self.d_res2[i % nbuf]
        if i >= nbuf and k > 1:
            self.wait_step_done(st, i - nbuf)          # the previous writer of this result buffer: another stream when inflight > 1
        if env.world > 1:
            st.wait_event(self.ev_done[i % nbuf])      # the gather that last read this result buffer has finished
        if args.time_builds:
            self.ev_a[2 * i].record(st)                # (an event record is a packet between two kernels: timing-only ones are optional)
        gm.align_batch_dev(I.d_scans.data_ptr(), I.d_off.data_ptr(), I.B, I.total_points, I.d_init.data_ptr(),
                           out.data_ptr(), shared_scan=I.c5, stream=st.cuda_stream, ctx=cx)
        self.launch_no[i] = self.n_launched[i % k]
        self.n_launched[i % k] += 1
        if args.time_builds:
            self.ev_a[2 * i + 1].record(st)
        # A collective that raises is reported (comm.*_error) and not tried again: the matches are what the metric
        # counts, and a first run on RCCL must not lose the headline figure to the gather of 55 KB of records.
        if env.world > 1 and not I.c5 and "gather_error" not in comm:    # gather of poses (the only collective on this path)
            self.wait_step_done(self.side, i)
            with torch.cuda.stream(self.side):
                try:
                    self.gathered[i % nbuf] = shard.gather_results(out if not env.rehearsal else out.cpu(), dst=0)
                except Exception as e:                      # noqa: BLE001
                    comm["gather_error"] = "%s: %s" % (type(e).__name__, e)
                self.ev_done[i % nbuf].record(self.side)
        if env.world > 1 and I.c5 and "argmax_error" not in comm:   # configs[4]: arg-max of the hypothesis scores over all ranks (a few bytes)
            self.wait_step_done(self.side, i)
            with torch.cuda.stream(self.side):
                try:
                    tp = out.view(I.B, self.capi.RESULT_BYTES)[:, self.tp_off:self.tp_off + 8].contiguous().view(torch.float64).reshape(I.B)
                    self.best_log.append(shard.best_hypothesis_t(tp.to(env.comm_dev), self.seed_index))    # stays on the device
                except Exception as e:                      # noqa: BLE001
                    comm["argmax_error"] = "%s: %s" % (type(e).__name__, e)
                self.ev_done[i % nbuf].record(self.side)

    def settle_build(self):
        """Collect the verdict on the grid the open rebuild was queued with.  NDT_REBUILT: the cloud's voxel bounding box
        had moved (the common case for the reference's sliding local map, src/PointCloudMap.cpp:119-131) -- the library has
        queued the build again with the right grid, and the matches of that step, which ran on the stale one, are queued
        again here: the step costs one more build and one more launch."""
        if self.open_build:
            gm, j = self.open_build.pop()
            if gm.rebuild_end():
                self.stats["rebuilt_steps"] += 1
                self.launch(j)

    def step(self, i):
        args = self.args
        gm = self.gmaps[i % self.nbuf]
        def prepare(stream):
            # the part of step i's matches that needs nothing but the scans, their guesses and the grid's geometry -- optimiser
            # start, window geometry, voxel order -- as a kernel of its own (ndt_align_batch_prepare_dev; the launch below finds
            # the prepared batch)
            I = self.I
            gm.prepare_batch_dev(I.d_scans.data_ptr(), I.d_off.data_ptr(), I.B, I.total_points, I.d_init.data_ptr(),
                                 shared_scan=I.c5, stream=stream.cuda_stream, ctx=self.mctx[i % args.inflight])
        if args.prepare == "own-stream":
            prepare(self.pstream)
        # a2: rebuild the voxel grid of this step in place, as soon as the matches of step i - nbuf (the last
        # readers of this grid) are done
        if i >= self.nbuf:
            self.wait_step_done(self.bstream, i - self.nbuf)
        if args.time_builds:
            self.ev_m[2 * i].record(self.bstream)
        # two-phase rebuild (ndt_map_rebuild_begin / _end): the build is queued with the voxel grid of the map's last
        # build and the host goes on -- it collects the verdict on that grid one step later, so the GPU never waits for
        # the host's wake-up from the bounding-box read-back (a plain rebuild blocks right here in every step)
        self.settle_build()
        gm.rebuild_begin(self.cloud_of(i).data_ptr(), len(self.I.map_xy), 8)
        self.open_build.append((gm, i))
        if args.prepare == "build-stream":
            prepare(self.bstream)
        if args.time_builds:
            self.ev_m[2 * i + 1].record(self.bstream)
        self.launch(i)

    def run(self, first, count):
        """`count` pipelined steps from step number `first`, fenced on both sides; seconds elapsed."""
        self.env.fence()
        t0 = time.perf_counter()
        for i in range(first, first + count):
            self.step(i)
        self.settle_build()                                    # the verdict on the last step's grid belongs to the timed region
        self.env.fence()
        return time.perf_counter() - t0


# ------------------------------------------------------------------------------------------------ the timed region
def run_timed(pipe, env, args):
    """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; MAX over ranks."""
    import torch
    import torch.distributed as dist
    for i in range(args.warmup):
        pipe.step(i)
    elapsed = pipe.run(args.warmup, args.steps)
    rebuilt_timed = pipe.stats["rebuilt_steps"]
    if env.world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=env.comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, rebuilt_timed


def kernel_figures(pipe, env, args):
    """What the library's own events (attached to the kernels' dispatches) say about the timed steps."""
    import torch
    import torch.distributed as dist
    from ndt_slam_amd import capi
    k, nst = args.inflight, args.steps + args.warmup
    F = SimpleNamespace()
    F.kern_ms = ([pipe.ev_a[2 * i].elapsed_time(pipe.ev_a[2 * i + 1]) for i in range(args.warmup, nst)]
                 if args.time_builds else None)                 # all kernels of a launch
    per_launch = []
    for i in range(args.warmup, nst):
        later = len([j for j in range(i + 1, nst) if j % k == i % k])
        if later < 64:
            per_launch.append(pipe.mctx[i % k].kernel_timing(later))
    F.order_ms = float(max(c.prepare_timing() for c in pipe.mctx))
    F.match_ms = float(np.mean([t[0] for t in per_launch]))
    F.fit_ms = float(np.mean([t[1] for t in per_launch]))
    # step-to-step intervals inside the timed region (start of a launch's match kernel to the start of the next one's),
    # from the same events: the spread shows clock ramp and scheduling noise that a 12 ms timed region hides
    F.step_iv = []
    for i in range(args.warmup + k, nst):
        later = len([j for j in range(i + 1, nst) if j % k == i % k])
        if later + 1 < 64:
            F.step_iv.append(pipe.mctx[i % k].launch_interval(later) / k)
    F.map_ms = ([pipe.ev_m[2 * i].elapsed_time(pipe.ev_m[2 * i + 1]) for i in range(args.warmup, nst)]
                if args.time_builds else None)
    last = (nst - 1) % pipe.nbuf
    F.res = np.frombuffer(pipe.d_res2[last].cpu().numpy().tobytes(), dtype=capi.RESULT_DTYPE)
    assert np.all(F.res["status"] == 0)
    F.per_rank = None
    if env.world > 1:      # per-rank kernel and step figures, so that imbalance across ranks is visible in the SCALE record
        mine = torch.tensor([F.match_ms, float(max(t[0] for t in per_launch)), float(F.res["evals"].sum())],
                            dtype=torch.float64, device=env.comm_dev)
        allr = [torch.empty_like(mine) for _ in range(env.world)]
        dist.all_gather(allr, mine)
        F.per_rank = {"kernel_ms": [float(t[0]) for t in allr], "kernel_ms_max": [float(t[1]) for t in allr],
                      "evals": [float(t[2]) for t in allr]}
    return F


# ------------------------------------------------------------------------------------------------ side legs (N = 1)
def leg_steady(pipe, args, first):
    """The same pipelined step for >= 200 more steps -- the driver's 20 timed steps are 10 ms, about the length of the
    GPU's clock ramp."""
    el = pipe.run(first, args.steady_steps)
    return {"steps": args.steady_steps, "ms_per_step": 1e3 * el / args.steady_steps,
            "value": pipe.I.B * args.steady_steps / el,
            "note": "the timed loop continued for this many more steps (clock ramped, queues warm)"}


def leg_moving(pipe, args, first_i, margin):
    """`--moving-steps` steps over two clouds that take turns every `--moving-every` steps; ndt_params::grid_margin =
    margin: a local map whose voxel bounding box moves, the path the reference takes (src/PointCloudMap.cpp:119-131) --
    the speculative grid of the two-phase rebuild is then wrong, the library builds again and the step's matches are
    queued again (Pipeline.settle_build)."""
    torch, capi, I = pipe.torch, pipe.capi, pipe.I
    map_xy = I.map_xy
    # cloud B = the cloud with ONE point moved a voxel beyond the bounding box's lower corner: the grid's origin moves by a
    # voxel in x and y (every voxel index changes), the matches stay what they were
    mv = map_xy.copy()
    mv[0] = map_xy.min(axis=0) - np.float32(I.cfg["resolution"])
    d_map_b = torch.from_numpy(mv).to(pipe.env.dev)
    clouds = [pipe.d_map, d_map_b]
    k_mv = max(1, args.moving_every)
    pm = capi.Params.from_buffer_copy(pipe.prm)
    pm.grid_margin = margin
    for g in pipe.gmaps:
        g.params = pm
    pipe.cloud_of = lambda i: clouds[((i - first_i) // k_mv) % 2]
    before = pipe.stats["rebuilt_steps"]
    el = pipe.run(first_i, args.moving_steps)
    last_i = first_i + args.moving_steps - 1
    got_mv = pipe.d_res2[last_i % pipe.nbuf].cpu().numpy().tobytes()
    fresh = capi.Map(pipe.ctx, params=pipe.prm, dev_ptr=pipe.cloud_of(last_i).data_ptr(), n=len(map_xy), stride=8)   # a build from scratch, exact grid
    chk = torch.zeros(I.B * capi.RESULT_BYTES, dtype=torch.uint8, device=pipe.env.dev)
    fresh.align_batch_dev(I.d_scans.data_ptr(), I.d_off.data_ptr(), I.B, I.total_points, I.d_init.data_ptr(), chk.data_ptr(),
                          shared_scan=I.c5, stream=pipe.stream.cuda_stream)
    torch.cuda.synchronize()
    leg = {"steps": args.moving_steps, "box_moves_every": k_mv, "grid_margin": margin,
           "rebuilt_steps": pipe.stats["rebuilt_steps"] - before,
           "ms_per_step_moving": 1e3 * el / args.moving_steps, "value_moving": I.B * args.moving_steps / el,
           "identical_to_a_fresh_build": bool(chk.cpu().numpy().tobytes() == got_mv)}
    fresh.close()
    pipe.cloud_of = lambda i: pipe.d_map
    for g in pipe.gmaps:
        g.params = pipe.prm
    return leg


def leg_single_scan(pipe, out):
    """configs[1]: one scan (latency of a single match, same kernel at B = 1), with every idle CU helping and alone."""
    torch, capi, I = pipe.torch, pipe.capi, pipe.I
    one = torch.zeros(capi.RESULT_BYTES, dtype=torch.uint8, device=pipe.env.dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def one_scan(cx):
        ts = []
        for _ in range(5):
            e0.record(pipe.stream)
            pipe.gmap.align_batch_dev(I.d_scans.data_ptr(), I.d_off.data_ptr(), 1, int(I.off_host[1]), I.d_init.data_ptr(),
                                      one.data_ptr(), stream=pipe.stream.cuda_stream, ctx=cx)
            e1.record(pipe.stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))
    out["single_scan_ms"] = one_scan(pipe.ctx)
    r1 = np.frombuffer(one.cpu().numpy().tobytes(), dtype=capi.RESULT_DTYPE)[0]
    solo_ctx = capi.Context(pipe.env.local_rank)
    solo_ctx.set_stream(pipe.stream.cuda_stream)
    solo_ctx.set_option(capi.OPT_MAX_HELPERS, 0)
    solo_ms = one_scan(solo_ctx)
    # per-evaluation latency: one derivative pass over the 10k points of one scan (setup and the fitness pass
    # included in the numerator, so an upper bound), by one workgroup alone and with the idle CUs helping
    out["per_eval_us"] = {"one_workgroup": 1e3 * solo_ms / int(r1["evals"]), "all_cus_helping": 1e3 * out["single_scan_ms"] / int(r1["evals"]),
                          "evals": int(r1["evals"]), "single_scan_one_workgroup_ms": solo_ms}
    solo_ctx.close()


def leg_reference_faithful(pipe, out):
    """The reference's own call pattern (src/ScanMatcher.cpp:40,45; its timer src/PoseEstimator.cpp:15,38-40 covers
    setInputTarget + align): ONE scan per call, the NDT map rebuilt inside every call, host pointers in and out --
    ndt_map_build (1M points over PCIe, synchronous) + ndt_align (one 10k-point scan, result back), median of 24 calls."""
    capi, I = pipe.capi, pipe.I
    rf_ctx = capi.Context(pipe.env.local_rank)
    rf_map = capi.Map(rf_ctx, I.map_xy, pipe.prm)
    ts, tb = [], []
    for k in range(24):
        sc = I.scans_host[int(I.off_host[k % I.B]):int(I.off_host[k % I.B + 1])]
        t0 = time.perf_counter()
        rf_map.rebuild(xy=I.map_xy)
        t1 = time.perf_counter()
        rr = rf_map.align(sc, I.inits[k % I.B])
        ts.append(time.perf_counter() - t0); tb.append(t1 - t0)
        assert int(rr["status"]) == 0
    out["reference_faithful"] = {
        "pattern": "per call: ndt_map_build(host pointer, %d points) + ndt_align(host pointer, one %d-point scan): what "
                   "src/PoseEstimator.cpp:15-40 times; PCIe transfers and every host synchronisation included" % (len(I.map_xy), I.n_scan),
        "gpu_ms_per_match": 1e3 * float(np.median(ts)), "gpu_matches_per_s": 1.0 / float(np.median(ts)),
        "gpu_map_build_ms": 1e3 * float(np.median(tb)), "calls": len(ts)}
    rf_map.close(); rf_ctx.close()


def leg_multi_hypothesis(pipe, env):
    """configs[4] as a side figure of the default run, on every rank: 512 seeds x one scan vs a 5M-point map (map
    rebuild + matches).  At N > 1 this is the whole configs[4] pattern -- the scan broadcast from rank 0, the 4096-seed
    lattice cut into strided shards, the arg-max of the scores over all ranks -- so that the driver's multi-GPU run
    executes it too; a failure here is reported in the line, it does not take the headline figure down."""
    import torch
    import torch.distributed as dist
    from ndt_slam_amd import capi, shard, synth
    rank, world, dev = env.rank, env.world, env.dev
    stream = pipe.stream
    mh = {}
    try:
        cfg5 = synth.CONFIGS["C5"]
        m5 = synth.make_map(cfg5["n_map"], cfg5["half"])
        n5 = cfg5["n_scan"]
        if rank == 0:
            sf5 = synth.ScanFactory(m5, cfg5["half"], n5)
            sc5, truth5, _ = sf5.make(0)
            pay = torch.from_numpy(np.concatenate([sc5.ravel().astype(np.float64), truth5])).to(env.comm_dev)
        else:
            pay = torch.empty(2 * n5 + 3, dtype=torch.float64, device=env.comm_dev)
        if world > 1:
            env.fence()
            t0 = time.perf_counter()
            dist.broadcast(pay, src=0)
            env.fence()
            mh["broadcast_scan_ms"] = (time.perf_counter() - t0) * 1e3
        pay = pay.cpu().numpy()
        sc5 = pay[:2 * n5].astype(np.float32).reshape(-1, 2); truth5 = pay[2 * n5:]
        seeds5 = np.ascontiguousarray(synth.hypothesis_seeds(truth5, cfg5["seeds"])[rank::SEED_SHARDS])
        d_m5 = torch.from_numpy(m5).to(dev); d_s5 = torch.from_numpy(sc5).to(dev)
        d_o5 = torch.tensor([0, len(sc5)], dtype=torch.int64, device=dev); d_i5 = torch.from_numpy(seeds5).to(dev)
        d_r5 = torch.zeros(len(seeds5) * capi.RESULT_BYTES, dtype=torch.uint8, device=dev)
        g5 = capi.Map(pipe.ctx, params=capi.default_params(resolution=cfg5["resolution"]), dev_ptr=d_m5.data_ptr(), n=len(m5), stride=8)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        tb, tm, tw = [], [], []
        gidx = (rank + SEED_SHARDS * torch.arange(len(seeds5), dtype=torch.int64)).to(env.comm_dev)
        tpo = capi.RESULT_DTYPE.fields["trans_prob"][1]
        best_g = None
        for _ in range(4):
            env.fence()
            t0 = time.perf_counter()
            e0.record(stream)
            g5.rebuild(dev_ptr=d_m5.data_ptr(), n=len(m5), stride=8)
            e1.record(stream)
            g5.align_batch_dev(d_s5.data_ptr(), d_o5.data_ptr(), len(seeds5), len(sc5), d_i5.data_ptr(), d_r5.data_ptr(),
                               shared_scan=True, stream=stream.cuda_stream)
            e2.record(stream)
            if world > 1:
                with torch.cuda.stream(stream):
                    tp = d_r5.view(len(seeds5), capi.RESULT_BYTES)[:, tpo:tpo + 8].contiguous().view(torch.float64).reshape(-1)
                    best_g = shard.best_hypothesis_t(tp.to(env.comm_dev), gidx)
            env.fence()
            tw.append((time.perf_counter() - t0) * 1e3)
            tb.append(e0.elapsed_time(e1)); tm.append(e1.elapsed_time(e2))
        r5 = np.frombuffer(d_r5.cpu().numpy().tobytes(), dtype=capi.RESULT_DTYPE)
        b5 = int(np.argmax(r5["trans_prob"]))
        fit5 = pipe.ctx.kernel_timing(0)[1]
        mh.update({"workload": "configs[4]: %d seeds per GPU x %d GPU(s), one %d-pt scan vs %d-pt map" % (len(seeds5), world, len(sc5), len(m5)),
                   "map_build_ms": float(np.median(tb[1:])), "match_ms": float(np.median(tm[1:])), "fitness_kernels_ms": float(fit5),
                   "step_wall_ms": float(np.median(tw[1:])),
                   "seeds_per_s": world * len(seeds5) / (float(np.median(tw[1:])) * 1e-3),
                   "mean_evals": float(r5["evals"].mean()),
                   "best_seed_err_m": float(np.hypot(*(r5["pose"][b5][:2] - truth5[:2])))})
        if best_g is not None:
            mh["best_over_all_ranks"] = {"trans_prob": float(best_g[0].item()), "seed": int(best_g[1].item())}
        g5.close()
        del d_m5, d_r5
    except Exception as e:                              # noqa: BLE001  (reported, not fatal)
        mh["error"] = "%s: %s" % (type(e).__name__, e)
    return mh


def leg_front_end(pipe, out):
    """Row f1 (source pre-filter, pcl::ApproximateVoxelGrid): raw scans 3x oversampled -> filtered scans, all on the
    device; reported beside the headline metric, not part of it (the 10k-pt scans of the metric are post-filter clouds by
    definition).  Then a whole front-end step for the batch without leaving the device (rows f2 + f1 +
    a2 + a3-a9 + f2): odometry prediction -> pre-filter -> map rebuild -> matches -> EKF fusion."""
    torch, capi, I = pipe.torch, pipe.capi, pipe.I
    dev, stream, ctx, B = pipe.env.dev, pipe.stream, pipe.ctx, I.B
    scans, off = I.scans_host, I.off_host
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
    inits = I.inits
    pred0 = np.column_stack([inits[:, 0], inits[:, 1], np.degrees(inits[:, 2])])
    d_last = torch.from_numpy(pred0).to(dev); d_prevo = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d_curo = torch.zeros_like(d_prevo)                       # zero odometry motion: prediction = last pose
    d_mo = torch.zeros_like(d_prevo); d_pred = torch.zeros_like(d_prevo); d_in2 = torch.zeros_like(d_prevo)
    d_lc = torch.from_numpy(np.tile(np.eye(3).ravel() * 1e-4, (B, 1))).to(dev)
    d_fu = torch.zeros_like(d_prevo); d_cv = torch.zeros(B, 9, dtype=torch.float64, device=dev)
    d_ok = torch.zeros(B, dtype=torch.int32, device=dev)
    fprm = capi.default_fuse_params(score_thre=0.5)
    fmap = capi.Map(ctx, params=pipe.prm, dev_ptr=pipe.d_map.data_ptr(), n=len(I.map_xy), stride=8)
    ts = []
    for _ in range(5):
        e0.record(stream)
        ctx.predict_batch_dev(d_curo.data_ptr(), d_prevo.data_ptr(), d_last.data_ptr(), B, d_mo.data_ptr(),
                              d_pred.data_ptr(), d_in2.data_ptr(), stream=stream.cuda_stream)
        ctx.prefilter_batch_dev(d_raw.data_ptr(), 8, d_roff.data_ptr(), B, len(raw), 0.05, d_f.data_ptr(),
                                d_foff.data_ptr(), stream=stream.cuda_stream)
        fmap.rebuild(dev_ptr=pipe.d_map.data_ptr(), n=len(I.map_xy), stride=8)
        fmap.align_batch_dev(d_f.data_ptr(), d_foff.data_ptr(), B, len(raw), d_in2.data_ptr(), pipe.d_res2[0].data_ptr(),
                             stream=stream.cuda_stream)
        ctx.fuse_batch_dev(pipe.d_res2[0].data_ptr(), d_pred.data_ptr(), d_mo.data_ptr(), d_last.data_ptr(), d_lc.data_ptr(), B,
                           fprm, d_fu.data_ptr(), d_cv.data_ptr(), d_ok.data_ptr(), stream=stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    out["front_end_step"] = {"stages": "predict + pre-filter + map rebuild + match + fuse, all on the device",
                             "scans": B, "raw_points_per_scan": int(len(raw) // B), "ms": float(np.median(ts)),
                             "scans_per_s": B / (float(np.median(ts)) * 1e-3), "accepted": int(d_ok.sum().item())}
    fmap.close()


def leg_local_map(pipe, args, out):
    """Row f3 (local-map assembly, Submap::makeMap with moving-object removal): a submap of 12 registered scans
    of the metric's size (walls seen again by every scan + an object that moves, synth.submap_scans) assembled
    on the device; the oracle's literal octree does the same on one host core (checker and CPU figure).
    Reported beside the headline metric, not part of it."""
    torch, I = pipe.torch, pipe.I
    from ndt_slam_amd import synth
    dev, stream = pipe.env.dev, pipe.stream
    ns = 12
    reg = synth.submap_scans(ns, I.cfg["n_scan"])
    reg_off = np.zeros(ns + 1, np.uint64)
    reg_off[1:] = np.cumsum([len(r) for r in reg])
    d_reg = torch.from_numpy(np.concatenate(reg)).to(dev)
    d_lm = torch.empty((len(d_reg) + 1, 2), dtype=torch.float32, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(6):
        e0.record(stream)
        pipe.ctx.make_map_dev(d_reg.data_ptr(), 8, reg_off, True, True, True, 0.05, 0.1, d_lm.data_ptr(), d_cnt.data_ptr(),
                              stream=stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    n_lm = int(d_cnt.item())
    lm = {"scans": ns, "points": int(len(d_reg)), "kept": n_lm, "resol": 0.05, "thre_neighbor": 0.1,
          "ms": float(np.median(ts[1:]))}
    if not args.no_cpu_baseline:
        from oracle import ndt_oracle as O                 # the checker (test infrastructure), behind the timed region
        t = time.perf_counter()
        ref_lm = O.make_map(reg, True, True, True, 0.05, 0.1)
        lm["cpu_ms_1core"] = (time.perf_counter() - t) * 1e3
        lm["identical"] = bool(n_lm == len(ref_lm) and d_lm[:n_lm].cpu().numpy().tobytes() == ref_lm.tobytes())
    out["local_map"] = lm


def leg_cpu_baseline(args, I, res, out):
    """CPU baseline: the oracle (a port -- PCL itself is absent) on this box's host cores, rank 0 at N = 1 only, on a
    bounded sample of the same batch (median of --cpu-reps repetitions).  `value` runs every derivative pass the reference
    runs; `memoised` is the same port with the GPU path's skip switched on (a line-search trial that repeats the step
    length of the pass before it is not run again): the like-for-like figure beside the reference-faithful one."""
    from oracle import ndt_oracle as O                      # the checker (test infrastructure), behind the timed region
    B, cfg, c5 = I.B, I.cfg, I.c5
    scans_host, off_host, inits = I.scans_host, I.off_host, I.inits
    ns = min(args.cpu_sample, B)
    t = time.perf_counter()
    om = O.Map(I.map_xy, O.default_params(resolution=cfg["resolution"]))
    t_build = time.perf_counter() - t
    reps = max(1, args.cpu_reps)
    if c5:
        run = lambda nt, **kw: om.align_batch(scans_host, off_host, inits[:ns], nthreads=nt, shared_scan=True, **kw)
    else:
        sub_off = off_host[:ns + 1]
        run = lambda nt, **kw: om.align_batch(scans_host[:int(sub_off[-1])], sub_off, inits[:ns], nthreads=nt, **kw)
    run(1)                                   # one warm-up, excluded
    t1 = []
    for _ in range(reps):
        t = time.perf_counter()
        ref = run(1, run_stats=True)
        t1.append(time.perf_counter() - t)
    tm = []
    for _ in range(min(reps, 3)):
        t = time.perf_counter()
        ref_m = run(1, memoise=True, run_stats=True)
        tm.append(time.perf_counter() - t)
    # all-cores legs: OpenMP over independent matches with the box's share of host threads for one GPU
    # (nproc / 8, at most 32) and with every hardware thread (nproc).  The sample is repeated so that every thread
    # gets several matches (256 matches on 256 threads would time the slowest match).
    nproc = os.cpu_count() or 1
    legs = {}
    for ncpu in sorted({min(nproc, 32), nproc}):
        rep_k = max(1, (4 * ncpu + ns - 1) // ns)
        if c5:
            big = lambda nt, k=rep_k: om.align_batch(scans_host, off_host, np.tile(inits[:ns], (k, 1)), nthreads=nt, shared_scan=True)
        else:
            pts = scans_host[:int(sub_off[-1])]
            boff = np.concatenate([[0], np.cumsum(np.tile(np.diff(sub_off.astype(np.int64)), rep_k))]).astype(np.uint64)
            big = lambda nt, k=rep_k, pts=pts, boff=boff: om.align_batch(np.tile(pts, (k, 1)), boff, np.tile(inits[:ns], (k, 1)), nthreads=nt)
        tn = []
        for _ in range(min(reps, 3)):
            t = time.perf_counter()
            big(ncpu)
            tn.append(time.perf_counter() - t)
        legs[ncpu] = {"value": rep_k * ns / float(np.median(tn)), "cores": ncpu, "matches_timed": rep_k * ns}
    t_align = float(np.median(t1))
    d = res["pose"][:ns] - ref["pose"]
    d[:, 2] = (d[:, 2] + math.pi) % (2 * math.pi) - math.pi
    out["cpu_baseline"] = {
        "value": ns / t_align, "unit": "matches/s", "cores": 1, "kind": "port",
        "sample": "first %d of the %d matches, median of %d repetitions (one warm-up excluded), 1 thread, map built once "
                  "(amortised); oracle/ndt_oracle.c, -O2, grid-hash neighbour lookup (faster than PCL's kd-tree); it runs every "
                  "derivative pass the reference runs, the repeated line-search trials included" % (ns, B, reps),
        "cpu_model": cpu_model(), "nproc": os.cpu_count(),
        "map_build_s": t_build,
        "reference_faithful_matches_per_s": 1.0 / (t_build + t_align / ns),
        "memoised": {"value": ns / float(np.median(tm)), "cores": 1,
                     "passes_run_mean": float(ref_m["evals"].mean()), "passes_reference_mean": float(ref["evals"].mean()),
                     "passes_the_gpu_runs_mean": float(ref["evals_run"].mean()),
                     "identical_to_the_full_run": bool(all(np.array_equal(ref_m[k], ref[k]) for k in
                                                           ("T00", "T10", "T03", "T13", "iters", "ref_evals", "fitness", "score", "H"))),
                     "note": "the same port with the GPU path's skip (ndt_oracle_set_memoise): a trial at the step length of the "
                             "pass just run re-uses that pass's totals.  It still runs PCL's Hessian-only passes and the getHessian "
                             "pass as passes of their own (the GPU path has them fused into the passes it runs: "
                             "passes_the_gpu_runs_mean)"},
        "all_cores": legs[min(nproc, 32)], "all_cores_nproc": legs[nproc],
    }
    if "reference_faithful" in out:
        out["reference_faithful"]["cpu_matches_per_s"] = out["cpu_baseline"]["reference_faithful_matches_per_s"]
        out["reference_faithful"]["gpu_over_cpu"] = (out["reference_faithful"]["gpu_matches_per_s"] /
                                                     out["cpu_baseline"]["reference_faithful_matches_per_s"])
    out["parity"] = {"max_dpos_m": float(np.abs(d[:, :2]).max()), "max_dyaw_rad": float(np.abs(d[:, 2]).max()),
                     "same_iters": bool(np.all(res["iters"][:ns] == ref["iters"])), "sample": ns,
                     # the whole-path integer check at config scale: pairs over exactly the passes the device runs
                     "same_pairs_run": bool(np.all(np.abs(res["kbar"][:ns] - ref["kbar_run"]) <= 1e-12 * np.maximum(1.0, ref["kbar_run"])))}
    out["gpu_over_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"]
    out["gpu_over_cpu_1core_note"] = ("mixed: the CPU figure runs the repeated line-search trials the GPU skips; like for like: "
                                      "gpu_over_cpu_1core_memoised")
    out["gpu_over_cpu_1core_memoised"] = out["value"] / out["cpu_baseline"]["memoised"]["value"]


# ------------------------------------------------------------------------------------------------ the JSON line
def headline(env, args, I, pipe, F, elapsed, rebuilt_timed, legs):
    cfg, c5, B, n_scan, world, res = I.cfg, I.c5, I.B, I.n_scan, env.world, F.res
    traffic, traffic_src = measured_traffic()
    alg_bytes = algorithmic_bytes(res, n_scan, fitness=False)
    achieved = alg_bytes / (F.match_ms * 1e-3) / 1e9
    fit_bytes = float(len(res) * n_scan * 16.0)
    accepted = (res["converged"] == 1) & (res["fitness"] <= 0.5)      # src/ScanMatcher.cpp:50 with score_thre 0.5
    what = ("BASELINE configs[4] share of one GPU: %d seed poses x one %d-pt scan vs %d-pt map, 0.5 m voxels (x%d ranks of the "
            "8 x 512 = 4096-seed lattice)" % (B, n_scan, cfg["n_map"], world)) if c5 else (
            "BASELINE configs[2]: batch of %d scans x %d pts vs shared %d-pt map, 0.5 m voxels, per GPU (configs[3] sharding "
            "at N>1)" % (B, n_scan, cfg["n_map"]))
    v_cells = int(pipe.gmaps[0].info().n_cells)              # voxels of the search set (>= min_pts points)
    mb_bytes = len(I.map_xy) * 8.0 + v_cells * 24.0
    mb_ms = float(np.median(pipe.solo_build_ms))
    mb_traffic, mb_src = side_traffic("c5_build" if c5 else "build")
    ft_traffic, ft_src = side_traffic("c5_fitness" if c5 else "fitness")
    map_build_roofline = {"kernels": "the chain of ndt_map_build_dev (DESIGN.md 4.1), alone on the GPU", "ms": mb_ms,
                          "map_points": int(len(I.map_xy)), "voxels": v_cells, "algorithmic_bytes_per_build": mb_bytes,
                          "achieved_GBps": mb_bytes / (mb_ms * 1e-3) / 1e9, "frac": mb_bytes / (mb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "traffic": mb_traffic, "traffic_source": mb_src,
                          "note": "SURVEY 8d: M x 8 B read + V x 24 B written"}
    out = {
        "metric": "scan-matches/sec (10k-pt scan vs 1M-pt NDT map)" if not c5 else "scan-matches/sec (seed poses of one 10k-pt scan vs 5M-pt NDT map)",
        "value": world * B * args.steps / elapsed,
        "unit": "matches/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "ms_per_step_min": float(min(F.step_iv)) if F.step_iv else None, "ms_per_step_max": float(max(F.step_iv)) if F.step_iv else None,
        "ms_per_step_median": float(np.median(F.step_iv)) if F.step_iv else None,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": what + "; map rebuilt every step from the same cloud (the rebuild of step i+1 overlaps the end of step i's "
                               "matches; the cloud's voxel bounding box never moves in the timed region, so the grid the two-phase "
                               "rebuild queues ahead with is always the right one -- `moving_map` times a box that moves, "
                               "`reference_faithful` rebuilds synchronously); %d match launch(es) in flight; parameter preset PCL 1.10" % args.inflight,
                   "matches_per_gpu": B, "scans_per_gpu": B if not c5 else 1, "scan_points": n_scan, "map_points": cfg["n_map"],
                   "resolution": cfg["resolution"], "inflight": args.inflight, "workgroups": args.workgroups, "max_helpers": args.max_helpers,
                   "prepared_ahead": args.prepare,
                   "parallelism": ("seed-shards x%d, scan broadcast, arg-max of scores" % world) if c5 else
                                  ("scan-shards x%d, %s, gather of results" % (world, "every rank generates its shard" if (args.no_scatter or world == 1) else "scatter from rank 0"))},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "note": "achieved = SURVEY 8d algorithmic bytes of the match kernel's passes (E x N x (8 + 20 Kbar) per "
                             "match, E = the derivative passes actually RUN: a line-search trial that repeats the step length "
                             "of the pass before it -- a sixth of the reference's passes on this workload -- has that pass's "
                             "totals and is not run again, DESIGN.md 4.2; `ref_evals_mean` counts what the reference runs) / "
                             "duration of one launch of that kernel.  The kernel is VALU-issue bound, not HBM bound: "
                             "each voxel record is staged once per match in LDS, so the measured HBM traffic is below the "
                             "algorithmic bytes (DESIGN.md 4.2).  The fitness score (N x 16 B per match) is a kernel of its "
                             "own, listed under `fitness`",
                     "kernel": "ndt_align_kernel", "kernel_ms": F.match_ms,
                     "order_kernel": {"kernel": "ndt_order_kernel", "ms": F.order_ms, "prepared_ahead": args.prepare,
                                      "note": "optimiser start + window geometry + voxel order of the step's scans as a kernel of its own on the build "
                                              "stream, behind the map's rebuild (ndt_align_batch_prepare_dev): it runs while the previous step's "
                                              "matches run out; with --no-prepare the owners do the same work inside ndt_align_kernel (+ ~20 us per scan)"},
                     "launch_interval_ms": float(np.mean(F.kern_ms)) if F.kern_ms else None,
                     "fitness": {"kernels": "fitness_points_kernel + fitness_far_kernel + fitness_reduce_kernel" if c5 else "fitness_points_kernel + fitness_reduce_kernel", "ms": F.fit_ms,
                                 "algorithmic_bytes_per_launch": fit_bytes,
                                 "achieved_GBps": fit_bytes / (F.fit_ms * 1e-3) / 1e9 if F.fit_ms > 0 else None,
                                 "frac": fit_bytes / (F.fit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if F.fit_ms > 0 else None,
                                 "traffic": ft_traffic, "traffic_source": ft_src,
                                 "note": "SURVEY 8d: N x 16 B per match (the point + its nearest map point)"},
                     "map_build": map_build_roofline,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "mean_evals": float(res["evals"].mean()), "max_evals": int(res["evals"].max()),
                     "ref_evals_mean": float(res["ref_evals"].mean()),
                     "mean_kbar": float(res["kbar"].mean())},
        "map_build_ms": mb_ms, "map_build_in_step_ms": float(np.mean(F.map_ms)) if F.map_ms else None,
        "converged": int(res["converged"].sum()),
        "accepted": int(accepted.sum()), "accepted_frac": float(accepted.mean()),
        "rebuilt_steps": rebuilt_timed,
    }
    out.update(legs)
    # fp64 arithmetic rate next to the byte rate (SURVEY 8d asks for it so that the HBM figure is not misread):
    # per (point, voxel) pair ~ 100 flops, per point-evaluation ~ 60 (transform, voxel index, 9 radius tests)
    pe = float(np.sum(res["evals"].astype(np.float64) * n_scan))
    flops = pe * 60.0 + float(np.sum(res["evals"] * res["kbar"])) * n_scan * 100.0
    valu = measured_valu() or {}
    valu.update({"point_evals_per_launch": pe, "flops_per_launch_estimate": flops,
                 "fp64_frac_of_peak": flops / (F.match_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                 "point_evals_per_us": pe / (F.match_ms * 1e3)})
    out["roofline"]["valu"] = valu
    if I.truths is not None:
        err = res["pose"] - I.truths
        err[:, 2] = (err[:, 2] + math.pi) % (2 * math.pi) - math.pi
        out["median_abs_err_m"] = float(np.median(np.hypot(err[:, 0], err[:, 1])))
    if c5:
        best = int(np.argmax(res["trans_prob"]))
        out["best_hypothesis"] = {"local_index": best, "trans_prob": float(res["trans_prob"][best]),
                                  "err_m": float(np.hypot(*(res["pose"][best][:2] - I.truth[:2])))}
        if pipe.best_log:
            out["best_hypothesis"]["global"] = {"trans_prob": float(pipe.best_log[-1][0].item()), "seed": int(pipe.best_log[-1][1].item())}
    if env.comm:
        out["comm"] = env.comm
    if F.per_rank:
        out["per_rank"] = F.per_rank
    return out


# ------------------------------------------------------------------------------------------------ main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args, argv))          # the launcher: nothing below runs in this process
    import torch.distributed as dist
    env = init_env(args)
    I = make_inputs(env, args)
    rank, world, c5 = env.rank, env.world, I.c5

    nst = args.steps + args.warmup
    side_legs = world == 1 and not args.no_single_scan and not c5      # steady-state and moving-map legs behind the timed region
    n_extra = (max(0, args.steady_steps) + 2 * max(0, args.moving_steps)) if side_legs else 0
    pipe = Pipeline(env, args, I, nst + n_extra)

    elapsed, rebuilt_timed = run_timed(pipe, env, args)
    F = kernel_figures(pipe, env, args)

    legs, nxt = {}, nst
    if side_legs and args.steady_steps > 0:
        legs["steady_state"] = leg_steady(pipe, args, nxt)
        nxt += args.steady_steps
    if side_legs and args.moving_steps > 0:
        legs["moving_map"] = leg_moving(pipe, args, nxt, 0)
        legs["moving_map"]["note"] = ("two map buffers, each speculating on the grid of ITS last build: a move of the box costs "
                                      "one extra build + one repeated launch on each of them")
        nxt += args.moving_steps
        legs["moving_map_margin"] = leg_moving(pipe, args, nxt, args.moving_margin)
        legs["moving_map_margin"]["note"] = ("ndt_params::grid_margin = %d voxels: the grid queued ahead stays good while the box "
                                             "moves inside the margin (same records; include/ndt_mi355x.h)" % args.moving_margin)
        nxt += args.moving_steps

    out = headline(env, args, I, pipe, F, elapsed, rebuilt_timed, legs) if rank == 0 else None

    side_figures = rank == 0 and world == 1 and not args.no_single_scan
    if side_figures and not c5:
        leg_single_scan(pipe, out)
        leg_reference_faithful(pipe, out)
    if not c5 and not args.no_single_scan and (world > 1 or rank == 0):
        mh = leg_multi_hypothesis(pipe, env)
        if rank == 0:
            out["multi_hypothesis"] = mh
    if side_figures and not c5:
        leg_front_end(pipe, out)
        leg_local_map(pipe, args, out)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        leg_cpu_baseline(args, I, F.res, out)

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
