"""What does other work on the chip cost the match kernel?  Back-to-back launches of the bench batch (no map build) with, on a
second low-priority... (normal) stream, (a) nothing, (b) a kernel that only computes (fp64 FMA chains, no memory traffic),
(c) a kernel that only streams memory -- each sized to last about as long as the launch and queued beside every launch.  The
match kernel holds every CU at first (one workgroup per CU, all of its LDS), so the companion's workgroups start as the match
kernel's idle workgroups leave: exactly the CU time "left idle" by the launch's tail.  Prints the match kernel's own duration
(ndt_kernel_timing) in the three cases.  Needs build_tmp/spin.so (tools/repro/spin.hip)."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndt_slam_amd import capi, synth               # noqa: E402

spin = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_tmp", "spin.so"))
spin.launch_spin_alu.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
spin.launch_spin_mem.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
cfg = synth.CONFIGS["C3"]
B = 256
m = synth.make_map(cfg["n_map"], cfg["half"])
sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
scans, off, truths, inits = sf.batch(0, B)
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
st, other = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
ctx.set_stream(st.cuda_stream)
gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
d_scans = torch.from_numpy(scans).to(dev); d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
d_init = torch.from_numpy(inits).to(dev)
out = torch.zeros(B * capi.RESULT_BYTES, dtype=torch.uint8, device=dev)
sink = torch.zeros(1 << 20, dtype=torch.float64, device=dev)
big = torch.zeros(64 << 20, dtype=torch.float32, device=dev)          # 256 MB: past every cache
N = 120


def solo(kind, arg):
    """duration of the companion alone"""
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(other); launch(kind, arg); e1.record(other); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def launch(kind, arg):
    if kind == "alu":
        assert spin.launch_spin_alu(other.cuda_stream, 256, arg, sink.data_ptr()) == 0
    elif kind == "mem":
        assert spin.launch_spin_mem(other.cuda_stream, 256, big.data_ptr(), sink.data_ptr(), big.numel() // 4, arg) == 0


for kind, arg in (("none", 0), ("alu", 900), ("alu", 300), ("mem", 1), ("none", 0)):
    if kind != "none":
        for _ in range(2): solo(kind, arg)
        alone = solo(kind, arg)
    else:
        alone = 0.0
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            gm.align_batch_dev(d_scans.data_ptr(), d_off.data_ptr(), B, len(scans), d_init.data_ptr(), out.data_ptr(), stream=st.cuda_stream, ctx=ctx)
            launch(kind, arg)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N * 1e3
    k = [ctx.kernel_timing(j) for j in range(40)]
    km, kf = np.mean([x[0] for x in k]), np.mean([x[1] for x in k])
    print("%-5s arg %4d: companion alone %.3f ms | per launch %.4f ms | match kernel %.4f fitness %.4f" % (kind, arg, alone, dt, km, kf))
