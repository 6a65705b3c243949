"""The match launch beside foreign queues (round-2 verdict, weak point 5): the library lives inside processes that own
other streams (ROS drivers, PyTorch), and the match kernel is a persistent kernel whose workgroups wait for each other.

(i)  a foreign kernel holds 48 CUs (150 KB of LDS each: no match workgroup fits beside it) for 20 ms at a time while
     match launches run back to back: the 48 match workgroups that cannot become resident must not be waited for -- the
     resident ones take their scans over (claims, ndt_match.hip.h);
(ii) an idle stream and a low-priority stream exist, created before and after the context.
Every launch must finish in well under 50 ms with byte-identical records and no NDT_E_HIP."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hog(tmp_path_factory):
    import torch                                             # first: the process must end up with ONE HIP runtime (torch's)
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available to build the foreign kernel")
    out = str(tmp_path_factory.mktemp("hog") / "hog.so")
    subprocess.check_call([hipcc, "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out,
                           os.path.join(ROOT, "tools", "repro", "hog.hip")])
    L = ctypes.CDLL(out)
    L.hog_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_void_p]
    L.hog_launch.restype = ctypes.c_int
    return L


@pytest.mark.parametrize("B", [256, 24, 1])
def test_match_launches_beside_foreign_queues(hog, B):
    import torch
    from ndt_slam_amd import capi, synth
    dev = torch.device("cuda", 0)
    lo, hi = -1, 0
    try:
        lo, hi = torch.cuda.Stream.priority_range()          # (least, greatest) where available
    except Exception:
        pass
    idle_before = torch.cuda.Stream(device=dev)
    low_before = torch.cuda.Stream(device=dev, priority=max(lo, hi))
    cfg = synth.CONFIGS["C3"]
    m = synth.make_map(cfg["n_map"], cfg["half"])
    sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
    scans, off, truths, inits = sf.batch(0, B)
    ctx = capi.Context(0)
    st = torch.cuda.Stream(device=dev)
    ctx.set_stream(st.cuda_stream)
    gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
    idle_after = torch.cuda.Stream(device=dev)
    low_after = torch.cuda.Stream(device=dev, priority=max(lo, hi))
    hog_stream = torch.cuda.Stream(device=dev)
    d_sc = torch.from_numpy(scans).to(dev); d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_in = torch.from_numpy(inits).to(dev)
    d_res = torch.zeros(B * capi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def launches(k):
        for _ in range(k):
            gm.align_batch_dev(d_sc.data_ptr(), d_off.data_ptr(), B, len(scans), d_in.data_ptr(), d_res.data_ptr(),
                               stream=st.cuda_stream)
        st.synchronize()
        times = [sum(ctx.kernel_timing(j)) for j in range(k)]
        res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=capi.RESULT_DTYPE)
        return times, res

    launches(3)                                              # warm-up
    t_alone, first = launches(20)
    assert np.all(first["status"] == 0)
    # (ii) streams that merely exist, plus a trickle of small kernels on the low-priority ones
    with torch.cuda.stream(low_before):
        junk = torch.ones(1 << 20, device=dev)
    t_quiet, r = launches(20)
    assert r.tobytes() == first.tobytes()
    # (i) 48 CUs held by a foreign kernel for 20 ms at a time, match launches back to back beside it
    t_hog = []
    for rnd in range(10):
        assert hog.hog_launch(hog_stream.cuda_stream, 48, 2_000_000, 150 * 1024, sink.data_ptr()) == 0
        with torch.cuda.stream(low_after):
            junk = junk * 1.0001
        t, r = launches(20)
        t_hog += t
        assert np.all(r["status"] == 0), "watchdog abort beside the foreign kernel"
        assert r.tobytes() == first.tobytes(), "records changed beside the foreign kernel (round %d)" % rnd
    torch.cuda.synchronize()
    a, q, h = float(np.median(t_alone)), float(np.median(t_quiet)), float(np.median(t_hog))
    print("B=%d " % B, end="")
    print("match + fitness per launch: alone %.3f ms, with idle / low-priority streams %.3f ms, beside a 48-CU foreign kernel "
          "median %.3f ms max %.3f ms (%d launches, slowdown %.2fx)" % (a, q, h, max(t_hog), len(t_hog), h / a))
    assert max(t_hog) < 50.0 and max(t_quiet) < 50.0
    assert q < 2.0 * a, "an idle stream must not double the launch time"
    del idle_before, idle_after


def test_a_launch_that_fails_behind_its_first_kernel_leaves_the_context_ordered():
    """Error path of ndt_align_batch_dev (round 5): NDT_OPT_INJECT_FAULT makes launch_align return NDT_E_HIP right behind the
    dispatch of the match kernel.  That kernel is queued and uses the context's scratch (control words, ordered copies), so
    the failing call must still close its scratch bracket: the next call on ANOTHER stream is ordered behind the orphan --
    without that its memset of the control words lands under a running persistent kernel.  The failed launch is also entered
    in the ring and in the map's readers: ndt_ctx_wait_launch finds it."""
    import torch
    from ndt_slam_amd import capi, synth
    dev = torch.device("cuda", 0)
    cfg = synth.CONFIGS["C3"]
    m = synth.make_map(cfg["n_map"], cfg["half"])
    sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
    B = 256
    scans, off, _, inits = sf.batch(0, B)
    ctx = capi.Context(0)
    s_a, s_b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ctx.set_stream(s_a.cuda_stream)
    gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
    d_sc = torch.from_numpy(scans).to(dev); d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_in = torch.from_numpy(inits).to(dev)
    d_ref = torch.zeros(B * capi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    d_junk, d_res = torch.zeros_like(d_ref), torch.zeros_like(d_ref)
    args = (d_sc.data_ptr(), d_off.data_ptr(), B, len(scans), d_in.data_ptr())
    gm.align_batch_dev(*args, d_ref.data_ptr(), stream=s_a.cuda_stream)
    torch.cuda.synchronize()
    ref = d_ref.cpu().numpy().tobytes()
    assert np.all(np.frombuffer(ref, dtype=capi.RESULT_DTYPE)["status"] == 0)
    for rnd in range(5):
        ctx.set_option(capi.OPT_INJECT_FAULT, 1)
        with pytest.raises(capi.NdtError, match="injected fault"):
            gm.align_batch_dev(*args, d_junk.data_ptr(), stream=s_a.cuda_stream)
        # at once, on another stream, with the same context: ordered behind the orphan kernel on s_a
        gm.align_batch_dev(*args, d_res.data_ptr(), stream=s_b.cuda_stream)
        ctx.wait_launch(1, s_b.cuda_stream)                   # the failed launch has a place in the ring
        s_b.synchronize()
        got = d_res.cpu().numpy().tobytes()
        assert got == ref, "round %d: records differ after a failed launch" % rnd
        # the orphan ran to its end (its records are the matches without a fitness score: the fitness kernels were not queued)
        s_a.synchronize()
        junk = np.frombuffer(d_junk.cpu().numpy().tobytes(), dtype=capi.RESULT_DTYPE)
        full = np.frombuffer(ref, dtype=capi.RESULT_DTYPE)
        assert np.all(junk["status"] == 0) and np.array_equal(junk["T03"], full["T03"]) and np.all(junk["fitness"] > 1e300)
    # a two-phase rebuild that has to be queued again waits for a failed launch like for any other reader
    ctx.set_option(capi.OPT_INJECT_FAULT, 0)
    gm.close(); ctx.close()
