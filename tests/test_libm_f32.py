"""ndt_params::libm_f32 = 1: the float32 cos / sin of a trial's yaw as glibc computes them (what Eigen's AngleAxisf calls on the
reference's platform), restated for the device in ndt_slam_amd/csrc/ndt_libm_f32.hip.h.  Its C twin against this machine's
libm on EVERY float with |x| < 120; the device version against libm on a sample (-m gpu, through ndt_eval_at's transform is
indirect -- the direct check is tests/test_gpu_parity.py::test_device_cos_sin_are_the_platforms)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _glibc_at_least_228():
    import platform
    lib, ver = platform.libc_ver()
    try:
        major, minor = (int(v) for v in ver.split(".")[:2])
    except ValueError:
        return False
    return lib == "glibc" and (major, minor) >= (2, 28)


@pytest.mark.skipif(not _glibc_at_least_228(), reason="the restatement is glibc >= 2.28's sinf / cosf")
def test_twin_equals_libm_on_every_float_below_120(tmp_path):
    exe = str(tmp_path / "twin")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-DUSE_FMA", "-mfma",
                           os.path.join(ROOT, "tests", "libm_f32_twin.c"), "-o", exe, "-lm"])
    out = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    assert out.stdout.split()[:4] == ["sinf", "0", "cosf", "0"] and int(out.stdout.split()[-1]) == 2246049792, out.stdout


def test_twin_source_and_device_source_hold_the_same_constants():
    """The two files are kept in step by hand: every hexadecimal double of the device header appears in the twin."""
    import re
    dev = open(os.path.join(ROOT, "ndt_slam_amd", "csrc", "ndt_libm_f32.hip.h")).read()
    twin = open(os.path.join(ROOT, "tests", "libm_f32_twin.c")).read().lower()
    glibc_part = dev[:dev.index("sincos_small")]                 # (the short fp64 sincos behind it has a twin of its own)
    consts = set(c.lower() for c in re.findall(r"0x1\.[0-9a-fA-F]+p[+-]?\d+", glibc_part))
    assert len(consts) >= 9
    for c in consts:
        assert c.replace("p+", "p") in twin.replace("p+", "p"), c


def test_short_fp64_sincos_is_within_one_ulp_and_rounds_to_the_same_floats(tmp_path):
    """sincos_small (the optimiser step's fp64 cos / sin of a yaw: ~50 instructions instead of the device library's ~240 on
    the one lane every pass waits for): its C twin against libm."""
    exe = str(tmp_path / "sc")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", os.path.join(ROOT, "tests", "sincos_small_twin.c"),
                           "-o", exe, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    assert "max ulp sin 1 cos 1" in out.stdout or "max ulp sin 0" in out.stdout, out.stdout
    assert out.stdout.strip().endswith(": 0"), out.stdout
    dev = open(os.path.join(ROOT, "ndt_slam_amd", "csrc", "ndt_libm_f32.hip.h")).read()
    twin = open(os.path.join(ROOT, "tests", "sincos_small_twin.c")).read()
    for c in ("1.66666666666666324348e-01", "1.58969099521155010221e-10", "1.13596475577881948265e-11", "0x1.1a62633145c07p-54"):
        assert c in dev and c in twin


def _glibc_before_241():
    import platform
    lib, ver = platform.libc_ver()
    try:
        major, minor = (int(v) for v in ver.split(".")[:2])
    except ValueError:
        return False
    return lib == "glibc" and (major, minor) < (2, 41)          # (2.41 switched these to correctly rounded CORE-MATH code)


@pytest.mark.skipif(not _glibc_before_241(), reason="the restatement is the fdlibm-style float atanf / atan2f of glibc < 2.41")
def test_atan2f_twin_equals_libm(tmp_path):
    exe = str(tmp_path / "at")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", os.path.join(ROOT, "tests", "atan2f_twin.c"), "-o", exe, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout
    assert "atanf mismatches (every 3rd float): 0" in out.stdout and "atan2f mismatches: 0 of" in out.stdout, out.stdout
    dev = open(os.path.join(ROOT, "ndt_slam_amd", "csrc", "ndt_libm_f32.hip.h")).read()
    twin = open(os.path.join(ROOT, "tests", "atan2f_twin.c")).read()
    for c in ("4.6364760399e-01f", "3.3333334327e-01f", "1.6285819933e-02f", "-8.7422776573e-08f", "7.5497894159e-08f"):
        assert c in dev and c in twin, c
