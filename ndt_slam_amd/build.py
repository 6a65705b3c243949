"""Builds libndt_mi355x.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "ndt_mi355x.hip")
OUT = os.path.join(HERE, "libndt_mi355x.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
         "-ffp-contract=off",            # explicit fma only (see the header of the .hip file)
         "-fhip-fp32-correctly-rounded-divide-sqrt",   # (the default, spelled out: ndt_libm_f32.hip.h restates float code of Eigen / glibc)
         "-fno-fast-math", "-fgpu-rdc" if False else "-fno-gpu-rdc",
         "-I" + os.path.join(ROOT, "include")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    csrc = os.path.dirname(SRC)
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
    deps += [os.path.join(ROOT, "include", "ndt_mi355x.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=(), out=None):
    """out: another output path (diagnostic / experimental variants, loaded through NDT_LIB_PATH)."""
    if out is None and not force and not needs_build():
        return OUT
    cmd = [HIPCC] + FLAGS + list(extra) + ["-o", out or OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out or OUT


if __name__ == "__main__":
    # python -m ndt_slam_amd.build [--force] [--usage] [--out PATH] [-DNAME[=V] ...] [--flag=<hipcc flag> ...]
    argv = sys.argv[1:]
    out = argv[argv.index("--out") + 1] if "--out" in argv else None
    extra = [a for a in argv if a.startswith("-D")]
    for a in argv:                                     # experiments: --flag=-mllvm --flag=-amdgpu-... passed through to hipcc
        if a.startswith("--flag="):
            extra.append(a[len("--flag="):])
    if "--usage" in argv:
        extra.append("-Rpass-analysis=kernel-resource-usage")
    print(build(force="--force" in argv, verbose=True, extra=extra, out=out))
