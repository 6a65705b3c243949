"""Row b: the boundary is a C ABI -- include/ndt_mi355x.h must compile as plain C99 (no C++, no torch, no HIP types in the
signatures) and a C program must be able to link the shared library, fill the parameter presets and be told, without a
GPU, that there is no device (the library never falls back to the CPU).  The struct sizes the C compiler sees are the
ones the ctypes binding assumes."""
import ctypes
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAM = r'''
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "ndt_mi355x.h"

int main(void) {
  ndt_params p, q, r;
  ndt_fuse_params f;
  ndt_ctx *ctx = NULL;
  int rc;
  memset(&p, 0x5a, sizeof p);
  if (ndt_default_params(&p) != NDT_OK || ndt_params_pcl18(&q) != NDT_OK || ndt_params_pcl_new(&r) != NDT_OK) return 2;
  if (ndt_fuse_default_params(&f) != NDT_OK) return 3;
  if (ndt_default_params(NULL) != NDT_E_ARG) return 4;
  printf("sizes %zu %zu %zu\n", sizeof(ndt_params), sizeof(ndt_result), sizeof(ndt_map_info));
  printf("offsets %zu %zu %zu %zu\n", offsetof(ndt_params, grid_margin), offsetof(ndt_params, libm_f32),
         offsetof(ndt_result, fitness), offsetof(ndt_result, flags));
  printf("preset %g %d %d %d %d %d %d | %d %d | %d %d\n", (double)p.resolution, p.max_iter, p.cov_init_identity,
         p.cov_unbiased, p.transform_sse, p.libm_f32, p.grid_margin, q.transform_sse, q.libm_f32, r.cov_unbiased,
         r.cov_init_identity);
  rc = ndt_ctx_create(0, &ctx);
  printf("ctx %d %s\n", rc, rc == NDT_OK ? "ok" : ndt_last_error(NULL));
  if (rc == NDT_OK) ndt_ctx_destroy(ctx);
  else if (ctx != NULL) return 5;
  return 0;
}
'''


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_header_is_c99_and_a_c_program_links_the_library(tmp_path):
    from ndt_slam_amd import build, capi
    lib = build.build()
    src = tmp_path / "user.c"
    src.write_text(PROGRAM)
    exe = str(tmp_path / "user")
    libdir = os.path.dirname(lib)
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        str(src), "-o", exe, "-L" + libdir, "-lndt_mi355x", "-Wl,-rpath," + libdir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr[-2000:])
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    sizes = [int(v) for v in lines["sizes"].split()]
    assert sizes == [ctypes.sizeof(capi.Params), capi.RESULT_DTYPE.itemsize, ctypes.sizeof(capi.MapInfo)]
    offs = [int(v) for v in lines["offsets"].split()]
    assert offs == [capi.Params.grid_margin.offset, capi.Params.libm_f32.offset,
                    capi.RESULT_DTYPE.fields["fitness"][1], capi.RESULT_DTYPE.fields["flags"][1]]
    # the presets as the header documents them: PCL 1.10 default, <= 1.8, >= 1.11
    assert lines["preset"].split("|")[0].split() == ["1", "35", "1", "0", "1", "1", "0"]
    assert lines["preset"].split("|")[1].split() == ["0", "0"] and lines["preset"].split("|")[2].split() == ["1", "0"]
    import torch
    if torch.cuda.is_available():
        assert lines["ctx"].startswith("0 ok")
    else:
        assert lines["ctx"].split()[0] == str(capi.NDT_E_NO_DEVICE), lines["ctx"]      # no device: refused, no CPU path
