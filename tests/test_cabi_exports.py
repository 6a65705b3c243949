"""The C-ABI library builds for gfx950 here (no GPU) and exports every symbol the header
declares; without a device every entry point refuses instead of falling back to a CPU path."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from ndt_slam_amd import build
    return build.build()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "ndt_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ndt_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("ndt_ctx_create", "ndt_map_build", "ndt_align", "ndt_align_batch", "ndt_align_batch_dev",
                 "ndt_eval_at", "ndt_fitness_at", "ndt_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(built):
    L = ctypes.CDLL(built)
    for name in declared_functions():
        assert hasattr(L, name), name
    from ndt_slam_amd import capi
    assert sorted(capi.EXPORTS) == declared_functions()


def test_struct_layouts_match_between_binding_and_oracle(built):
    from ndt_slam_amd import capi
    from oracle import ndt_oracle as O
    assert ctypes.sizeof(capi.Params) == ctypes.sizeof(O.Params)
    assert [f[0] for f in capi.Params._fields_] == [f[0] for f in O.Params._fields_]
    assert capi.RESULT_DTYPE == O.RESULT_DTYPE
    p = capi.default_params(); q = O.default_params()
    assert bytes(p) == bytes(q)


def test_no_cpu_fallback_without_a_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from ndt_slam_amd import capi
    with pytest.raises(capi.NdtError):
        capi.Context(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ndt_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".c")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "ndt_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
