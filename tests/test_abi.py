"""The C-ABI library loads without a GPU and exports every symbol include/bgnn.h declares; the ctypes
binding covers exactly that set; structure layouts agree with the header."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "bgnn.h")).read()


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(bgnn_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from bathymetric_gnn_amd import runtime
    if not os.path.exists(runtime.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return runtime.load_library()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from bathymetric_gnn_amd import runtime
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in bgnn.h but not exported"
    assert sorted(runtime._SIGNATURES) == syms
    assert lib.bgnn_abi_version() == int(re.search(r"#define BGNN_ABI_VERSION (\d+)", HEADER).group(1))


def test_build_id_is_the_kernel_source_hash(lib):
    """bgnn_build_id() of a library built by __graft_entry__.build() = the hash of the kernel sources in the tree = what bench.py
    compares profiles/pmc_traffic.json's stamp against before it attaches counter traffic to a roofline."""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__
    import bench
    from bathymetric_gnn_amd import runtime
    assert __graft_entry__.kernel_source_sha() == bench.kernel_source_sha()
    __graft_entry__.build()                       # no-op when up to date
    assert runtime.build_id() == bench.kernel_source_sha() and len(runtime.build_id()) == 16


def test_struct_layouts():
    from bathymetric_gnn_amd import runtime as rt
    assert C.sizeof(rt.ModelDesc) == 36
    assert C.sizeof(rt.GraphOpts) == 4 * (2 + 1 + 8 + 1 + 4)
    assert C.sizeof(rt.Outputs) == 9 * C.sizeof(C.c_void_p)
    assert C.sizeof(rt.Tiles) == 8 + 5 * C.sizeof(C.c_void_p)
    n_k = int(re.search(r"BGNN_K_COUNT = (\d+)", HEADER).group(1))
    assert len(rt.K_NAMES) == n_k


def test_weight_count_matches_reference_parameter_count(lib):
    """182 469 parameters for in=8 (docs/QUICK_REFERENCE.md:185 says '182K'); the library's blob adds
    the BatchNorm running statistics (2 x 832 floats), which are buffers, not parameters."""
    from bathymetric_gnn_amd import runtime as rt
    from bathymetric_gnn_amd.models import BathymetricGNN
    m = BathymetricGNN(in_channels=8, edge_dim=3)
    assert sum(p.numel() for p in m.parameters()) == 182469
    d = m._desc()
    assert lib.bgnn_model_weight_count(C.byref(d)) == 182469 + 2 * (256 * 3 + 64) == m.pack_weights().size


def test_errors_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.bgnn_ctx_create(0, None, C.byref(h))
    assert rc != 0 and lib.bgnn_last_error()
    from bathymetric_gnn_amd import runtime as rt
    from bathymetric_gnn_amd.data import GraphBuilder
    import numpy as np
    with pytest.raises(rt.BgnnError):                 # no CPU fallback: fail loudly
        GraphBuilder().build_graph(np.zeros((4, 4), np.float32))
    with pytest.raises(ValueError):
        GraphBuilder(connectivity="6-connected")


@pytest.mark.parametrize("kind,per_layer", [("GCN", 64 * 64 + 64), ("GraphSAGE", 2 * 64 * 64 + 64), ("GIN", 2 * (64 * 64 + 64))])
def test_weight_count_other_backbones(kind, per_layer, lib):
    """Blob sizes of the GCN / GraphSAGE / GIN backbones (include/bgnn.h lists the order)."""
    from bathymetric_gnn_amd.models import BathymetricGNN
    m = BathymetricGNN(in_channels=7, gnn_type=kind, num_gnn_layers=3)
    d = m._desc()
    fe = 64 * 7 + 64 + 64 * 64 + 64
    heads = (32 * 64 + 32 + 3 * 32 + 3) + 2 * (32 * 64 + 32 + 32 + 1)
    assert lib.bgnn_model_weight_count(C.byref(d)) == fe + 3 * (per_layer + 4 * 64) + heads == m.pack_weights().size


def test_header_is_plain_c_and_library_loads_from_c(tmp_path):
    """include/bgnn.h compiles as C99 (-pedantic) and a C program resolves every declared entry point with dlsym."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "c_abi_smoke"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_abi_smoke.c"), "-o", str(exe), "-ldl"], check=True)
    r = subprocess.run([str(exe), os.path.join(root, "bathymetric-gnn_amd", "libbgnn_hip.so")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok")
    declared = set(declared_symbols())
    src = open(os.path.join(root, "tests", "c_abi_smoke.c")).read()
    assert declared == set(re.findall(r"RESOLVE\((bgnn_[a-z_0-9]+)\)", src))
