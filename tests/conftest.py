import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def golden_names(full_only=False):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    names = [n for n in names if not n.startswith(("tiling_", "model_"))]
    if full_only:
        names = [n for n in names if not n.startswith("C2_")]
    return names


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["mask_bool"] = g["mask"].astype(bool)
    g["mask_arg"] = g["mask_bool"] if bool(g["mask_given"]) else None
    g["unc_arg"] = g.get("unc")
    g["res"] = (float(g["resolution"][0]), float(g["resolution"][1]))
    g["conn"] = str(g["connectivity"])
    g["loops"] = bool(g["self_loops"])
    return g


def ulp_diff_f32(a, b):
    """Distance in float32 ulps between two float32 arrays (same shape)."""
    a = np.ascontiguousarray(a, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, np.int64(-2 ** 31) - ai, ai)
    bi = np.where(bi < 0, np.int64(-2 ** 31) - bi, bi)
    return np.abs(ai - bi)


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
