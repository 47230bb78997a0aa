"""Ablation of the front GEMM (gemm_wres64_kernel, diagnostic build only): the launch timed with its row stores (1), its MFMAs (2)
or its X loads (4) switched off through the `gemm_diag` context option.

    python __graft_entry__.py --diag && BGNN_LIB=bathymetric-gnn_amd/libbgnn_hip_diag.so python tools/gemm_ablation.py [--matrix-path bf16]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BGNN_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bathymetric-gnn_amd", "libbgnn_hip_diag.so"))
import numpy as np, torch
from bathymetric_gnn_amd import synthetic
from bathymetric_gnn_amd.data import GraphBuilder
from bathymetric_gnn_amd.models import BathymetricGNN
from bathymetric_gnn_amd.models.pipeline import TileBatchEngine
ap = argparse.ArgumentParser(); ap.add_argument("--matrix-path", default="exact_f32"); args = ap.parse_args()
dev = torch.device("cuda:0")
sd = synthetic.synthetic_state_dict(seed=1234)
model = BathymetricGNN(in_channels=7, edge_dim=3, dropout=0.0); model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}); model.to(dev).eval()
gb = GraphBuilder(device=dev); eng = TileBatchEngine(model, gb, dev)
eng.ctx.set_option("matrix_path", args.matrix_path)
B, S = 128, 256
depth, mask, _ = synthetic.synthetic_tile_batch(8, S, S, 100, "V0"); depth = np.concatenate([depth] * 16); mask = np.concatenate([mask] * 16)
d_t = torch.from_numpy(depth).to(dev).reshape(-1); m_t = torch.from_numpy(mask.view(np.uint8)).to(dev).reshape(-1)
hw = np.tile(np.array([[S, S]], np.int32), (B, 1)); res = np.full((B, 2), 0.5)
out = torch.empty((3, d_t.numel()), device=dev)
for bits, name in ((0, "nothing"), (1, "row stores"), (2, "MFMAs"), (4, "X loads"), (3, "stores + MFMAs"), (6, "MFMAs + X loads")):
    eng.ctx.set_option("gemm_diag", bits)
    for _ in range(2):
        eng.infer_device(hw, res, d_t, m_t, None, out=out)
    torch.cuda.synchronize()
    eng.ctx.profile(["gemm"])
    for _ in range(5):
        eng.infer_device(hw, res, d_t, m_t, None, out=out)
    p = eng.ctx.profile_read()["gemm"]
    eng.ctx.profile([])
    print(f"without {name:20s} front GEMM {p['ms'] / 5:7.3f} ms/step")
eng.ctx.set_option("gemm_diag", 0)
