"""Phase ablation of the fused layer kernels (diagnostic build only): the whole forward timed with parts of the kernel
switched off through the `diag_mask` context option.  Unlike the cycle stamps this does not perturb the kernel (no atomics
in the counted waits).  Bits: 1 gather, 2 MFMAs, 4 slab DMA, 8 W DMA, 32 phase A (attention coefficients), 64 final epilogue.

    python __graft_entry__.py --diag && BGNN_LIB=bathymetric-gnn_amd/libbgnn_hip_diag.so python tools/fused_ablation.py \
        [--connectivity 16-dilated --matrix-path bf16]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BGNN_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bathymetric-gnn_amd", "libbgnn_hip_diag.so"))
import numpy as np, torch
from bathymetric_gnn_amd import runtime as rt, synthetic
from bathymetric_gnn_amd.data import GraphBuilder
from bathymetric_gnn_amd.models import BathymetricGNN
from bathymetric_gnn_amd.models.pipeline import TileBatchEngine
dev = torch.device("cuda:0")
sd = synthetic.synthetic_state_dict(seed=1234)
model = BathymetricGNN(in_channels=7, edge_dim=3, dropout=0.0); model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}); model.to(dev).eval()
ap = argparse.ArgumentParser()
ap.add_argument("--connectivity", default="8-connected"); ap.add_argument("--matrix-path", default="exact_f32")
ap.add_argument("--lds-pad-kb", type=int, default=0, help="pad the fused kernels' LDS request (occupancy experiment: > 80 -> one workgroup per CU)")
ap.add_argument("--masks", default="", help="comma-separated diag_mask values instead of the standard ablation list")
args = ap.parse_args()
gb = GraphBuilder(device=dev, connectivity=args.connectivity); eng = TileBatchEngine(model, gb, dev)
eng.ctx.set_option("matrix_path", args.matrix_path)
if args.lds_pad_kb:
    eng.ctx.set_option("fused_lds_pad_kb", args.lds_pad_kb)
B, S = 128, 256
depth, mask, _ = synthetic.synthetic_tile_batch(8, S, S, 100, "V0"); depth = np.concatenate([depth] * 16); mask = np.concatenate([mask] * 16)
d_t = torch.from_numpy(depth).to(dev).reshape(-1); m_t = torch.from_numpy(mask.view(np.uint8)).to(dev).reshape(-1)
hw = np.tile(np.array([[S, S]], np.int32), (B, 1)); res = np.full((B, 2), 0.5)
out = torch.empty((3, d_t.numel()), device=dev)
names = {1: "gather", 2: "mfma", 4: "slab-dma", 8: "w-dma", 32: "phaseA", 64: "epilogue"}
masks = [int(x) for x in args.masks.split(",")] if args.masks else [0, 1, 2, 3, 4 + 8, 2 + 4 + 8, 1 + 2 + 4 + 8, 1 + 2 + 4 + 8 + 32,
                                                                     1 + 2 + 4 + 8 + 32 + 64, 64, 32, 2 + 64, 1 + 2 + 64]
for mask_bits in masks:
    eng.ctx.set_option("diag_mask", mask_bits)
    for _ in range(2):
        eng.infer_device(hw, res, d_t, m_t, None, out=out)
    torch.cuda.synchronize()
    eng.ctx.profile(["fused"])
    for _ in range(5):
        eng.infer_device(hw, res, d_t, m_t, None, out=out)
    p = eng.ctx.profile_read()["fused"]
    eng.ctx.profile([])
    off = "+".join(v for k, v in names.items() if mask_bits & k) or "nothing"
    print(f"without {off:45s} fused {p['ms'] / 5:7.2f} ms/step")
eng.ctx.set_option("diag_mask", 0)
