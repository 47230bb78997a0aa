"""Per-phase cycle sums of the fused layer kernel.  Needs the diagnostic build of the library:
    python __graft_entry__.py --diag && BGNN_LIB=bathymetric-gnn_amd/libbgnn_hip_diag.so python tools/fused_stamps.py"""
import argparse, os, sys, ctypes as C
os.environ.setdefault("BGNN_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bathymetric-gnn_amd", "libbgnn_hip_diag.so"))
os.environ["BGNN_FUSED_STAMPS"]="1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bathymetric_gnn_amd import runtime as rt, synthetic
from bathymetric_gnn_amd.data import GraphBuilder
from bathymetric_gnn_amd.models import BathymetricGNN
from bathymetric_gnn_amd.models.pipeline import TileBatchEngine
dev=torch.device("cuda:0")
sd = synthetic.synthetic_state_dict(seed=1234)
model = BathymetricGNN(in_channels=7, edge_dim=3, dropout=0.0); model.load_state_dict({k: torch.as_tensor(v) for k,v in sd.items()}); model.to(dev).eval()
ap = argparse.ArgumentParser()
ap.add_argument("--connectivity", default="8-connected"); ap.add_argument("--matrix-path", default="exact_f32")
ap.add_argument("--persistent", action="store_true", help="the 256 -> 256 exact instance in its persistent form (own counters)")
args = ap.parse_args()
gb=GraphBuilder(device=dev, connectivity=args.connectivity); eng=TileBatchEngine(model, gb, dev)
eng.ctx.set_option("matrix_path", args.matrix_path)
if args.persistent:
    eng.ctx.set_option("fused_persistent", 1)
B,S=128,256
depth,mask,_=synthetic.synthetic_tile_batch(8,S,S,100,"V0"); depth=np.concatenate([depth]*16); mask=np.concatenate([mask]*16)
d_t=torch.from_numpy(depth).to(dev).reshape(-1); m_t=torch.from_numpy(mask.view(np.uint8)).to(dev).reshape(-1)
hw=np.tile(np.array([[S,S]],np.int32),(B,1)); res=np.full((B,2),0.5)
eng.infer_device(hw,res,d_t,m_t,None); torch.cuda.synchronize()
lib=rt.load_library(); lib.bgnn_debug_stamps.argtypes=[C.c_void_p, C.POINTER(C.c_uint64)]
buf=(C.c_uint64*64)(); lib.bgnn_debug_stamps(eng.ctx.handle, buf)
eng.infer_device(hw,res,d_t,m_t,None); torch.cuda.synchronize()
lib.bgnn_debug_stamps(eng.ctx.handle, buf)
n=buf[15]; names=["slab 0 DMA issue","halo-table barrier","phase A","wait slab+bar","gather","wait W+bar","MFMA(+slab issue)","bar+W issue","final epilogue",
                  "decode + round 1 issue", "round 1 wait", "round 2 (+ table writes)"]
NS=len(names)
tot=sum(buf[i] for i in range(NS))
print("blocks",n, "total cycles/block", tot/n)
for i,nm in enumerate(names): print("%-20s %10.0f cycles/block  %5.1f%%"%(nm, buf[i]/n, 100*buf[i]/tot))
if buf[14]:
    print("in-kernel clock: %.0f MHz (s_memtime / s_memrealtime x 100 MHz over every workgroup's lifetime)" % (100.0 * buf[13] / buf[14]))
    print("workgroup lifetime: %.1f us" % (buf[14] / n / 100.0))

if buf[31]:
    # the persistent form of the 256 -> 256 instance: timers summed in registers, one set of atomics per workgroup (unperturbed)
    pn = buf[31]; pnames = ["block-start wait + barrier", "phase A", "slab wait + barrier", "prefetch hooks", "gather + BN/ReLU", "MFMA (+ DMA requests)", "pre-epilogue barrier", "epilogue"]
    order = [0, 1, 2, 3, 4, 5, 6] ; vals = [buf[16 + 0], buf[16 + 1], buf[16 + 2] , buf[16 + 3], buf[16 + 4], buf[16 + 5], buf[16 + 6], buf[16 + 7]]
    lab = {0: "block-start wait + barrier", 1: "phase A", 2: "slab wait + barrier + prefetch hooks", 3: "gather + BN/ReLU", 4: "MFMA (+ DMA requests)", 5: "pre-epilogue wait + barrier", 6: "epilogue", 7: "loop bookkeeping"}
    ptot = sum(vals)
    print("persistent kernel: blocks", pn, "cycles/block", ptot / pn)
    for i in range(8): print("  %-40s %10.0f cycles/block  %5.1f%%" % (lab[i], vals[i] / pn, 100 * vals[i] / ptot))

for base, title in ((32, "256 -> 64 instance"), (48, "heads instance")):
    if buf[base + 15]:
        nb = buf[base + 15]; tt = sum(buf[base + i] for i in range(NS))
        print(title, "blocks", nb, "cycles/block", tt / nb)
        for i, nm in enumerate(names): print("  %-20s %10.0f cycles/block  %5.1f%%" % (nm, buf[base + i] / nb, 100 * buf[base + i] / tt))

if buf[15] > buf[47] + buf[63]:
    nb = buf[15] - buf[47] - buf[63]; vals = [buf[i] - buf[32 + i] - buf[48 + i] for i in range(NS)]; tt = sum(vals)
    print("256 -> 256 instances (by difference) blocks", nb, "cycles/block", tt / nb)
    for i, nm in enumerate(names): print("  %-20s %10.0f cycles/block  %5.1f%%" % (nm, vals[i] / nb, 100 * vals[i] / tt))
