import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from bathymetric_gnn_amd import runtime as rt, synthetic
from bathymetric_gnn_amd.data import GraphBuilder
from bathymetric_gnn_amd.models import BathymetricGNN
from bathymetric_gnn_amd.models.pipeline import TileBatchEngine
dev = torch.device("cuda:0")
sd = synthetic.synthetic_state_dict(in_channels=8, seed=1234)
model = BathymetricGNN(in_channels=8, edge_dim=3, dropout=0.0); model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}); model.to(dev).eval()
gb = GraphBuilder(device=dev)
grids = synthetic.vr_grid_stream(4096, seed0=1000)
batches, cur, n = [], [], 0
for d, u, r in grids:
    m = (d != 1.0e6) & np.isfinite(d); cur.append((d, m, u, r)); n += int(m.sum())
    if n >= 50000: batches.append(cur); cur, n = [], 0
if cur: batches.append(cur)
devb = []
for b in batches:
    hw, res, d, m, u = gb.upload_tiles([x[0] for x in b], [x[1] for x in b], [x[2] for x in b], [x[3] for x in b])
    devb.append((hw, res, d, m, u, torch.empty((3, d.numel()), device=dev)))
for ns in (1, 2, 3, 4, 6):
    engs = [TileBatchEngine(model, gb, dev, ctx=rt.new_context(dev)) for _ in range(ns)]
    def step():
        for i, (hw, res, d, m, u, o) in enumerate(devb):
            engs[i % ns].infer_device(hw, res, d, m, u, out=o, defer_end=True)
        for e in engs: e.ctx.end()
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(ns, "streams: host enqueue %.2f ms, total %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
