#!/usr/bin/env python3
"""Where does the host time of the reference-shaped VR loop go?  cProfile of run_refinements (synchronous and pipelined) on the
synthetic 28 x 28 VR BAG bench.py's `processor_api` measurement uses, plus a micro-timing of CPU access to pinned memory."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.data import GraphBuilder, VRBagHandler
    from bathymetric_gnn_amd.models import BathymetricGNN
    from bathymetric_gnn_amd.scripts.inference_native import NativeVRProcessor, run_refinements
    dev = torch.device("cuda:0")
    # pinned vs pageable CPU access
    n = 1 << 20
    pin = torch.empty((n, 2), dtype=torch.float32, pin_memory=True).numpy()
    pag = np.empty((n, 2), np.float32)
    src = np.random.default_rng(0).random(n, dtype=np.float32)
    for name, buf in (("pinned", pin), ("pageable", pag)):
        t0 = time.perf_counter(); buf[:, 0] = src; t1 = time.perf_counter(); buf[:] = 1.0; t2 = time.perf_counter()
        c = np.array(buf); t3 = time.perf_counter()
        print(f"{name}: strided write {n * 4 / (t1 - t0) / 1e9:.2f} GB/s, fill {n * 8 / (t2 - t1) / 1e9:.2f} GB/s, read-copy {n * 8 / (t3 - t2) / 1e9:.2f} GB/s")
    sd = synthetic.synthetic_state_dict(in_channels=8, seed=1234)
    m = BathymetricGNN(in_channels=8, edge_dim=3, dropout=0.0)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    proc = NativeVRProcessor(m.to(dev).eval(), GraphBuilder(device=dev), dev)
    md, ref = synthetic.synthetic_vr_bag(28, 28, seed=4242)
    h = VRBagHandler.from_arrays(md, ref)
    for mode in (False, True):
        run_refinements(proc, h, h.copy_and_open_for_writing(), 0.0, pipelined=mode)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            w = h.copy_and_open_for_writing()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = run_refinements(proc, h, w, 0.0, pipelined=mode)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        dt = best
        print(f"pipelined={mode}: {dt * 1e3:.1f} ms, {st['cells_processed'] / dt / 1e6:.1f} M nodes/s, {st['grids_processed']} grids")
        pr = cProfile.Profile()
        pr.enable()
        run_refinements(proc, h, h.copy_and_open_for_writing(), 0.0, pipelined=mode)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
