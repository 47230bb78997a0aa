#!/usr/bin/env python3
"""Per-call wall times inside the pipelined VR loop (which call stalls, and when?), with and without the garbage collector."""
import gc
import os
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
    from bathymetric_gnn_amd.scripts import inference_native as inn
    dev = torch.device("cuda:0")
    sd = synthetic.synthetic_state_dict(in_channels=8, seed=1234)
    m = BathymetricGNN(in_channels=8, edge_dim=3, dropout=0.0)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    proc = inn.NativeVRProcessor(m.to(dev).eval(), GraphBuilder(device=dev), dev)
    md, ref = synthetic.synthetic_vr_bag(28, 28, seed=4242)
    h = VRBagHandler.from_arrays(md, ref)
    log = []

    def wrap(obj, name):
        f = getattr(obj, name)

        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            log.append((name, (time.perf_counter() - t0) * 1e3))
            return r
        setattr(obj, name, g)

    for name in ("_launch", "_finish", "add_to_batch"):
        wrap(proc, name)
    # finer: the statements of _launch
    orig_copy = torch.Tensor.copy_

    def timed_copy(self, src, non_blocking=False):
        t0 = time.perf_counter()
        r = orig_copy(self, src, non_blocking=non_blocking)
        log.append((f"copy_ {'D2H' if self.device.type == 'cpu' else 'H2D'} contiguous={self.is_contiguous()}", (time.perf_counter() - t0) * 1e3))
        return r
    torch.Tensor.copy_ = timed_copy
    for label, prep in (("gc on", lambda: None), ("gc off", gc.disable)):
        prep()
        for rep in range(3):
            log.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = inn.run_refinements(proc, h, h.copy_and_open_for_writing(), 0.0, pipelined=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            big = [(n, round(t, 2)) for n, t in log if t > 1.0]
            tot = {}
            for n, t in log:
                tot[n] = tot.get(n, 0.0) + t
            print(f"[{label}] run {rep}: {dt:.1f} ms; totals " + ", ".join(f"{k} {v:.1f}" for k, v in tot.items()) + f"; calls > 1 ms: {big}")


if __name__ == "__main__":
    main()
