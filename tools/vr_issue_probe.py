"""Is the ragged-grid (configs[3]) step bound by the host thread that issues it?  Times, for the bench's own vr workload, the host
time spent inside the calls that enqueue a step (no synchronisation inside) against the step's wall time, for one and two contexts,
and splits one call into Python glue and the C entry point.  Run on the GPU box: python tools/vr_issue_probe.py"""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import bench as B


def main():
    bn = B.Bench(torch.device("cuda:0"), 0, 1, 4)
    for streams in (1, 2, 3):
        wl = bn.vr(4096, 50000, streams)
        for _ in range(3):
            wl["step"]()
        torch.cuda.synchronize()
        n = 10
        issue = wall = 0.0
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            wl["step"]()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            issue += t1 - t0; wall += t2 - t0
        nb = wl["batches"]
        print(f"contexts={streams}: batches={nb}  issue {issue / n * 1e3:7.3f} ms/step ({issue / n / nb * 1e6:6.1f} us/batch)   "
              f"wall {wall / n * 1e3:7.3f} ms/step ({wall / n / nb * 1e6:6.1f} us/batch)   {wl['nodes_per_step'] * n / wall / 1e6:6.1f} M nodes/s")
        wl["close"]()


if __name__ == "__main__":
    main()
