#!/usr/bin/env python3
"""Summary of tools/collect_mfma_util.sh (gpurun_out/prof_util/{exact,c3}_{MfmaUtil,VALUBusy}): per kernel instance the derived
metric averaged over the full-batch launches (largest grid seen for that kernel) of `bench.py --no-extras`, stamped with the
build id of the library the passes ran on (gpurun_out/prof_util/build_id.txt, written by the collector).
    python tools/summarise_mfma_util.py --round 5      ->  profiles/r05_mfma_valu_util.json"""
import argparse, collections, csv, glob, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser(); ap.add_argument("--round", type=int, required=True); ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out", "prof_util"))
a = ap.parse_args()
stamp = None
for cand in (os.path.join(a.src, "build_id.txt"), os.path.join(ROOT, "gpurun_out", "prof_final", "build_id.txt")):
    if os.path.exists(cand):
        stamp = open(cand).read().strip(); break
out = {"_note": "rocprofv3 --kernel-trace --pmc MfmaUtil / --pmc VALUBusy (separate passes, tools/collect_mfma_util.sh), average over the "
                "full-batch launches of bench.py --no-extras (exact = the headline workload, c3 = --workload c3)",
       "build_id": stamp}
for mode in ("exact", "c3"):
    for ctr in ("MfmaUtil", "VALUBusy"):
        acc = collections.defaultdict(list)
        files = glob.glob(os.path.join(a.src, f"{mode}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
        # (gpurun MERGES a run's output into gpurun_out/: an earlier run's file may sit beside the new one -- newest only)
        files = sorted(files, key=os.path.getmtime)[-1:]
        for f in files:
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != ctr:
                    continue
                k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("bgnn::", ""); k = re.sub(r"\(.*\)$", "", k)
                if k.startswith(("gat_", "gemm_", "features", "stats_")):
                    acc[k].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
        for k, v in acc.items():
            gmax = max(g for g, _ in v); vals = [x for g, x in v if g == gmax]
            out.setdefault(f"{mode}:{k}", {})[ctr] = round(sum(vals) / len(vals), 2)
dst = os.path.join(ROOT, "profiles", f"r{a.round:02d}_mfma_valu_util.json")
json.dump(out, open(dst, "w"), indent=1); print(open(dst).read())
