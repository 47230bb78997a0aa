#!/usr/bin/env python3
"""Summary of tools/collect_sq_counters.sh: per fused / GEMM kernel instance, every counter averaged over the full-batch launches and,
for the SQ cycle counters, as a share of SQ_WAVE_CYCLES.   python tools/summarise_sq_counters.py c3 > profiles/rNN_sq_counters_c3.json"""
import collections, csv, glob, json, os, re, sys
mode = sys.argv[1] if len(sys.argv) > 1 else "c3"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
files = glob.glob(os.path.join(root, "gpurun_out", f"prof_sq_{mode}", "**", "*counter_collection.csv"), recursive=True)
newest_of = {}
for f in files:                              # (one file per counter group; an earlier run's file merged beside it is dropped)
    g = os.path.relpath(f, os.path.join(root, "gpurun_out", f"prof_sq_{mode}")).split(os.sep)[0]
    if g not in newest_of or os.path.getmtime(f) > os.path.getmtime(newest_of[g]):
        newest_of[g] = f
for f in newest_of.values():
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("bgnn::", ""); k = re.sub(r"\(.*\)$", "", k)
        if "fused" in k or "gemm" in k or "2p_kernel" in k or "extractor" in k:
            acc[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
bid = os.path.join(root, "gpurun_out", f"prof_sq_{mode}", "build_id.txt")
out = {"build_id": open(bid).read().strip() if os.path.exists(bid) else None,
       "_note": f"rocprofv3 --kernel-trace --pmc (a few counters per pass, tools/collect_sq_counters.sh {mode}); full-batch launches of "
                "bench.py --no-extras; 'share' = counter / SQ_WAVE_CYCLES for the SQ cycle counters"}
for k, cs in acc.items():
    row = {}
    for c, v in cs.items():
        gmax = max(g for g, _ in v); vals = [x for g, x in v if g == gmax]; row[c] = sum(vals) / len(vals)
    wc = row.get("SQ_WAVE_CYCLES", 0)
    out[k] = {c: ({"value": v, "share_of_wave_cycles": round(v / wc, 4)} if wc and c.startswith("SQ_") and "INSTS" not in c and c != "SQ_WAVE_CYCLES" else v)
              for c, v in sorted(row.items())}
print(json.dumps(out, indent=1))
