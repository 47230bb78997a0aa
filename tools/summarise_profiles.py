#!/usr/bin/env python3
"""Turn the raw output of tools/collect_profiles.sh (gpurun_out/prof_final/) into the committed files under profiles/:

  r<NN>_bench_*.json            the bench lines of the evidence run
  r<NN>_{mode}_kernel_stats.csv rocprofv3 --kernel-trace --stats summaries (copied unchanged)
  r<NN>_pmc_traffic_all.json    per kernel instance: FETCH_SIZE / WRITE_SIZE per launch
  pmc_traffic.json              per kernel CLASS (what bench.py attaches as roofline.traffic)

Counter handling is the one /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes, both are in KiB, and FETCH_SIZE is doubled on gfx950.  Only full-batch launches are averaged
(the launches whose grid is the largest one seen for that kernel): the bench also runs single-tile forwards.

usage: python tools/summarise_profiles.py [--round 1] [--src gpurun_out/prof_final]
"""
import argparse
import csv
import glob
import json
import os
import re
import shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("bgnn::", "")
    return re.sub(r"\(.*\)$", "", name)


def newest(files, window_s=300):
    """gpurun MERGES a run's output into the local gpurun_out/: files of an earlier round's run may sit beside the new ones.
    Only the files of the most recent run (modified within five minutes of the newest file of the same pass) are summarised."""
    if not files:
        return files
    t = max(os.path.getmtime(f) for f in files)
    return sorted((f for f in files if os.path.getmtime(f) >= t - window_s), key=os.path.getmtime, reverse=True)


def per_launch(counter_dir, counter):
    """kernel -> [(grid, KiB)] from one --pmc pass (one row per dispatch and counter; rows of one dispatch are summed)"""
    files = newest(glob.glob(os.path.join(counter_dir, "**", "*counter_collection.csv"), recursive=True))
    by_dispatch = defaultdict(float)
    meta = {}
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                key = (f, r["Dispatch_Id"])
                by_dispatch[key] += float(r["Counter_Value"])
                meta[key] = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
    out = defaultdict(list)
    for key, v in by_dispatch.items():
        k, g = meta[key]
        out[k].append((g, v))
    return out


def full_batch_mean(samples):
    gmax = max(g for g, _ in samples)
    vals = [v for g, v in samples if g == gmax]
    return sum(vals) / len(vals), len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", type=int, default=1)
    ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out", "prof_final"))
    a = ap.parse_args()
    tag = f"r{a.round:02d}"
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)

    for src_name, dst_name in (("bench_fused", "bench_fused"), ("bench_unfused", "bench_unfused"), ("bench_split", "bench_split"),
                               ("bench_split_f16", "bench_split_f16"), ("bench_vr_50k", "bench_vr_budget50k"),
                               ("bench_vr_50k_1ctx", "bench_vr_budget50k_1ctx"), ("bench_vr_50k_2ctx", "bench_vr_budget50k_2ctx"),
                               ("bench_survey_20000", "bench_survey_20000"),
                               ("bench_vr_1M", "bench_vr_budget1M"), ("bench_c3", "bench_c3_k16_bf16"), ("bench_bf16_k8", "bench_bf16_k8"),
                               ("bench_GCN", "bench_gcn"), ("bench_GraphSAGE", "bench_graphsage"), ("bench_GIN", "bench_gin")):
        p = os.path.join(a.src, src_name + ".json")
        if os.path.exists(p):
            # the stdout line is compact since round 4; the full record (rooflines of every kernel class, side measurements in
            # full) is the --detail side file of the same run.  Both are kept: <name>.json = full record, <name>_line.json = stdout.
            line = [l for l in open(p).read().splitlines() if l.startswith("{")][-1]
            d = os.path.join(a.src, src_name + ".detail.json")
            full = json.load(open(d)) if os.path.exists(d) else json.loads(line)
            json.dump(full, open(os.path.join(dst, f"{tag}_{dst_name}.json"), "w"), indent=1)
            if os.path.exists(d):
                open(os.path.join(dst, f"{tag}_{dst_name}_line.json"), "w").write(line + "\n")

    all_k, classes = {}, {}
    for mode, label in (("fused", "fused, exact f32"), ("unfused", "unfused"), ("split", "fused, bf16x3"),
                        ("c3", "fused, k=16, bf16 storage (BASELINE configs[2])")):
        st = newest(glob.glob(os.path.join(a.src, f"{mode}_stats", "**", "*kernel_stats.csv"), recursive=True))
        if st:
            shutil.copy(st[0], os.path.join(dst, f"{tag}_{mode}_kernel_stats.csv"))
        fetch = per_launch(os.path.join(a.src, f"{mode}_fetch"), "FETCH_SIZE")
        write = per_launch(os.path.join(a.src, f"{mode}_write"), "WRITE_SIZE")
        cls_acc = defaultdict(lambda: [0.0, 0])
        for k in sorted(set(fetch) & set(write)):
            f_kib, n = full_batch_mean(fetch[k])
            w_kib, _ = full_batch_mean(write[k])
            fb, wb = f_kib * 1024.0, w_kib * 1024.0
            all_k[f"{mode}:{k}"] = {"launches_sampled": n, "fetch_bytes_per_launch_raw": fb,
                                    "fetch_bytes_per_launch_x2_corrected": 2.0 * fb, "write_bytes_per_launch": wb,
                                    "hbm_bytes_per_launch": 2.0 * fb + wb}
            cls = re.sub(r"<.*$", "", k)
            if cls == "features_tiled_kernel":
                cls = "features_kernel"          # the LDS-tiled form of the same kernel class (bench key "features")
            if cls == "gat_layer_bf16_2p_kernel":
                cls = "gat_layer_fused_kernel"   # the two-phase form of the bf16 256 -> 256 instance (bench key "fused")
            cls_acc[cls][0] += (2.0 * fb + wb) * n
            cls_acc[cls][1] += n
        for cls, (tot, n) in cls_acc.items():
            if cls in ("gat_layer_fused_kernel", "gat_aggregate_tiled_kernel", "gemm_f32_kernel", "gemm_wres64_kernel", "features_kernel", "extractor_af_kernel"):
                key = cls if mode not in ("split", "c3") else cls + ":" + mode
                if mode == "unfused" and cls != "gat_aggregate_tiled_kernel":
                    continue
                classes[key] = {"hbm_bytes_per_launch": tot / n, "launches_sampled": n, "path": label}
    bid = os.path.join(a.src, "build_id.txt")
    if os.path.exists(bid):
        classes["_kernel_source_sha"] = open(bid).read().strip()      # bgnn_build_id() of the library that was profiled
    classes["_note"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                        "(tools/collect_profiles.sh: bench.py, 128 tiles of 256x256; full-batch launches only), KiB -> bytes, "
                        "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; launch-weighted mean over the "
                        f"instances of a kernel class; per instance and raw values: profiles/{tag}_pmc_traffic_all.json "
                        "(written by tools/summarise_profiles.py)")
    json.dump(all_k, open(os.path.join(dst, f"{tag}_pmc_traffic_all.json"), "w"), indent=1)
    json.dump(classes, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    for k, v in classes.items():
        if not k.startswith("_"):
            print(f"{k:40s} {v['hbm_bytes_per_launch'] / 1e9:8.2f} GB/launch  ({v['launches_sampled']} launches, {v['path']})")


if __name__ == "__main__":
    main()
