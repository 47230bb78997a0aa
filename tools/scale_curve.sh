#!/usr/bin/env bash
# The 1 -> N GPU curve of the four bench workloads on ONE node, plus the bit-identity check of the row-band sharded survey under
# RCCL: one JSON line per run under $OUT (default gpurun_out/scale/).  Nothing here has run on more than one GPU yet (DESIGN.md 6:
# no 8-GPU node has been available to builders); the script exists so that the first time a node is there the curve and the
# nccl transport of the halo rows are ONE command:
#
#     tools/scale_curve.sh                 # N = 1 2 4 8
#     GPUS="1 2" tools/scale_curve.sh      # a smaller node
#     SHARE_GPU=1 GPUS="1 2" STEPS=2 SURVEY=6000 tools/scale_curve.sh   # REHEARSAL on a one-GPU box (ranks share device 0, gloo):
#                                                                       # checks this script and the sharded code path, measures nothing
#
# bench.py --gpus N starts its N ranks itself (one process per GPU, torch.distributed over RCCL on 127.0.0.1) before anything
# touches a GPU, and exits non-zero when fewer than N GPUs are visible -- this script then stops (set -e), so a line that exists
# was measured.  tiles / c3 / vr are weak scaling (per-rank batches, no collective); survey is strong scaling (one survey, row
# bands, halo tile rows point to point) and carries the sha256 of the stitched result: the N-rank hashes must equal the 1-rank one.
set -euo pipefail
cd "$(dirname "$0")/.."
OUT="${OUT:-gpurun_out/scale}"
GPUS="${GPUS:-1 2 4 8}"
STEPS="${STEPS:-10}"
SURVEY="${SURVEY:-20000}"
EXTRA="${SHARE_GPU:+--share-gpu}"
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in $GPUS; do
  for wl in tiles c3 vr; do
    echo "== $wl on $n GPU(s)" >&2
    python bench.py --gpus "$n" --workload "$wl" --steps "$STEPS" --warmup 2 --no-extras --no-cpu-baseline $EXTRA \
        --detail "$OUT/${wl}_n${n}_detail.json" > "$OUT/${wl}_n${n}.json"
  done
  echo "== survey ${SURVEY}^2 on $n GPU(s), with checksum" >&2
  python bench.py --gpus "$n" --workload survey --survey-size "$SURVEY" --steps 1 --warmup 0 --no-extras --no-cpu-baseline --checksum $EXTRA \
      --detail "$OUT/survey_n${n}_detail.json" > "$OUT/survey_n${n}.json"
done
python - "$OUT" $GPUS <<'PY'
import json, sys
out, gpus = sys.argv[1], [int(g) for g in sys.argv[2:]]
rows, base, sha = [], {}, {}
for wl in ("tiles", "c3", "vr", "survey"):
    for n in gpus:
        r = json.load(open(f"{out}/{wl}_n{n}.json"))
        base.setdefault(wl, r["value"] / r["n_gpus"] if n == gpus[0] else None)
        if wl == "survey":
            sha[n] = r.get("survey", {}).get("stitched_sha256")
        rows.append({"workload": wl, "n_gpus": r["n_gpus"], "value": r["value"], "unit": r["unit"], "scaling": r["scaling"],
                     "ms_per_step": r["ms_per_step"]})
summary = {"runs": rows, "survey_sha256": sha, "survey_bit_identical": len(set(sha.values())) == 1 and None not in sha.values()}
json.dump(summary, open(f"{out}/summary.json", "w"), indent=1)
print(json.dumps(summary))
if not summary["survey_bit_identical"]:
    sys.exit("scale_curve: the stitched survey differs between GPU counts (or a checksum is missing)")
PY
