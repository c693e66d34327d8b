#!/bin/bash
# Profiles the train step of bench.py on the GPU box: kernel trace (50 steps) + the two HBM-traffic PMC passes
# (separate runs, counters only with --kernel-trace), then profiles/collect.py writes the summaries to
# gpurun_out/<tag>/summary/ (copy the ones to keep into profiles/).
#   usage: bash tools/profile_step.sh TAG "B=32,N=1024" [bench.py args...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; WORKLOAD=$2; shift 2
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python3 "$ROOT/bench.py" --steps 50 --warmup 5 --step-only "$@" > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o f -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --step-only "$@" > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o w -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --step-only "$@" > "$OUT/write.log" 2>&1
grep -h '^{"metric"' "$OUT/trace.log" > "$OUT/summary_bench_line.json" || true
python3 "$ROOT/profiles/collect.py" "$OUT" "$TAG" "$WORKLOAD"
# keep the merged-back payload small: the databases stay on the box
rm -rf "$OUT/trace" "$OUT/fetch" "$OUT/write"
