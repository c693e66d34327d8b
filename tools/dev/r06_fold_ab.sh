#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_10_replay_gpu.py -q -m gpu -x -k "finalise or determin" 2>&1 | tail -2
for rep in 1 2 3; do
for fold in 1 0; do
for cfg in "" "--per-gpu-batch 128"; do
  CLOUDAAE_EC_FOLD=$fold python bench.py --step-only $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep fold=$fold [$cfg]', d['value'], d['ms_per_step'], d['step_ms_min'], d['step_ms_median'], d['step_ms_max'])"
done; done; done | tee "$OUT/r06_fold_ab.log"
bash tools/kernel_sequence.sh > "$OUT/r06_step_kernel_sequence_b32.txt" 2>&1
grep "ec_stats\|ec_bwd_stats\|finalize" "$OUT/r06_step_kernel_sequence_b32.txt"
