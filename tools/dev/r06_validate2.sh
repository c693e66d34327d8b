#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_10_replay_gpu.py tests/test_01_layers_gpu.py tests/test_03_configs_gpu.py tests/test_05_variants_gpu.py tests/test_09_sync_bn_gpu.py -q -m gpu -x > "$OUT/r06_fold_tests.log" 2>&1
tail -4 "$OUT/r06_fold_tests.log"
for cfg in "" "--per-gpu-batch 128" "--per-gpu-batch 256 --gemm-dtype bf16" "--config5 --steps 20 --warmup 5"; do
  python bench.py --step-only $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg]', d['value'], d['ms_per_step'], d['step_ms_min'], d['step_ms_median'], d['step_ms_max'])"
done | tee "$OUT/r06_fold_bench.log"
bash tools/kernel_sequence.sh > "$OUT/r06_step_kernel_sequence_b32.txt" 2>&1
wc -l "$OUT/r06_step_kernel_sequence_b32.txt"; grep -c finalize "$OUT/r06_step_kernel_sequence_b32.txt"
bash tools/dev/knn3_debug_run.sh "0 1 2" 1200 | grep "^==\|steps 1200" | cut -c1-200
