#!/bin/bash
# the hinted kNN against the plain one: kernel times at the headline and config-5 shapes, then the config-5 / B = 32 steps
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r06
{
for shape in "32 1024 64 320 10" "128 1024 64 320 10" "32 4096 64 320 20 3"; do
  for h in "" exact noisy; do KNN_HINT=$h python tools/bench_knn1.py $shape; done
done
for hint in 0 1; do
  CLOUDAAE_KNN_HINT=$hint python bench.py --config5 --step-only --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5 hint=$hint', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
  CLOUDAAE_KNN_HINT=$hint python bench.py --step-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=32 hint=$hint', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/r06_knn_hint_ab.log
