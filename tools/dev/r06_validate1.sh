#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
echo "== Adam under the two-process load, deterministic graph: pre-fix objects (dbg1) then the shipped library"
CLOUDAAE_HIP_LIB=$ROOT/cloudaae_amd/libcloudaae_hip_dbg1.so DETERMINISTIC=1 timeout 900 python tools/dev/knn3_debug_stress.py 2000 2 2>&1 | grep -v "amdgpu.ids\|queries wrong"
DETERMINISTIC=1 timeout 900 python tools/dev/knn3_debug_stress.py 2000 2 2>&1 | grep -v "amdgpu.ids\|queries wrong"
echo "== the two-rank tests, single attempts, 6 times over"
for i in 1 2 3 4 5 6; do timeout 900 python -m pytest tests/test_08_dp_gpu.py tests/test_09_sync_bn_gpu.py -q -m gpu 2>&1 | tail -1; done
} > "$OUT/r06_two_rank_tests_strict.log" 2>&1
cut -c1-400 "$OUT/r06_two_rank_tests_strict.log"
timeout 1500 python -m pytest tests -q -m gpu -x > "$OUT/r06_suite_run1.log" 2>&1
tail -5 "$OUT/r06_suite_run1.log"
