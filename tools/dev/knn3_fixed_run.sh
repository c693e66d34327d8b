#!/bin/bash
# one GPU call: the shipped library under the two-process load, every layer-1 kernel (round 6, after the packed-fp32 fix);
# last, the pre-fix Adam (libcloudaae_hip_dbg1.so: old objects, kNN norms repaired) for contrast
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
STEPS=${1:-2000}
LOG=$OUT/r06_knn3_fixed.log
{
echo "== shipped library, [16, 256], default dispatch (knn3_wide)";            timeout 900 python tools/dev/knn3_debug_stress.py $STEPS 2
echo "== shipped library, [16, 256], CLOUDAAE_KNN3_WIDE=0 (knn3_scan)";         CLOUDAAE_KNN3_WIDE=0 timeout 900 python tools/dev/knn3_debug_stress.py $STEPS 2
echo "== shipped library, [16, 256], CLOUDAAE_KNN_SCAN=0 (first generation)";   CLOUDAAE_KNN_SCAN=0 timeout 900 python tools/dev/knn3_debug_stress.py $STEPS 2
echo "== shipped library, [32, 128], default dispatch (knn3_scan)";             timeout 900 python tools/dev/knn3_debug_stress.py $STEPS 2 32 128
echo "== shipped library, [32, 128], CLOUDAAE_KNN_SCAN=0 (first generation)";   CLOUDAAE_KNN_SCAN=0 timeout 900 python tools/dev/knn3_debug_stress.py $STEPS 2 32 128
echo "== shipped library, [8, 1024], default dispatch";                         timeout 900 python tools/dev/knn3_debug_stress.py 1000 2 8 1024
echo "== pre-fix step.o / synth.o (dbg1: kNN norms repaired, Adam with its op_sel packed instructions), [16, 256]"
CLOUDAAE_HIP_LIB=$ROOT/cloudaae_amd/libcloudaae_hip_dbg1.so CLOUDAAE_KNN3_WIDE=0 timeout 900 python tools/dev/knn3_debug_stress.py $STEPS 2
} > "$LOG" 2>&1
grep -v "queries wrong\|amdgpu.ids" "$LOG" | cut -c1-400
