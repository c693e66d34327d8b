#!/bin/bash
# one GPU call: the self-checking scan kernel under the two-process load (profiles/notes_two_processes_one_gpu.md, round 6)
# variants of the staging code (cloudaae_amd/csrc/knn.hip, CLOUDAAE_KNN3_VARIANT): 0 as shipped, 1 the norm without the packed
# chain, 2 three one-word loads instead of global_load_dwordx3, 3 / 4 / 5 the shipped sequence in inline assembly (as is / two
# wait states between the packed instructions / wait states between the v_mov and the packed multiply)
#   bash tools/dev/knn3_debug_run.sh "3 4 5" 1500
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
VARIANTS=${1:-"0 1 2"}
STEPS=${2:-1200}
LOG=$OUT/r06_knn3_debug_v$(echo $VARIANTS | tr -d ' ').log
{
for v in $VARIANTS; do
  echo "== variant $v, two processes, $STEPS steps each, [16, 256], scan kernel forced"
  CLOUDAAE_HIP_LIB=$ROOT/cloudaae_amd/libcloudaae_hip_dbg$v.so CLOUDAAE_KNN3_WIDE=0 timeout 600 python tools/dev/knn3_debug_stress.py $STEPS 2
  echo "== variant $v, two processes, $STEPS steps each, [32, 128]"
  CLOUDAAE_HIP_LIB=$ROOT/cloudaae_amd/libcloudaae_hip_dbg$v.so timeout 600 python tools/dev/knn3_debug_stress.py $STEPS 2 32 128
done
} > "$LOG" 2>&1
grep -v "queries wrong\|amdgpu.ids" "$LOG" | cut -c1-700 | grep "^==\|steps $STEPS\|kind 6" | awk '/kind 6/{k++; if(k>14) next} {print}'
