#!/bin/bash
# The GPU test suite file by file, each under its own time limit (a hung kernel costs its file's limit, not the whole call's),
# slowest tests listed.   bash tools/gpu_suite.sh [log]     -- prints one summary line per file and a total
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
LOG=${1:-$ROOT/gpurun_out/gpu_suite.log}
mkdir -p "$(dirname "$LOG")"
: > "$LOG"
export HSA_ENABLE_IPC_MODE_LEGACY=0
fail=0
for f in tests/test_*_gpu.py; do
  lim=420; case $f in *test_03*) lim=600;; esac
  echo "== $f" >> "$LOG"
  timeout $lim python -m pytest "$f" -q -m gpu --durations=3 >> "$LOG" 2>&1
  rc=$?
  [ $rc -ne 0 ] && fail=1
  echo "$f rc=$rc $(grep -E '^[0-9]+ (passed|failed)|passed|failed' "$LOG" | tail -1)"
done
echo "suite: $([ $fail -eq 0 ] && echo GREEN || echo FAILED)"
exit $fail
