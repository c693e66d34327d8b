#!/bin/bash
# A/B of two builds of the library on the bench lines:  bash tools/dev/ab_lib.sh <libA.so> <libB.so> [tag]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
A=$1; B=$2; TAG=${3:-ab}
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
LOG=$OUT/r06_${TAG}.log
: > "$LOG"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%.1f clouds/s  %.4f ms" % (d["value"], d["ms_per_step"]), {k: d[k] for k in ("step_ms_min","step_ms_median","step_ms_max") if k in d})
except Exception as e:
    print("no line:", e)
PY
}
for rep in 1 2; do
  for which in A B; do
    lib=$A; [ $which = B ] && lib=$B
    for cfg in "" "--per-gpu-batch 128" "--per-gpu-batch 256 --gemm-dtype bf16" "--config5 --steps 20 --warmup 5"; do
      CLOUDAAE_HIP_LIB=$ROOT/$lib python bench.py --step-only $cfg > /tmp/ab_line.json 2>/dev/null
      echo "rep $rep lib $which [$cfg]: $(line /tmp/ab_line.json)" | tee -a "$LOG"
    done
  done
done
for which in A B; do
  lib=$A; [ $which = B ] && lib=$B
  CLOUDAAE_HIP_LIB=$ROOT/$lib python bench.py > $OUT/r06_${TAG}_full_$which.json 2>/dev/null
  python - $OUT/r06_${TAG}_full_$which.json <<'PY' | tee -a "$LOG"
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("full", sys.argv[1][-6], d["value"], [ (c["shape"][:28], c["us_per_launch"]) for c in d["chamfer_kernel"]], [(f["shape"], f["us_per_launch"]) for f in d["fps_kernel"]], d["roofline"]["launch_ms"])
PY
done
