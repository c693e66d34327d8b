#!/bin/bash
# the reproducer alone, then next to a process that runs training steps (one GPU)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r06
{
echo "== alone"
timeout 120 tools/dev/pk_opsel_repro 30000
echo "== next to a process running training steps (bench.py, B = 32)"
( timeout 200 python bench.py --step-only --steps 30000 --warmup 5 > /dev/null 2>&1 ) &
LOAD=$!
sleep 25
timeout 120 tools/dev/pk_opsel_repro 30000
timeout 120 tools/dev/pk_opsel_repro 30000
kill $LOAD 2>/dev/null; wait $LOAD 2>/dev/null
sleep 2
for k in 1 2 3 4; do echo "== ONE process, load kernel $k on a second stream"; timeout 200 tools/dev/pk_opsel_repro 20000 $k; done
} 2>&1 | tee gpurun_out/r06/r06_pk_opsel_repro.log
