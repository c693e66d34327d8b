#!/bin/bash
# forward launches of the FC stack over the number of workgroups (knobs CLOUDAAE_FC_FWD_BLOCKS_BN / CLOUDAAE_FC_FWD_BLOCKS), back to
# back (weights warm in the memory-side cache) and behind a 1 GB flush (--cold): profiles/r05_fc_fwd_block_sweep.log
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
for cold in "" "--cold"; do
  for bn in 32 64 128 256; do
    echo "== CLOUDAAE_FC_FWD_BLOCKS_BN=$bn $cold"
    timeout 120 python3 tools/bench_fc.py --rows 32 128 --iters 200 --knob CLOUDAAE_FC_FWD_BLOCKS_BN=$bn $cold 2>&1 | grep -E "depth [12]"
  done
  for bl in 128 256 384 512; do
    echo "== CLOUDAAE_FC_FWD_BLOCKS=$bl $cold"
    timeout 120 python3 tools/bench_fc.py --rows 32 128 --iters 200 --knob CLOUDAAE_FC_FWD_BLOCKS=$bl $cold 2>&1 | grep -E "depth 3"
  done
done
