#!/bin/bash
# after tools/collect_round.sh (and gpurun's merge): the summaries worth keeping -> profiles/
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
R=r06
for tag in ${R}_trainstep_b32_n1024 ${R}_trainstep_b128_n1024 ${R}_trainstep_b256_n1024_bf16 ${R}_config5_b32_n4096_k20; do
  for f in gpurun_out/$tag/summary/*; do
    [ -f "$f" ] || continue
    case $(basename $f) in roofline_traffic.json) [ $tag = ${R}_trainstep_b32_n1024 ] && cp $f profiles/roofline_traffic.json;; *) cp $f profiles/;; esac
  done
done
for t in ${R}_knn64_wide ${R}_knn64_wide_hinted_k20_n4096 ${R}_hull_pmc_n8593 ${R}_fps_pmc; do
  [ -f gpurun_out/$t/summary.json ] && cp gpurun_out/$t/summary.json profiles/${t}_pmc.json
done
cp gpurun_out/$R/${R}_step_kernel_sequence_*.txt gpurun_out/$R/${R}_bench_*.json gpurun_out/$R/${R}_bench_fps.txt gpurun_out/$R/${R}_smoke.log profiles/ 2>/dev/null
ls profiles | grep "^${R}_" | wc -l
