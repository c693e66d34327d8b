#!/bin/bash
# One call on the GPU box: the round's profiles.  Summaries land in gpurun_out/<tag>/summary/ and gpurun_out/r06/;
# copy them into profiles/ (tools/collect_copy.sh).  RUN IT LAST: bench.py takes `roofline.traffic` only from a
# profiles/roofline_traffic.json whose source hashes match the sources that are running (tests/test_profiles_fresh.py
# fails when they do not).
#   bash tools/collect_round.sh            (about 10 minutes)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
R=r06
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
# the step profiles: kernel trace + the two HBM-traffic PMC passes each (tools/profile_step.sh)
timeout 400 bash tools/profile_step.sh ${R}_trainstep_b32_n1024 "B=32,N=1024"
timeout 400 bash tools/profile_step.sh ${R}_trainstep_b128_n1024 "B=128,N=1024" --per-gpu-batch 128
timeout 500 bash tools/profile_step.sh ${R}_trainstep_b256_n1024_bf16 "B=256,N=1024" --per-gpu-batch 256 --gemm-dtype bf16
timeout 500 bash tools/profile_step.sh ${R}_config5_b32_n4096_k20 "B=32,N=4096" --config5
# (the bench lines below take `roofline.traffic` from profiles/roofline_traffic.json: put THIS run's file there first, so that a
#  collection made right after a priced source changed carries the number in its own lines)
[ -f "$ROOT/gpurun_out/${R}_trainstep_b32_n1024/summary/roofline_traffic.json" ] && cp "$ROOT/gpurun_out/${R}_trainstep_b32_n1024/summary/roofline_traffic.json" "$ROOT/profiles/roofline_traffic.json"
# kernel sequences of one replayed step
bash tools/kernel_sequence.sh > "$OUT/${R}_step_kernel_sequence_b32.txt" 2>&1
bash tools/kernel_sequence.sh --config5 > "$OUT/${R}_step_kernel_sequence_config5.txt" 2>&1
# SQ counters of the C = 64 kNN kernel at the headline shape and (hinted) at config 5's, of the hull-vertex kernel, of FPS
timeout 300 bash tools/pmc_kernel.sh ${R}_knn64_wide knn64_wide -- python3 "$ROOT/tools/bench_knn1.py" 32 1024 64 320 10
KNN_HINT=noisy timeout 300 bash tools/pmc_kernel.sh ${R}_knn64_wide_hinted_k20_n4096 knn64 -- python3 "$ROOT/tools/bench_knn1.py" 32 4096 64 320 20 3
timeout 300 bash tools/pmc_kernel.sh ${R}_hull_pmc_n8593 hull_vertex -- python3 "$ROOT/tools/dev/run_hpr.py" 32 8192 1
timeout 300 bash tools/pmc_kernel.sh ${R}_fps_pmc fps_kernel -- python3 "$ROOT/tools/bench_fps.py"
# the bench lines (no profiler): the driver-shaped full lines of the named configurations, then the A/B lines
python3 bench.py > "$OUT/${R}_bench_b32_n1024.json" 2> "$OUT/bench_b32.err"
python3 bench.py --per-gpu-batch 128 > "$OUT/${R}_bench_b128_n1024.json" 2>/dev/null
python3 bench.py --per-gpu-batch 256 --gemm-dtype bf16 > "$OUT/${R}_bench_b256_n1024_bf16.json" 2>/dev/null
python3 bench.py --config5 --steps 20 --warmup 5 > "$OUT/${R}_bench_config5_b32_n4096_k20.json" 2>/dev/null
python3 bench.py --gemm-dtype f32 --step-only > "$OUT/${R}_bench_b32_n1024_f32_mfma.json" 2>/dev/null
python3 tools/bench_fps.py > "$OUT/${R}_bench_fps.txt" 2>/dev/null
# the RCCL path with ONE rank (no second GPU on this box): the same replayed step with the gradient exchange's two
# all-reduces (and, second line, SyncBN's 22) going through RCCL -- the `comm` block of the N > 1 bench line
export HSA_ENABLE_IPC_MODE_LEGACY=0
CLOUDAAE_FORCE_COLLECTIVES=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port 29611 bench.py --gpus 1 --per-gpu-batch 128 --step-only > "$OUT/${R}_bench_b128_one_rank_rccl.json" 2> "$OUT/rccl.err"
CLOUDAAE_FORCE_COLLECTIVES=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port 29612 bench.py --gpus 1 --per-gpu-batch 128 --step-only --sync-bn > "$OUT/${R}_bench_b128_one_rank_rccl_syncbn.json" 2>> "$OUT/rccl.err"
# the driver's own entry points, as it runs them
python3 -c "import __graft_entry__ as g; g.smoke()" > "$OUT/${R}_smoke.log" 2>&1
ls -la "$OUT"
