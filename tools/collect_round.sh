#!/bin/bash
# One call on the GPU box: SQ counters of the default C=64 kNN kernel + the three step profiles (B=32 fp32,
# B=128 fp32, B=256 bf16).  Summaries land in gpurun_out/<tag>/summary/; copy them into profiles/.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
timeout 300 bash tools/pmc_kernel.sh r02_knn64_wide knn64_wide -- python3 "$ROOT/tools/bench_knn1.py" 32 1024 64 320 10
timeout 400 bash tools/profile_step.sh r02_trainstep_b32_n1024 "B=32,N=1024"
timeout 400 bash tools/profile_step.sh r02_trainstep_b128_n1024 "B=128,N=1024" --per-gpu-batch 128
timeout 500 bash tools/profile_step.sh r02_trainstep_b256_n1024_bf16 "B=256,N=1024" --per-gpu-batch 256 --gemm-dtype bf16
