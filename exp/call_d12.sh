#!/bin/bash
# round 5, call 12: the projection GEMM's dot epilogue for heads of 16 (DPP sum, three waves per SIMD) against the round-5a form (dotold)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py -x -q -m gpu -k "matmul or attn_dot or rgat_layer" 2>&1 | tail -3 || exit 1
exp/ab_dense.sh "" dotold cur dotold cur dotold@HET_SIDE_STREAM=0@HET_RGAT_OVERLAP=0 cur@HET_SIDE_STREAM=0@HET_RGAT_OVERLAP=0 2>&1 | tee gpurun_out/ab_dense_12.txt | cut -c1-400
