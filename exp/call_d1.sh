#!/bin/bash
# round 5, dense kernels, call 1: bias gradient from the self-loop dW launch; GEMM workgroup target; what the dW flush costs
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py -x -q -m gpu -k "column_sums or rgat_layer or node_backward" 2>&1 | tail -3 || exit 1
exp/ab_dense.sh "" cur@HET_RGAT_BIAS_IN_DW=0 cur cur@HET_GEMM_WGS=4096 cur@HET_GEMM_WGS=8192 dw_noatomic dw_noepi cur@HET_RGAT_BIAS_IN_DW=0 cur 2>&1 | tee gpurun_out/ab_dense_1.txt | cut -c1-900
exp/ab_dense.sh "--model hgt" cur cur@HET_GEMM_WGS=4096 cur@HET_GEMM_WGS=8192 dw_noepi cur 2>&1 | tee -a gpurun_out/ab_dense_1.txt | cut -c1-900
