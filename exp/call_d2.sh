#!/bin/bash
# round 5, dense kernels, call 2: the dW flush on four waves (dwold = the flush on wave 0, colsum ignored); row-dot dW workgroup count
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py -x -q -m gpu -k "matmul or column_sums or layer" 2>&1 | tail -3 || exit 1
exp/ab_dense.sh "" dwold cur cur@HET_ROWDOT_DW_WGS=1024 cur@HET_ROWDOT_DW_WGS=2048 cur@HET_ROWDOT_DW_WGS=4096 dwold cur 2>&1 | tee gpurun_out/ab_dense_2.txt | cut -c1-900
exp/ab_dense.sh "--model hgt" dwold cur dwold cur 2>&1 | tee -a gpurun_out/ab_dense_2.txt | cut -c1-900
exp/ab_dense.sh "--model rgcn" dwold cur dwold cur 2>&1 | tee -a gpurun_out/ab_dense_2.txt | cut -c1-900
