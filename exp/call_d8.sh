#!/bin/bash
# round 5, call 8: non-temporal stores / loads in the node-major passes (node_dx, node_rows_sum_w16): experiment builds
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
exp/ab_dense.sh "" cur ntst ntld ntstld cur ntst 2>&1 | tee gpurun_out/ab_dense_8.txt | cut -c1-700
exp/ab_dense.sh "--model rgcn" cur ntst ntld ntstld cur 2>&1 | tee -a gpurun_out/ab_dense_8.txt | cut -c1-700
exp/ab_dense.sh "--model hgt" cur ntst ntld ntstld cur 2>&1 | tee -a gpurun_out/ab_dense_8.txt | cut -c1-700
