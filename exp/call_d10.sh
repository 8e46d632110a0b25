#!/bin/bash
# round 5, call 10: the drow pass over the er rows in the order of their destination nodes (HET_RGAT_DROW_ORDER)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py tests/test_gpu_plan_memory.py -x -q -m gpu -k "rgat or grouping or released" 2>&1 | tail -3 || exit 1
exp/ab_dense.sh "" cur@HET_RGAT_DROW_ORDER=0 cur cur@HET_RGAT_DROW_ORDER=0 cur cur@HET_RGAT_DROW_ORDER=0@HET_SIDE_STREAM=0@HET_RGAT_OVERLAP=0 cur@HET_SIDE_STREAM=0@HET_RGAT_OVERLAP=0 2>&1 | tee gpurun_out/ab_dense_10.txt | cut -c1-400
