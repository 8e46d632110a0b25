#!/bin/bash
# round 5, call 4: the node-major forward pass (HET_RGAT_NODE_FWD) against the three launches
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "node_forward or node_backward" 2>&1 | tail -5 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_fullsize.py -x -q -m gpu -k "rgat" 2>&1 | tail -5 || exit 1
exp/ab_dense.sh "" cur@HET_RGAT_NODE_FWD=0 cur cur@HET_RGAT_NODE_FWD=0 cur 2>&1 | tee gpurun_out/ab_dense_4.txt | cut -c1-900
