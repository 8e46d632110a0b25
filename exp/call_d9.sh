#!/bin/bash
# round 5, call 9: node_dx with non-temporal stores (now the product); the node-major forward with non-temporal row stores (fwdnt)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py -x -q -m gpu -k "node_forward or node_backward or rgat_layer" 2>&1 | tail -3 || exit 1
exp/ab_dense.sh "" cur@HET_RGAT_NODE_FWD=0 cur fwdnt cur@HET_RGAT_NODE_FWD=0 cur fwdnt 2>&1 | tee gpurun_out/ab_dense_9.txt | cut -c1-700
