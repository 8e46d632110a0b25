#!/bin/bash
# round 5, dense kernels, call 3: the node-major input gradient with 16 / 12 waves per CU against the 8-wave form
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py -x -q -m gpu -k "node_backward or rgat_layer or column_sums" 2>&1 | tail -3 || exit 1
HET_NODE_DX_WAVES=12 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "node_backward" 2>&1 | tail -2 || exit 1
exp/ab_dense.sh "" cur@HET_NODE_DX_W16=0 cur cur@HET_NODE_DX_WAVES=12 cur@HET_NODE_DX_W16=0 cur cur@HET_NODE_DX_WAVES=12 2>&1 | tee gpurun_out/ab_dense_3.txt | cut -c1-900
