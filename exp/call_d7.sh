#!/bin/bash
# round 5, call 7: what the node-major forward costs under the node orders
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
exp/ab_dense.sh "" cur@HET_RGAT_NODE_FWD=0 cur cur@HET_NODE_SUM_MIX=0 cur@HET_RGAT_NODE_ORDER=0 cur@HET_RGAT_NODE_FWD=0@HET_SIDE_STREAM=0@HET_RGAT_OVERLAP=0 cur@HET_SIDE_STREAM=0@HET_RGAT_OVERLAP=0 cur@HET_SIDE_STREAM=0@HET_RGAT_OVERLAP=0@HET_NODE_SUM_MIX=0 2>&1 | tee gpurun_out/ab_dense_7.txt | cut -c1-900
