#!/bin/bash
# round 5, call 13: one more wave per SIMD for the backward gather kernels (register bounds; experiment build `occ`)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
exp/ab_dense.sh "" cur occ cur occ cur@HET_SIDE_STREAM=0@HET_RGAT_OVERLAP=0 occ@HET_SIDE_STREAM=0@HET_RGAT_OVERLAP=0 2>&1 | tee gpurun_out/ab_dense_13.txt | cut -c1-330
exp/ab_dense.sh "--model hgt" cur occ cur occ 2>&1 | tee -a gpurun_out/ab_dense_13.txt | cut -c1-330
