#!/bin/bash
# round 5, GPU call B: the round-4 library against the current one on one box (two streams and alone on the chip), then the per-launch
# counters of the gather kernels for both (the current one also at the round-4 hub threshold, so that the launches cover the same edges)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py tests/test_gpu_plan_memory.py -x -q -m gpu -k "rgat or 64_bit or released" 2>&1 | tail -2 || exit 1
E="@HET_SIDE_STREAM=0@HET_RGAT_OVERLAP=0"
exp/ab_r05.sh "" r04 cur r04 cur r04$E cur$E cur@HET_RGAT_HUB_MIN=128 cur${E}@HET_RGAT_HUB_MIN=128 2>&1 | tee gpurun_out/ab_r05_final.txt | cut -c1-420
bash profiles/tools/gather_counters.sh r04lib r04 > gpurun_out/gc_r04.log 2>&1; tail -12 gpurun_out/counters_r04lib.txt | cut -c1-330
export HET_RGAT_HUB_MIN=128
bash profiles/tools/gather_counters.sh r05_hub128 > gpurun_out/gc_r05h.log 2>&1; tail -12 gpurun_out/counters_r05_hub128.txt | cut -c1-330
unset HET_RGAT_HUB_MIN
bash profiles/tools/gather_counters.sh r05 > gpurun_out/gc_r05.log 2>&1; tail -12 gpurun_out/counters_r05.txt | cut -c1-330
