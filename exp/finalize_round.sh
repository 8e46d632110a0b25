#!/bin/bash
# End-of-round evidence in ONE gpurun call (one box acquisition): A/B of the last switch, the whole GPU test suite, the four profile
# configurations (profiles/tools/collect_all.sh), the committed profiles replaced ON THE BOX so that the bench lines that follow
# quote counters of the same kernel sources, then the three complete bench lines.  Afterwards (here): copy gpurun_out/<cfg>_r05/* and
# gpurun_out/bench_*.json into profiles/r05/.   gpurun --timeout 1190 -- 'bash exp/finalize_round.sh'
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p profiles/r05
python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1; echo tests rc=$?; tail -2 gpurun_out/final_tests.log
bash profiles/tools/collect_all.sh > gpurun_out/collect_all.log 2>&1; tail -4 gpurun_out/collect_all.log
for c in default default_serial rgcn hgt; do for f in kernel_stats.csv pmc.json summary.txt; do cp gpurun_out/${c}_r05/$f profiles/r05/${c}_$f; done; done
python bench.py > gpurun_out/bench_default_full.json 2> gpurun_out/bench_default_full.err; echo bench rc=$?
python bench.py --model rgcn > gpurun_out/bench_rgcn.json 2>/dev/null; python bench.py --model hgt > gpurun_out/bench_hgt.json 2>/dev/null; echo fin
