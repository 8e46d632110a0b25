#!/bin/bash
# End-of-round evidence, two gpurun calls (each under the 1200-second limit):
#   gpurun --timeout 1190 -- 'bash exp/finalize_round.sh tests'      the whole GPU test suite + the complete default bench line
#   gpurun --timeout 1190 -- 'bash exp/finalize_round.sh profiles'   the four profile configurations (profiles/tools/collect_all.sh),
#        one-step timelines, the committed profiles replaced ON THE BOX so that the bench lines that follow quote counters of the
#        same kernel sources, then the RGCN / HGT bench lines and a second default line
# Afterwards (here): copy gpurun_out/<cfg>_r05/{kernel_stats.csv,pmc.json,summary.txt}, gpurun_out/*_timeline.txt and
# gpurun_out/bench_*.json into profiles/r05/.
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p profiles/r05
if [ "$1" = "tests" ]; then
  python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1; echo tests rc=$?; tail -2 gpurun_out/final_tests.log
  python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo smoke rc=$?; tail -2 gpurun_out/final_smoke.log
  python bench.py > gpurun_out/bench_default_full_a.json 2> gpurun_out/bench_default_full_a.err; echo bench rc=$?
  tail -c 600 gpurun_out/bench_default_full_a.json
else
  python -m pytest tests/test_gpu_layers.py -x -q -m gpu -k "rgcn or hgt" > gpurun_out/final_tests_b.log 2>&1; echo layer tests rc=$?; tail -1 gpurun_out/final_tests_b.log
  bash profiles/tools/collect_all.sh > gpurun_out/collect_all.log 2>&1; tail -4 gpurun_out/collect_all.log
  for c in default default_serial rgcn hgt; do for f in kernel_stats.csv pmc.json summary.txt; do cp gpurun_out/${c}_r05/$f profiles/r05/${c}_$f; done; done
  python3 profiles/tools/timeline.py gpurun_out/default_r05/stats HET_rgat_aggregate_runs_packed 0.62 > gpurun_out/default_timeline.txt
  python3 profiles/tools/timeline.py gpurun_out/default_serial_r05/stats HET_rgat_aggregate_runs_packed 0.75 > gpurun_out/default_serial_timeline.txt
  python3 profiles/tools/timeline.py gpurun_out/hgt_r05/stats HET_hgt_aggregate_rows 0.95 > gpurun_out/hgt_timeline.txt
  python3 profiles/tools/timeline.py gpurun_out/rgcn_r05/stats HET_seg_dw_mfma 1.78 > gpurun_out/rgcn_timeline.txt
  python bench.py > gpurun_out/bench_default_full.json 2> gpurun_out/bench_default_full.err; echo bench rc=$?
  python bench.py --model rgcn > gpurun_out/bench_rgcn.json 2>/dev/null; python bench.py --model hgt > gpurun_out/bench_hgt.json 2>/dev/null; echo fin
fi
