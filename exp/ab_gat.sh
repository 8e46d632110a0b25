#!/bin/bash
# Same-box A/B of the grouped GAT kernels' unroll (env switches exist only in experiment builds).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() {  # tag, variant, env...
  tag=$1; variant=$2; shift 2
  env "$@" true
  ( export "$@"; rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/ab_$tag -o run -- python3 $R/bench.py --steps 6 --warmup 2 --variant $variant --no-cpu-baseline --no-variants > $R/gpurun_out/ab_$tag.log 2>&1 )
  python3 - <<PY
import csv
for r in csv.DictReader(open('$R/gpurun_out/ab_$tag/run_kernel_stats.csv')):
    if 'HET_gat' in r['Name'] and float(r['AverageNs']) > 2e5:
        print('$tag', r['Name'].replace('(anonymous namespace)::','')[5:60], round(float(r['AverageNs'])/1e6,3))
PY
}
run d_base default X=1
run d_b4 default HET_U_BWD=4
run d_b1 default HET_U_BWD=1
run d_a8 default HET_U_AGG=8
run d_a2 default HET_U_AGG=2
run d_base2 default X=1
run c_base compact X=1
run c_s4 compact HET_U_SRC=4
run c_s1 compact HET_U_SRC=1
run c_base2 compact X=1
