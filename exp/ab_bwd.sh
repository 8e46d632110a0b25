#!/bin/bash
# same-box A/B of grid size / unroll of the folded GAT backward (env switches exist only in the experiment build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "4096 2" "1280 2" "2048 2" "8192 2" "16384 2" "1000000 2" "4096 1" "1280 1" "8192 1" "4096 2"; do
  set -- $cfg
  HET_BWD_GRID=$1 HET_BWD_U=$2 python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-variants 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('grid $1 U $2:', d['roofline']['kernel_ms'], 'ms  step', d['ms_per_step'])"
done
