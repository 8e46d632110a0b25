#!/bin/bash
# same-box A/B of a numeric env switch of an experiment build: exp/ab_vals.sh VAR "bench args" v1 v2 ... (0 = unset)
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$1; A=$2; shift; shift
for val in "$@"; do
  if [ "$val" = 0 ]; then unset $V; else export $V=$val; fi
  python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants $A 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['per_op_ms']; print('$V=$val step', d['ms_per_step'], 'gat_fwd', p.get('relational_fused_gat_separate_coo'), 'gat_bwd', p.get('backward_relational_fused_gat_separate_coo'), 'mm_bwd', p.get('backward_rgnn_relational_matmul'))"
done
