#!/bin/bash
# same-box A/B of experiment builds of the library (exp/libs/lib_<name>.so, selected with HET_AMD_LIB):
#   exp/ab_libs.sh "bench args" name1 name2 ...     ("base" = the product library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; shift
for n in "$@"; do
  if [ "$n" = base ]; then unset HET_AMD_LIB; else export HET_AMD_LIB=$R/exp/libs/lib_$n.so; fi
  python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants $A 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['per_op_ms']; print('$n step', d['ms_per_step'], 'gat_fwd', p.get('relational_fused_gat_separate_coo'), 'gat_bwd', p.get('backward_relational_fused_gat_separate_coo'), 'mm_bwd', p.get('backward_rgnn_relational_matmul'))"
done
