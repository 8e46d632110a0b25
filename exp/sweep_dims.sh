#!/bin/bash
# the reference's own sweep shapes (hrt/utils/_do_all_cases.sh: --num_heads 1, in / out dims in {32, 64, 128}^2) through the
# train driver with the reference's flag names; prints forward + backward ms per full-graph epoch
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
# (the reference's sweep also crosses MulFlag = "" / --multiply_among_weights_first_flag for RGAT and HGT: pass the flag as "$@")
for m in ${MODELS:-rgat hgt rgcn}; do
  for dx in 32 64 128; do
    for dy in 32 64 128; do
      python3 -m het_amd.train --model $m -d mag --num_layers 1 --full_graph_training --num_classes $dx --n_infeat $dy --num_heads 1 -e 4 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$m out=$dx in=$dy fwd %.2f bwd %.2f ms' % (d['mean_forward_ms'], d['mean_backward_ms']))
except Exception as e: print('$m out=$dx in=$dy FAILED', e)"
    done
  done
done
