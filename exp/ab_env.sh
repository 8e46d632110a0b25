#!/bin/bash
# same-box A/B of an env switch that exists only in an experiment build: exp/ab_env.sh VAR [bench args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$1; shift
for rep in 1 2; do
  for on in 0 1; do
    if [ $on = 1 ]; then export $V=1; else unset $V; fi
    python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$V=$on step', d['ms_per_step'], {k:v for k,v in d['per_op_ms'].items()})"
  done
done
