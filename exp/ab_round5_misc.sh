#!/bin/bash
# Round 5, one box: (a) edge order of the synthetic lists, interleaved A/B x3; (b) hub threshold / backward pack
# threshold re-swept on the round-5 kernels; (c) HGT with the folding kernel on / off; (d) RGCN backward fork placements.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
B="--no-cpu-baseline --no-variants --no-models --no-dist-rehearsal"
one() {  # label, env..., -- bench args
  label=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python3 bench.py --steps 20 --warmup 5 $B "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$label', 'ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median_events'))"
}
echo "== (a) edge order, interleaved"
for i in 1 2 3; do one "src_dst#$i" -- --edge-order src_dst; one "random#$i" -- --edge-order random; one "src#$i" -- --edge-order src; done
echo "== (b) hub threshold (forward) and pack threshold (backward) on the round-5 kernels"
for v in 64 96 128 192 256; do one "HUB_MIN=$v" HET_RGAT_HUB_MIN=$v --; done
for v in 32 64 96 128; do one "BWD_PACK_T=$v" HET_RGAT_BWD_PACK_T=$v --; done
echo "== (c) HGT folding kernel"
for i in 1 2; do one "hgt fold kernel#$i" -- --model hgt; one "hgt torch fold#$i" HET_HGT_FOLD_KERNEL=0 -- --model hgt; done
echo "== (d) RGCN"
for i in 1 2; do one "rgcn#$i" -- --model rgcn; one "rgcn fork0#$i" HET_RGCN_BWD_FORK=0 -- --model rgcn; one "rgcn fork2#$i" HET_RGCN_BWD_FORK=2 -- --model rgcn; done
