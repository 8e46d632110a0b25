#!/bin/bash
# same-box A/B of library builds on the RGAT step (round 5):   exp/ab_r05.sh "<bench args>" name1 name2 ...
#   ("cur" = the product library het_amd/libhet_amd.so, other names = exp/libs/lib_<name>.so; every name runs in the given order)
# prints step time, the two gather ops (entry-to-return) and their launches' own durations
R=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; shift
#   a name may carry environment switches: cur@HET_SIDE_STREAM=0@HET_RGAT_OVERLAP=0
for spec in "$@"; do
  n=${spec%%@*}
  envs=""
  if [ "$spec" != "$n" ]; then envs=$(echo "${spec#*@}" | tr '@' ' '); fi
  if [ "$n" = cur ]; then unset HET_AMD_LIB; else export HET_AMD_LIB=$R/exp/libs/lib_$n.so; fi
  n=$(echo "$spec" | tr '@=' '__')
  env $envs python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-models --no-dist-rehearsal $A 2>$R/gpurun_out/ab_r05_$n.err | tail -1 > $R/gpurun_out/ab_r05_$n.json
  python3 - "$n" "$R/gpurun_out/ab_r05_$n.json" <<'PY'
import sys, json
n, path = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(path).read())
except Exception as e:
    print(n, "FAILED", e); sys.exit(0)
p, k = d.get("per_op_ms") or {}, d.get("kernel_ms") or {}
ks = {a: round(v["ms_per_step"], 3) for a, v in k.items() if a.startswith(("HET_rgat_aggregate_", "HET_rgat_backward_"))}
print(f"{n:40s} step {d['ms_per_step']:.3f}  fwd_op {p.get('rgat_aggregate_compact_runs')}  bwd_op {p.get('rgat_backward_compact_runs')}  {ks}")
PY
done
