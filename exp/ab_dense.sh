#!/bin/bash
# same-box A/B on any model's step with EVERY timed launch printed (round 5, the dense kernels):
#   exp/ab_dense.sh "<bench args, e.g. --model hgt>" name1 name2@ENV=v ...     (names as in exp/ab_r05.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; shift
for spec in "$@"; do
  n=${spec%%@*}
  envs=""
  if [ "$spec" != "$n" ]; then envs=$(echo "${spec#*@}" | tr '@' ' '); fi
  if [ "$n" = cur ]; then unset HET_AMD_LIB; else export HET_AMD_LIB=$R/exp/libs/lib_$n.so; fi
  tag=$(echo "$spec$A" | tr '@= -' '____')
  env $envs python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-models --no-dist-rehearsal $A 2>$R/gpurun_out/ab_dense_$tag.err | tail -1 > $R/gpurun_out/ab_dense_$tag.json
  python3 - "$spec $A" "$R/gpurun_out/ab_dense_$tag.json" <<'PY'
import sys, json
n, path = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(path).read())
except Exception as e:
    print(n, "FAILED", e); sys.exit(0)
k = d.get("kernel_ms") or {}
ks = {a.replace("HET_", ""): round(v["ms_per_step"], 3) for a, v in k.items() if v["ms_per_step"] >= 0.02}
print(f"{n:44s} step {d['ms_per_step']:.3f}  {ks}")
PY
done
