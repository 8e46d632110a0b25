#!/bin/bash
# Per-launch durations of the node-major passes (HET_node_rows_sum / HET_node_dx) of one model, serial (no side streams) so that
# every launch is alone on the chip:  bash exp/node_pass_trace.sh <model> <tag>
set -o pipefail
model=$1; tag=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export HET_SIDE_STREAM=0 HET_RGAT_OVERLAP=0
rocprofv3 --output-format csv --kernel-trace -d "$out" -o run -- python3 "$R/bench.py" --model $model --steps 4 --warmup 2 --no-cpu-baseline --no-variants --no-models > "$out/bench.log" 2>&1
find "$out" -type f ! -name "*.csv" ! -name "*.log" -delete
python3 - "$out" <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "node" in n or "segment_sum" in n or "seg_dw" in n or "colsum" in n or "seg_gemm" in n or "hgt" in n:
        d[n.split("(")[0][:70] + " grid=" + r.get("Grid_Size", "?")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(d.items()):
    v = v[len(v) // 2:]  # the later half: past the warm-up
    print(f"{k:100s} n={len(v):3d} avg_ms={sum(v)/len(v):.4f} min={min(v):.4f}")
P
