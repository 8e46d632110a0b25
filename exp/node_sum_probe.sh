#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/nsp
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python3 "$R/exp/node_sum_probe.py" > "$out/times.txt" 2>&1 || { tail -20 "$out/times.txt"; exit 1; }
HET_NODE_SUM_LDS_TILE=1 python3 "$R/exp/node_sum_probe.py" > "$out/times_tile.txt" 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc $grp -d "$out/pmc_$i" -o run -- python3 "$R/exp/node_sum_probe.py" one > "$out/pmc_$i.log" 2>&1 || { echo "pmc group $i failed"; tail -3 "$out/pmc_$i.log"; }
done
find "$out" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
python3 - "$out" > "$out/counters.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "node_rows_sum" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for kname, v in sorted(acc.items()):
    print(f"{kname:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
cat "$out/times.txt" "$out/times_tile.txt" "$out/counters.txt"
