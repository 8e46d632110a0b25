#!/bin/bash
# Locality experiment of round 4 (run on the GPU box through gpurun, from the repo root): bash profiles/tools/locality.sh
# 1. exp/locality_r04.py sweep: per-kernel HIP-event times of the run-sum forward / backward with the gathered table folded.
# 2. per fold in {1, 16, 256}: separate --pmc passes (never with sys/hip/hsa tracing): L2 hits / misses, texture-address busy,
#    L1 miss-queue stalls, wave cycles.  Output: gpurun_out/locality/{fold.txt, pmc_<fold>_<group>/..., counters.txt}
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/locality
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python3 "$R/exp/locality_r04.py" > "$out/fold.txt" 2>&1 || { tail -20 "$out/fold.txt"; exit 1; }
cat "$out/fold.txt"
export HET_SIDE_STREAM=0
for fold in 1 16 256; do
  i=0
  for grp in "TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc $grp -d "$out/pmc_${fold}_$i" -o run -- python3 "$R/exp/locality_r04.py" one $fold 3 > "$out/pmc_${fold}_$i.log" 2>&1 || { echo "pmc pass fold=$fold group=$i failed"; tail -5 "$out/pmc_${fold}_$i.log"; }
    echo "fold $fold group $i done" >> "$out/progress.txt"
  done
done
find "$out" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.json" -delete
python3 - "$out" > "$out/counters.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
names = ("HET_rgat_aggregate_runs_packed", "HET_rgat_aggregate_hub_items", "HET_rgat_finish_hubs", "HET_rgat_dst_pack",
         "HET_rgat_backward_src_coop", "HET_rgat_backward_src_long", "HET_rgat_grad_er_runs")
for fold in (1, 16, 256):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, f"pmc_{fold}_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].strip()
            if n in names:
                acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== fold {fold} (mean per launch over 3 launches) ==")
    for n in names:
        c = {k: sum(v) / len(v) for k, v in acc[n].items()}
        if not c:
            continue
        hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
        gui = c.get("GRBM_GUI_ACTIVE", 0) / 8  # summed over the 8 XCDs
        line = f"{n:34s} L2 hit {hit/1e6:7.2f} M miss {miss/1e6:7.2f} M ({100*hit/max(1,hit+miss):4.1f} % hits)"
        if gui:
            line += (f" | clocks {gui/1e6:5.2f} M | TA busy {c.get('TA_TA_BUSY_sum',0)/256/gui*100:5.1f} % (avr {c.get('TA_BUSY_avr',0)/gui*100:5.1f} %)"
                     f" | L1 miss-queue stall {c.get('TCP_PENDING_STALL_CYCLES_sum',0)/256/gui*100:5.1f} %"
                     f" | vmem rd insts {c.get('SQ_INSTS_VMEM_RD',0)/1e6:6.2f} M valu {c.get('SQ_INSTS_VALU',0)/1e6:7.2f} M"
                     f" | wave wait {100*c.get('SQ_WAIT_INST_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',1)):4.1f} %")
        print(line)
PY
cat "$out/counters.txt"
