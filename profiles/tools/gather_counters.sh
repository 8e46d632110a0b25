#!/bin/bash
# Per-launch counters of the RGAT gather kernels, every launch alone on the chip (run on the GPU box through gpurun, from the repo root):
#   bash profiles/tools/gather_counters.sh <tag> [r04lib]
# <tag> names the output (gpurun_out/counters_<tag>.txt); a second argument makes the run use exp/libs/lib_<r04lib>.so (the round-4
# library, for the before / after table of DESIGN.md section 4.3).  Separate --pmc passes, never with sys / hip / hsa tracing.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1
out=$R/gpurun_out/counters_$tag
mkdir -p "$out"
export HET_SIDE_STREAM=0 HET_RGAT_OVERLAP=0
if [ -n "$2" ]; then export HET_AMD_LIB=$R/exp/libs/lib_$2.so; fi
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAVES" "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --output-format csv --kernel-trace --pmc $grp -d "$out/pmc_$i" -o run -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-models --no-dist-rehearsal > "$out/pmc_$i.log" 2>&1 || { echo "pmc pass group=$i failed"; tail -5 "$out/pmc_$i.log"; }
  echo "group $i done" >> "$out/progress.txt"
done
find "$out" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.json" -delete
python3 - "$out" "$tag" > "$R/gpurun_out/counters_$tag.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d, tag = sys.argv[1], sys.argv[2]
names = ("HET_rgat_aggregate_runs_packed", "HET_rgat_aggregate_hub_items", "HET_rgat_finish_hubs", "HET_rgat_dst_pack", "HET_rgat_drow_rec",
         "HET_rgat_drow_pass", "HET_rgat_colsum_rows", "HET_rgat_backward_src_coop", "HET_rgat_backward_src_long", "HET_rgat_grad_er_runs")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].strip()
        if n in names:
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"== {tag}: mean per launch, every launch alone on the chip (HET_SIDE_STREAM=0 HET_RGAT_OVERLAP=0), bench.py default workload ==")
for n in names:
    c = {k: sum(v) / len(v) for k, v in acc[n].items()}
    if not c:
        continue
    hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    gui = c.get("GRBM_GUI_ACTIVE", 0) / 8  # summed over the 8 XCDs
    hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024 / 1e9
    print(f"{n:32s} clocks {gui/1e6:5.2f} M | VALU {c.get('SQ_INSTS_VALU',0)/1e6:7.2f} M  SALU {c.get('SQ_INSTS_SALU',0)/1e6:6.2f} M  LDS {c.get('SQ_INSTS_LDS',0)/1e6:5.2f} M  "
          f"vmem rd {c.get('SQ_INSTS_VMEM_RD',0)/1e6:5.2f} M wr {c.get('SQ_INSTS_VMEM_WR',0)/1e6:5.2f} M | waves {c.get('SQ_WAVES',0)/1e6:5.2f} M  wave wait "
          f"{100*c.get('SQ_WAIT_INST_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',1)):4.1f} % | TA busy {c.get('TA_TA_BUSY_sum',0)/256/max(1,gui)*100:5.1f} %  "
          f"L1 miss-queue stall {c.get('TCP_PENDING_STALL_CYCLES_sum',0)/256/max(1,gui)*100:5.1f} % | L2 hit {hit/1e6:6.2f} M miss {miss/1e6:6.2f} M "
          f"({100*hit/max(1,hit+miss):4.1f} %) | traffic {hbm:5.2f} GB (2 FETCH + WRITE)")
PY
cat "$R/gpurun_out/counters_$tag.txt"
