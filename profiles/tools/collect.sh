#!/bin/bash
# Evidence collection for one bench configuration (run on the GPU box through gpurun, from the repo root):
#   bash profiles/tools/collect.sh <tag> [bench.py flags...]
# Pass 1: kernel trace + stats (durations).  Passes 2-4: PMC counters, one family per pass and never together
# with sys/hip/hsa tracing (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "PMC slots").
# The program follows "--" directly (no env/bash hop).  Output: gpurun_out/<tag>/{stats,fetch,write,mfma}/...
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/stats" -o run -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-models --no-dist-rehearsal "$@" > "$out/bench_stats.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$out/fetch" -o run -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-models --no-dist-rehearsal "$@" > "$out/bench_fetch.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$out/write" -o run -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-models --no-dist-rehearsal "$@" > "$out/bench_write.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d "$out/mfma" -o run -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-models --no-dist-rehearsal "$@" > "$out/bench_mfma.log" 2>&1
for f in stats fetch write mfma; do echo "[$f] $(tail -c 300 "$out/bench_$f.log" | tail -2)"; done
find "$out" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.json" -delete
cd "$R" && python3 profiles/tools/summarize.py "$out" > "$out/summary.txt" 2>&1
tail -40 "$out/summary.txt"
