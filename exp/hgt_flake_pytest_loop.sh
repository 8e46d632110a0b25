#!/bin/bash
# Fresh pytest processes of the HGT layer tests, N times (default 25); the full log of every failing run is kept.
# usage: hgt_flake_pytest_loop.sh [N] [-k expression]
N=${1:-25}
K=${2:-hgt}
mkdir -p gpurun_out/flake
fails=0
for i in $(seq 1 $N); do
  if ! python -m pytest tests/test_gpu_layers.py -q -m gpu -x -k "$K" > gpurun_out/flake/run_$i.log 2>&1; then
    fails=$((fails+1)); echo "run $i FAILED"; grep -n "per node type\|Mismatched\|^FAILED" gpurun_out/flake/run_$i.log
  else
    rm -f gpurun_out/flake/run_$i.log
  fi
  echo "run $i done ($fails failures so far)"
done
echo "$fails failures in $N runs"
