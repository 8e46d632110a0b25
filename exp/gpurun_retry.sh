#!/bin/bash
# usage: exp/gpurun_retry.sh <timeout> '<command>'   -- resubmits while the pod has no free GPU slot (exit code 3: nothing charged)
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
