#!/bin/bash
# build container only: retry a gpurun call while the pool has no free slot (exit code 3: nothing charged)
# usage: tools/gpurun_retry.sh <timeout seconds> '<command>'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
