#!/bin/bash
# Retry a gpurun call while the pod's GPU slots are busy (exit code 3 = nothing charged).  Usage: scripts/gpu_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
