#!/bin/bash
# gpurun with retries while every GPU slot of the pod is busy (exit code 3: nothing charged).  usage: scripts/gpu/retry.sh <timeout> '<command>'
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
