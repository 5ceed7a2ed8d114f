#!/bin/bash
# (build container) gpurun with retries while every GPU slot of the pod is busy (exit 3: nothing charged)
# usage: tools/gpu_retry.sh <timeout seconds> '<command>'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
