#!/bin/bash
# retries a gpurun call while the pod's GPU slots are busy (exit code 3: nothing charged); usage: gpurun_retry.sh <timeout_s> '<command>'
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 45
done
exit 3
