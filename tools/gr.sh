#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged).  usage: tools/gr.sh TIMEOUT 'command'
t=$1; shift
for i in $(seq 1 60); do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 45
done
exit 3
