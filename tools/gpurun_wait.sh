#!/bin/bash
# gpurun with patience: retries ONLY while the pool reports "no box / no slot free" (exit code 3, nothing ran, nothing charged); any other
# outcome -- success, a failing command, a refusal -- is returned at once.  Usage: tools/gpurun_wait.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 20); do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 150
done
exit 3
