#!/bin/bash
# Runs a list of GPU steps in one gpurun call; stops at the first step that was killed by its time limit (never starts another GPU step
# after a kill), but carries on after an ordinary test failure so that one call returns as much information as possible.
# Usage: tools/gpu_session.sh "<seconds> <logfile> <command...>" ...
mkdir -p gpurun_out
for step in "$@"; do
    set -- $step
    secs=$1; log=$2; shift 2
    mkdir -p "$(dirname "gpurun_out/$log")"
    echo "=== [$(date +%H:%M:%S)] $* (limit ${secs}s) -> gpurun_out/$log"
    timeout -k 10 "$secs" "$@" > "gpurun_out/$log" 2>&1
    rc=$?
    echo "    rc=$rc"
    tail -n 6 "gpurun_out/$log" | cut -c1-400
    if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): stopping"; exit $rc; fi
done
exit 0
