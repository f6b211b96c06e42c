#!/bin/bash
# HBM traffic of ONE kernel from the PMC counters, two separate passes (FETCH_SIZE, WRITE_SIZE) as the pool requires: kernel-trace only.
# Usage (on the GPU box): tools/pmc_traffic.sh <tag> <one_kernel.py args...>   -> gpurun_out/pmc_<tag>_{fetch,write}.txt (name, counter value per dispatch)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$c
    rm -rf "$out"
    timeout -k 10 "${PMC_TIMEOUT:-180}" rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out" -o pmc -- python3 "$GRAFT_REPO_ROOT/tools/one_kernel.py" "$@" > "$out.log" 2>&1 || { tail -5 "$out.log"; exit 1; }
    f=$(find "$out" -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$c" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == sys.argv[2]:
        agg[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("%s %-90s dispatches %d  last %.6g  mean %.6g" % (sys.argv[2], k, len(v), v[-1], sum(v) / len(v)))
PY
done
