#!/bin/bash
# One rocprofv3 --pmc pass (kernel trace only) over tools/one_kernel.py; prints the counters of the LAST dispatch of every cf:: kernel.
# Usage (GPU box): tools/pmc_pass.sh <tag> "<COUNTER ...>" <one_kernel.py args...>
tag=$1; counters=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf "$out"
rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$out" -o pmc -- python3 "$GRAFT_REPO_ROOT/tools/one_kernel.py" "$@" > "$out.log" 2>&1 || { tail -5 "$out.log"; exit 1; }
python3 - "$(find "$out" -name "*counter_collection.csv" | head -1)" "$(find "$out" -name "*kernel_trace.csv" | head -1)" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
last = {}
for r in rows:
    k = r["Kernel_Name"]
    if "cf" not in k: continue
    last.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])      # later dispatches overwrite earlier ones
dur = {}
try:
    for r in csv.DictReader(open(sys.argv[2])):
        dur[r["Kernel_Name"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
except Exception as e:
    pass
for k, v in last.items():
    print(k[:100], "last dispatch %.1f us" % dur.get(k, float("nan")))
    for c, x in sorted(v.items()):
        print("    %-40s %.6g" % (c, x))
PY
