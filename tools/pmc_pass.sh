#!/bin/bash
# One rocprofv3 --pmc pass (kernel trace only) over tools/one_kernel.py; prints the counters of the LAST dispatch of every cf:: kernel.
# Usage (GPU box): tools/pmc_pass.sh <tag> "<COUNTER ...>" <one_kernel.py args...>
# Every rocprofv3 call runs under its own `timeout -k 10 ${PMC_TIMEOUT:-180}`: a counter set the hardware cannot collect in one pass makes
# rocprofv3 abort (error 38) and then sit in its finaliser until something kills it (gpurun_out/r03/call22.txt: 5 minutes to the step limit);
# with the timeout the pass returns non-zero within three minutes and the caller's `&&` chain stops.  The program stays directly after `--`.
tag=$1; counters=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf "$out"
timeout -k 10 "${PMC_TIMEOUT:-180}" rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$out" -o pmc -- python3 "$GRAFT_REPO_ROOT/tools/one_kernel.py" "$@" > "$out.log" 2>&1 || { rc=$?; echo "pmc pass $tag FAILED (rc=$rc; 124/137 = killed by its timeout)"; grep -m3 -E "error code|exceeds the capabilities" "$out.log"; tail -3 "$out.log" | cut -c1-300; exit 1; }
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
