#!/bin/bash
# rocprofv3 kernel-trace statistics of one bench step (run on the GPU box through gpurun).  Usage: tools/rocprof_bench.sh <tag> [bench args...]
# Writes gpurun_out/<tag>_kernel_stats.csv (copy the ones to be judged into profiles/).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf "$out"
rocprofv3 --kernel-trace --stats -d "$out" -o "$tag" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --steps 1 --warmup 1 "$@" > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.log" 2>&1
rc=$?
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv"
# ROCm 7.2 writes a rocpd database instead of CSV files: export its top_kernels view (name, calls, total us, average us, %)
db=$(find "$out" -name "*.db" | head -1)
if [ -z "$f" ] && [ -n "$db" ]; then
    python3 - "$db" "$GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv" <<'PY'
import csv, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.execute("select * from top_kernels")
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow([d[0] for d in cur.description])
    w.writerows(cur)
PY
fi
tail -1 "$GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.log" | cut -c1-300
exit $rc
