#!/bin/bash
# Per-kernel time of one command on an MI355X box: tools/kstats.sh <tag> <program> [args...]  (run through gpurun from the repo root).
# rocprofv3 --kernel-trace --stats; prints the top kernels (calls, average, total) and keeps the CSV under gpurun_out/kstats/<tag>/.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kstats/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$@" > $OUT/stdout.txt 2> $OUT/stderr.txt
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-90s %6s %12s %12s" % ("kernel", "calls", "avg us", "total ms"))
for r in rows[:int(__import__("os").environ.get("KSTATS_TOP", "16"))]:
    print("%-90s %6s %12.1f %12.3f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find $OUT -name "*kernel_trace.csv" -size +8M -delete
