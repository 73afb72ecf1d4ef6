#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/trace_ripu
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ripu -- python3 $R/bench.py --branch ripu --cpu-images 0 --steps 8 --warmup 2 > $OUT/bench_ripu_prof.json 2>/dev/null
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace_ripu/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-50s calls %4s avg %9.1f us" % (r["Name"].split("(")[0].replace("void ", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -c 400 $OUT/bench_ripu_prof.json
find $OUT/trace_ripu -name "*.csv" -size +8M -delete
