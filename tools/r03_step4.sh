#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; echo "all rc=$?" >> $OUT/pytest_all.log
tail -n 4 $OUT/pytest_all.log
METHODS=auto python tools/time_select.py > $OUT/select_timing.txt 2>&1; cat $OUT/select_timing.txt
rm -f $OUT/bench4_*.json
for r in undo kernel undo kernel; do
  timeout 600 python bench.py --cpu-images 0 --resets $r >> $OUT/bench4_$r.json 2>> $OUT/bench_err.log
done
cat $OUT/bench4_*.json | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['state_resets'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['ms_per_step'] - d['roofline']['avg_launch_ms'], 3))
"
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/trace_undo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_undo -- python3 $R/bench.py --cpu-images 0 > $OUT/bench_trace_undo.json 2> /dev/null
python3 $R/tools/tail_timeline.py $OUT/trace_undo > $OUT/tail_undo.txt 2>&1
cut -c1-90 $OUT/trace_undo/*/*kernel_stats.csv | head -24
find $OUT -name "*kernel_trace.csv" -size +20M -delete
