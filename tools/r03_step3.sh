#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_tail or bench or golden or helper or full_size" > $OUT/pytest_tail.log 2>&1; echo "tail rc=$?" >> $OUT/pytest_tail.log
for r in kernel undo fills kernel undo; do
  timeout 600 python bench.py --cpu-images 0 --resets $r >> $OUT/bench3_$r.json 2>> $OUT/bench_err.log
done
cd /tmp; export TMPDIR=/tmp
for r in undo; do
  rm -rf $OUT/trace_$r
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$r -- python3 $R/bench.py --cpu-images 0 --resets $r > $OUT/bench_trace_$r.json 2> /dev/null
  python3 $R/tools/tail_timeline.py $OUT/trace_$r > $OUT/tail_$r.txt 2>&1
done
cat $OUT/tail_undo.txt
tail -n 5 $OUT/pytest_tail.log
cat $OUT/bench3_*.json | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['state_resets'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['ms_per_step'] - d['roofline']['avg_launch_ms'], 3))
"
find $OUT -name "*kernel_trace.csv" -size +20M -delete
