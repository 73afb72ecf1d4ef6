#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_pool.py -x -q > $OUT/pytest_pool.log 2>&1; echo "pool rc=$?" >> $OUT/pytest_pool.log
cd /tmp; export TMPDIR=/tmp
for r in kernel undo; do
  rm -rf $OUT/trace_$r
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$r -- python3 $R/bench.py --cpu-images 0 --resets $r > $OUT/bench_trace_$r.json 2> /dev/null
  python3 $R/tools/tail_timeline.py $OUT/trace_$r > $OUT/tail_$r.txt 2>&1
done
cat $OUT/tail_kernel.txt $OUT/tail_undo.txt
tail -n 5 $OUT/pytest_pool.log
# keep the merged output small: only the stats csv
find $OUT/trace_kernel $OUT/trace_undo -name "*kernel_trace.csv" -size +20M -delete
