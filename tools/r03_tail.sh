#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_tail or golden or vs_oracle or narrow or range" > $OUT/pytest_tail.log 2>&1; echo "rc=$?" >> $OUT/pytest_tail.log
tail -n 4 $OUT/pytest_tail.log
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/trace_tail
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_tail -- python3 $R/bench.py --cpu-images 0 > $OUT/bench_trace_tail.json 2> /dev/null
python3 $R/tools/tail_timeline.py $OUT/trace_tail
cd $R
for i in 1 2; do
timeout 600 python bench.py --cpu-images 0 > $OUT/bench10.json 2>> $OUT/bench_err.log
python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench10.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['ms_per_step']-d['roofline']['avg_launch_ms'],3))"
done
find $OUT/trace_tail -name "*.csv" -size +8M -delete
