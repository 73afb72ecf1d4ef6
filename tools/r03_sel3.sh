#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pool.py -x -q -k "select or golden or region_selection or full_size or binned or plateau or round_state or two_ranks or bench or range or graph" > $OUT/pytest_sel.log 2>&1; echo "rc=$?" >> $OUT/pytest_sel.log
tail -n 5 $OUT/pytest_sel.log
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/trace_sel16
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_sel16 -- python3 $R/tools/prof_select16.py > /dev/null 2>&1
python3 $R/tools/prof_select16.py summarize $OUT/trace_sel16 | tee $OUT/select16_breakdown.txt
cd $R
METHODS=auto RANGED=1 python3 tools/time_select.py 2>&1 | grep -v amdgpu
for i in 1 2; do
timeout 600 python bench.py --cpu-images 0 > $OUT/bench11.json 2>> $OUT/bench_err.log
python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench11.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['ms_per_step']-d['roofline']['avg_launch_ms'],3))"
done
find $OUT/trace_sel16 -name "*.csv" -size +8M -delete
