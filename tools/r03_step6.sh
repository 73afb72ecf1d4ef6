#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; echo "all rc=$?" >> $OUT/pytest_all.log
tail -n 4 $OUT/pytest_all.log
timeout 600 python bench.py --cpu-images 0 > $OUT/bench6.json 2>> $OUT/bench_err.log
python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench6.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
bash tools/r03_gram_fullsize.sh 32 | tail -n 3
