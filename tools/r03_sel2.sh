#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pool.py -x -q -k "select or golden or region_selection or full_size or binned or plateau or round_state or two_ranks or bench or range" > $OUT/pytest_sel.log 2>&1; echo "rc=$?" >> $OUT/pytest_sel.log
tail -n 5 $OUT/pytest_sel.log
bash tools/r03_sel.sh
cd $R
for i in 1 2; do
timeout 600 python bench.py --cpu-images 0 > $OUT/bench9.json 2>> $OUT/bench_err.log
python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench9.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
timeout 600 python bench.py --cpu-images 0 --source lowres > $OUT/bench9l.json 2>> $OUT/bench_err.log
python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench9l.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['lowres_passes_ms'])"
