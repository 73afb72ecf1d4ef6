#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "lowres or gram or region_selection or narrow" -s > $OUT/pytest_lowres.log 2>&1; echo "lowres rc=$?" >> $OUT/pytest_lowres.log
tail -n 6 $OUT/pytest_lowres.log
grep "gram vs upsample" $OUT/pytest_lowres.log
for m in gram exact; do
  timeout 600 python bench.py --cpu-images 0 --source lowres --lr-mode $m > $OUT/bench_lowres_$m.json 2>> $OUT/bench_err.log
  python3 -c "
import json,sys
d=json.loads([l for l in open('$OUT/bench_lowres_$m.json') if l.startswith('{')][-1]); print('$m', d['value'], d['ms_per_step'], d['roofline'], d.get('lowres_passes_ms'))"
done
