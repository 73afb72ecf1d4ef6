#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 900 python tools/ab_lowres_dma.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_lowres_dma.txt
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -k "lowres or narrow or region_selection" > $OUT/pytest_lowres.log 2>&1; echo "rc=$?" >> $OUT/pytest_lowres.log
tail -n 4 $OUT/pytest_lowres.log
timeout 600 python bench.py --cpu-images 0 --source lowres --lr-mode exact > $OUT/bench_lowres_exact.json 2>> $OUT/bench_err.log
python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench_lowres_exact.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline'], d['lowres_passes_ms'])"
