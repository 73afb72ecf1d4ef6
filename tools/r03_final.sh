#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
cd $R
python bench.py --height 128 --width 256 --channels 16 --batch 4 --ring 8 --steps 1 --warmup 0 --cpu-images 0 | tail -c 300; echo
python bench.py --height 128 --width 256 --channels 16 --batch 4 --ring 8 --steps 2 --warmup 1 --cpu-images 1 --depth 1 | tail -c 200; echo
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r03/pytest_all.log 2>&1; echo "all rc=$?" >> gpurun_out/r03/pytest_all.log
tail -n 4 gpurun_out/r03/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/collect_profiles.sh r03 > gpurun_out/r03/collect.log 2>&1
tail -n 3 gpurun_out/r03_profiles/two_ranks_one_gpu.txt
