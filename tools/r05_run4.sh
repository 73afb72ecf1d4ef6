mkdir -p gpurun_out/r05d
python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r05d/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05d/tests.log
python tools/time_region_selection.py > gpurun_out/r05d/region_selection_timing.txt 2>&1
python tools/time_secondary.py > gpurun_out/r05d/secondary_kernels.txt 2>&1
tail -5 gpurun_out/r05d/tests.log
