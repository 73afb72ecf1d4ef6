mkdir -p gpurun_out/r05j
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pool.py -q -m gpu -x -k "region_selection or sharded or visualize or install or graph" > gpurun_out/r05j/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05j/tests.log
python tools/time_region_selection.py > gpurun_out/r05j/region_selection_timing.txt 2>&1
GPU_MAX_HW_QUEUES=4 python tools/time_region_selection.py 2>&1 | head -14 > gpurun_out/r05j/region_selection_timing_q4.txt
GPU_MAX_HW_QUEUES=8 python tools/time_region_selection.py 2>&1 | head -14 > gpurun_out/r05j/region_selection_timing_q8.txt
tail -3 gpurun_out/r05j/tests.log
