mkdir -p gpurun_out/r05i
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pool.py -q -m gpu -x -k "region_selection or sharded or visualize or install" > gpurun_out/r05i/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05i/tests.log
python tools/time_region_selection.py > gpurun_out/r05i/region_selection_timing.txt 2>&1
HALO_RS_GRAPH=0 python tools/time_region_selection.py 2>&1 | head -14 > gpurun_out/r05i/region_selection_timing_eager.txt
tail -3 gpurun_out/r05i/tests.log
