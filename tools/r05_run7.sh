mkdir -p gpurun_out/r05g
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "head or hypermlr or resize or bilinear or v2 or golden or fuzz or expmap or variants or gradient or autograd" > gpurun_out/r05g/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05g/tests.log
python tools/time_head.py > gpurun_out/r05g/head_timing.txt 2>&1
HALO_EXPMAP_NOREGS=1 python tools/time_head.py 2>&1 | grep "expmap" > gpurun_out/r05g/head_timing_tile.txt
HALO_BILINEAR_ROWS=1 python tools/time_head.py 2>&1 | grep "bilinear" > gpurun_out/r05g/head_timing_bl_gather.txt
HALO_BILINEAR_LDS1=1 python tools/time_head.py 2>&1 | grep "bilinear" > gpurun_out/r05g/head_timing_bl_lds1.txt
python tools/time_secondary.py 2>&1 | grep hypermlr > gpurun_out/r05g/secondary_mlr.txt
tail -3 gpurun_out/r05g/tests.log
