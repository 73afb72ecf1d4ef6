mkdir -p gpurun_out/r05l
echo "== default (f32 x2 rows for mild magnification)" > gpurun_out/r05l/f32v2.txt; python tools/time_head.py 2>&1 | grep bilinear >> gpurun_out/r05l/f32v2.txt
echo "== HALO_BILINEAR_F32V4=1" >> gpurun_out/r05l/f32v2.txt; HALO_BILINEAR_F32V4=1 python tools/time_head.py 2>&1 | grep "bilinear f32" >> gpurun_out/r05l/f32v2.txt
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "resize or bilinear or head or golden or region_selection" > gpurun_out/r05l/tests.log 2>&1; tail -2 gpurun_out/r05l/tests.log
