mkdir -p gpurun_out/r05c
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hand_over or sweep or select or binned or lowres_exact_mode or staging_variants" > gpurun_out/r05c/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05c/tests.log
python bench.py --data plateau --cpu-images 4 > gpurun_out/r05c/bench_plateau.json 2> gpurun_out/r05c/bench_plateau.err
python bench.py --source lowres > gpurun_out/r05c/bench_lowres.json 2> gpurun_out/r05c/bench_lowres.err
python bench.py > gpurun_out/r05c/bench_default.json 2> gpurun_out/r05c/bench_default.err
python tools/ab_lowres_dma.py > gpurun_out/r05c/ab_lowres_dma.txt 2>&1
echo "ab rc=$?" >> gpurun_out/r05c/ab_lowres_dma.txt
tail -3 gpurun_out/r05c/tests.log
