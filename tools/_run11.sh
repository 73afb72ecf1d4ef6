mkdir -p gpurun_out/r05k
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused or golden or config1 or full_size or tail or histogram or padding or lowres_sources" > gpurun_out/r05k/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05k/tests.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05k/trace -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-images 0 --steps 8 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05k/trace_lr -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-images 0 --steps 8 --warmup 2 --source lowres > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r05k -name "*kernel_trace.csv" -delete
python bench.py --cpu-images 0 > gpurun_out/r05k/bench.json 2>/dev/null
python bench.py --cpu-images 0 --source lowres > gpurun_out/r05k/bench_lr.json 2>/dev/null
tail -3 gpurun_out/r05k/tests.log
