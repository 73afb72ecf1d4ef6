mkdir -p gpurun_out/r05f
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "head or hypermlr or resize or bilinear or v2 or golden or fused_tail or config1" > gpurun_out/r05f/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05f/tests.log
python tools/time_head.py > gpurun_out/r05f/head_timing.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05f/trace_nt -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-images 0 --steps 8 --warmup 2 > /dev/null 2>&1
HALO_COMBINE_NO_NT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05f/trace_nont -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-images 0 --steps 8 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r05f -name "*kernel_trace.csv" -delete
python bench.py --cpu-images 0 > gpurun_out/r05f/bench_nt.json 2>/dev/null
HALO_COMBINE_NO_NT=1 python bench.py --cpu-images 0 > gpurun_out/r05f/bench_nont.json 2>/dev/null
tail -3 gpurun_out/r05f/tests.log
