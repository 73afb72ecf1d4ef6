mkdir -p gpurun_out/r05e
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "head or hypermlr or resize or bilinear or v2 or golden" > gpurun_out/r05e/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05e/tests.log
python tools/time_secondary.py > gpurun_out/r05e/secondary_kernels.txt 2>&1
HALO_BILINEAR_LDS1=1 python tools/time_secondary.py 2>&1 | grep "bilinear float32" > gpurun_out/r05e/secondary_lds1.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05e/trace_default -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-images 0 --steps 8 --warmup 2 > /dev/null 2>&1
HALO_NO_FUSE_HIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05e/trace_nohist -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-images 0 --steps 8 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r05e -name "*kernel_trace.csv" -delete
tail -3 gpurun_out/r05e/tests.log
