#!/bin/bash
# Round-2 measurement pass on one MI355X box: every number quoted in DESIGN.md / README.md / profiles/README.md.
# Run through gpurun from the repository root; raw output lands in gpurun_out/r02/, tools/distill_profiles.py
# (run afterwards in the build container) turns it into the tracked files under profiles/.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --cpu-images 0 > $OUT/bench_under_rocprof.json 2> /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-images 0 --ring 16 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-images 0 --ring 16 > /dev/null 2>&1
for v in "f32:--feat-dtype f32" "lowres:--source lowres" "lowres_gram:--source lowres --lr-mode gram" "c512:--channels 512 --ring 16" "ripu:--branch ripu" "hyper:--branch hyper" "pool2975:--pool-images 2975"; do
  name=${v%%:*}; args=${v#*:}
  python3 $R/bench.py --cpu-images 0 $args > $OUT/bench_$name.json 2> /dev/null
done
for br in ripu hyper; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$br -- python3 $R/bench.py --branch $br --cpu-images 0 --steps 8 --warmup 2 > /dev/null 2>&1
done
python3 $R/tools/time_select.py > $OUT/select_timing.txt 2>&1
MRAD=3 python3 $R/tools/time_select.py > $OUT/select_timing_mrad3.txt 2>&1
python3 $R/tools/time_region_selection.py > $OUT/region_selection_timing.txt 2>&1
python3 $R/tools/time_secondary.py > $OUT/secondary_kernels.txt 2>&1
python3 $R/tools/time_branches.py > $OUT/branches.txt 2>&1
python3 $R/tools/time_feat.py > $OUT/feat_alone.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_feat_alone -- python3 $R/tools/time_feat.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lowres -- python3 $R/bench.py --source lowres --cpu-images 0 --steps 8 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lowres_gram -- python3 $R/bench.py --source lowres --lr-mode gram --cpu-images 0 --steps 8 --warmup 2 > /dev/null 2>&1
python3 $R/tools/time_lowres.py > $OUT/lowres_timing.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_f32 -- python3 $R/bench.py --feat-dtype f32 --cpu-images 0 --steps 8 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_lowres -- python3 $R/tools/prof_lowres.py > /dev/null 2>&1
python3 $R/tools/time_training_ops.py > $OUT/training_ops.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_mlr -- python3 $R/tools/prof_mlr.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mlr_clk -- python3 $R/tools/prof_mlr.py > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_select -- python3 $R/tools/time_select.py > /dev/null 2>&1
ls -R $OUT | head -80
