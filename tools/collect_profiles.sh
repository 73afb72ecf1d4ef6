#!/bin/bash
# Measurement pass on one MI355X box: every number quoted in DESIGN.md / README.md / profiles/README.md for the current round.
# Run through gpurun from the repository root; raw output lands in gpurun_out/<round>_profiles/, tools/distill_profiles.py
# (run afterwards in the build container) turns it into the tracked files under profiles/.
RND=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${RND}_profiles
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
# A bench process that starts within a few seconds of the end of another large GPU process lands on a ~3 % slower plateau (every
# other one of back-to-back runs; profiles/r03_process_alternation.txt): a pause of 5 s before each bench run avoids it, i.e. the
# numbers below are the ones a single run on an idle GPU gets.
pause() { :; }          # bench.py waits by itself now (--settle 5, before its first GPU call)
pause
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --cpu-images 0 > $OUT/bench_under_rocprof.json 2> /dev/null
python3 $R/tools/tail_timeline.py $OUT/trace > $OUT/tail_timeline.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-images 0 --ring 16 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-images 0 --ring 16 > /dev/null 2>&1
for v in "f32:--feat-dtype f32" "lowres_exact:--source lowres --lr-mode exact" "lowres_gram:--source lowres --lr-mode gram" "c512:--channels 512 --ring 16" "ripu:--branch ripu" "hyper:--branch hyper" "pool2975:--pool-images 2975" "resets_kernel:--resets kernel" "resets_fills:--resets fills" "r02_equivalent:--resets fills --settle 0"; do
  name=${v%%:*}; args=${v#*:}
  pause
  python3 $R/bench.py --cpu-images 0 $args > $OUT/bench_$name.json 2> /dev/null
done
for br in ripu hyper; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$br -- python3 $R/bench.py --branch $br --cpu-images 0 --steps 8 --warmup 2 > /dev/null 2>&1
done
# the default line six more times, consecutive processes WITHOUT a pause: they alternate between two plateaus
for rep in 1 2 3 4 5 6; do
  python3 $R/bench.py --cpu-images 0 --settle 0 2> /dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('run $rep: %.1f images/s  ms_per_step %.3f  k_feat_reduce %.3f ms (frac %.4f)  tail %.3f ms  flat read of the same tensors %.0f GB/s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], d['ms_per_step'] - r['avg_launch_ms'], r['flat_read']['GB/s']))" >> $OUT/bench_repeats.txt
done
python3 $R/tools/ab_feat_map.py 2> /dev/null > $OUT/ab_feat_map.txt
METHODS=auto,serial python3 $R/tools/time_select.py > $OUT/select_timing.txt 2>&1
python3 $R/tools/time_region_selection.py > $OUT/region_selection_timing.txt 2>&1
HALO_RS_STAGING=device python3 $R/tools/time_region_selection.py 2>&1 | head -14 > $OUT/region_selection_timing_device_staging.txt
HALO_RETIRE_PYTHON=1 python3 $R/tools/time_region_selection.py 2>&1 | head -14 > $OUT/region_selection_timing_python_writer.txt
python3 $R/tools/time_persist.py > $OUT/host_pieces.txt 2>&1
# hardware queues, with the settled bench: two runs per value
for q in 1 2 4 8; do for rep in 1 2; do
  HALO_BENCH_HW_QUEUES=$q python3 $R/bench.py --cpu-images 0 2> /dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('GPU_MAX_HW_QUEUES=$q run $rep: %.1f images/s  ms_per_step %.3f  k_feat_reduce %.3f ms  tail %.3f ms' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], d['ms_per_step'] - r['avg_launch_ms']))" >> $OUT/hw_queues.txt
done; done
for q in 2 4; do
  HALO_BENCH_HW_QUEUES=$q python3 $R/bench.py --cpu-images 0 --source lowres 2> /dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GPU_MAX_HW_QUEUES=$q --source lowres (exact): %.1f images/s  ms_per_step %.3f' % (d['value'], d['ms_per_step']))" >> $OUT/hw_queues.txt
done
# the world-8 code on the one GPU (tests/test_gpu_pool.py runs the same): eight gloo ranks share the device, 2975-image pool, tiny shape
HALO_BENCH_BACKEND=gloo HALO_BENCH_SHARE_GPU=1 python3 $R/bench.py --gpus 8 --height 64 --width 128 --channels 8 --batch 8 --ring 16 --warmup 2 --pool-images 2975 --cpu-images 0 > $OUT/bench_world8_one_gpu_tiny.json 2> /dev/null
KSTATS_TOP=14 $R/tools/kstats.sh gram_ab python3 $R/tools/ab_gram.py > $OUT/gram_ab.txt 2>&1
python3 $R/tools/time_secondary.py > $OUT/secondary_kernels.txt 2>&1
python3 $R/tools/time_branches.py > $OUT/branches.txt 2>&1
python3 $R/tools/time_feat.py > $OUT/feat_alone.txt 2>&1
python3 $R/tools/ab_lowres_dma.py > $OUT/ab_lowres_dma.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_feat_alone -- python3 $R/tools/time_feat.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lowres -- python3 $R/bench.py --source lowres --lr-mode exact --cpu-images 0 --steps 8 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lowres_gram -- python3 $R/bench.py --source lowres --lr-mode gram --cpu-images 0 --steps 8 --warmup 2 > /dev/null 2>&1
python3 $R/tools/time_lowres.py > $OUT/lowres_timing.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_f32 -- python3 $R/bench.py --feat-dtype f32 --cpu-images 0 --steps 8 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_lowres -- python3 $R/tools/prof_lowres.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lowres_clk -- python3 $R/tools/prof_lowres.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_lowres_insts -- python3 $R/tools/prof_lowres.py > /dev/null 2>&1
python3 $R/tools/time_training_ops.py > $OUT/training_ops.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_mlr -- python3 $R/tools/prof_mlr.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mlr_clk -- python3 $R/tools/prof_mlr.py > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_select -- python3 $R/tools/time_select.py > /dev/null 2>&1
# the N > 1 code path at the full shape on the one GPU of the box: two ranks share it over gloo (never a measurement: both ranks
# stream from the same HBM); the gathered pool tables must equal the one-rank run's
pause
python3 $R/bench.py --cpu-images 0 --ring 16 --pool-images 96 --dump-tables $OUT/pool96_one.npz > $OUT/bench_pool96_one_rank.json 2> /dev/null
pause
HALO_BENCH_BACKEND=gloo HALO_BENCH_SHARE_GPU=1 python3 $R/bench.py --gpus 2 --cpu-images 0 --ring 16 --pool-images 96 --dump-tables $OUT/pool96_two.npz > $OUT/bench_pool96_two_ranks_one_gpu.json 2> $OUT/bench_pool96_two.err
python3 - <<PY > $OUT/two_ranks_one_gpu.txt 2>&1
import json, numpy as np
a, b = np.load("$OUT/pool96_one.npz"), np.load("$OUT/pool96_two.npz")
same = bool(np.array_equal(a["tables"].view(np.int64), b["tables"].view(np.int64)) and np.array_equal(a["counts"], b["counts"]))
d = json.loads([l for l in open("$OUT/bench_pool96_two_ranks_one_gpu.json") if l.startswith("{")][-1])
print("bench.py --gpus 2 (gloo, both ranks on the one GPU), 96 full-size images, 48 per rank:")
print("  gathered pool tables bit-identical to the one-rank run:", same)
print("  rows of other ranks checked against local results inside bench.py:", d["exchange"]["rows_checked_against_local_results"])
print("  exchange:", d["exchange"], " sharding:", d["config"]["sharding"])
PY
rm -f $OUT/pool96_one.npz $OUT/pool96_two.npz
METHODS=auto RANGED=1 python3 $R/tools/time_select.py > $OUT/select_timing_ranged.txt 2>&1
# issue-rate probe of the VALU (what DESIGN section 4 quotes for the lean softmax) and the two-stream overlap probe of the low-res passes
(cd $R/tools/micro && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -w -o op_rate op_rate.hip) > /dev/null 2>&1
$R/tools/micro/op_rate 150 > $OUT/op_rate.txt 2>&1
python3 $R/tools/micro/lr_overlap.py > $OUT/lowres_overlap_probe.txt 2>&1
cd /tmp
rm -rf $OUT/trace_sel16
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_sel16 -- python3 $R/tools/prof_select16.py > /dev/null 2>&1
python3 $R/tools/prof_select16.py summarize $OUT/trace_sel16 > $OUT/select16_breakdown.txt 2>&1
# keep the merged output small: the per-dispatch traces are only needed for the summaries computed above / by the distiller
find $OUT -name "*kernel_trace.csv" -size +12M -delete
ls -R $OUT | head -100
