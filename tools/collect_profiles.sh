#!/bin/bash
# Measurement pass on one MI355X box: every number quoted in DESIGN.md / README.md / profiles/CURRENT.md for the current round.
#     gpurun --timeout 3600 -- 'bash tools/collect_profiles.sh r06'
# Raw output lands in gpurun_out/<round>_profiles/; tools/distill_profiles.py (run afterwards in the build container) turns it into
# the tracked files under profiles/.  EVERY step goes through run(): a step that exits non-zero -- an A/B tool whose variants
# disagree, a bench whose tables differ from the oracle -- is recorded in <out>/FAILED with its stderr kept beside its output, and
# the script exits 1 at the end (round 4 sent stderr to /dev/null and had no failure path at all: VERDICT r4 #3).
RND=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${RND}_profiles
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() {         # run <output file> <command...>: stdout -> file, stderr -> file.err (removed when empty), failures -> FAILED
  local name=$1; shift
  "$@" > $OUT/$name 2> $OUT/$name.err
  local rc=$?
  grep -v "amdgpu.ids" $OUT/$name.err > $OUT/$name.err2; mv $OUT/$name.err2 $OUT/$name.err
  if [ $rc -ne 0 ]; then echo "FAILED rc=$rc: $name <- $*" >> $OUT/FAILED; tail -5 $OUT/$name.err >> $OUT/FAILED; fi
  [ -s $OUT/$name.err ] || rm -f $OUT/$name.err
  return 0
}
prof() {        # prof <trace dir> <rocprofv3 options...> -- <program...>  (the program itself behind --, never a shell)
  local dir=$1; shift
  timeout 900 rocprofv3 "$@" > $OUT/$dir.log 2>&1
  local rc=$?
  if [ $rc -ne 0 ]; then echo "FAILED rc=$rc: rocprofv3 $dir" >> $OUT/FAILED; tail -5 $OUT/$dir.log >> $OUT/FAILED; else rm -f $OUT/$dir.log; fi
  return 0
}
B="python3 $R/bench.py"
# ---- the BASELINE line (configs[1]) with the CPU baseline and the oracle check of the timed tables, then its traces / counters
# (the two counter passes come FIRST: their summary is handed to the default run, which reports it as roofline.traffic -- HBM bytes of
#  the same launch shape measured on this box in this call; separate --pmc passes with --kernel-trace only, as the guide prescribes)
prof pmc_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-images 0 --ring 16
prof pmc_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-images 0 --ring 16
run pmc_summary.txt python3 $R/tools/distill_profiles.py $RND --pmc-only $OUT/pmc_summary.json
export HALO_BENCH_PMC=$OUT/pmc_summary.json
run bench_default.json $B
unset HALO_BENCH_PMC
prof trace --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --cpu-images 0
run tail_timeline.txt python3 $R/tools/tail_timeline.py $OUT/trace
# ---- variants: shapes, branches, sources, and the value distributions that stress the selector
for v in "f32:--feat-dtype f32" "lowres_exact:--source lowres --lr-mode exact" "lowres_gram:--source lowres --lr-mode gram" \
         "c512:--channels 512 --ring 16" "ripu:--branch ripu" "hyper:--branch hyper" "pool2975:--pool-images 2975" \
         "resets_fills:--resets fills" "f32_selprio0:--feat-dtype f32 --sel-priority 0" "hyper_inline_tail:--branch hyper --tail inline"; do
  name=${v%%:*}; args=${v#*:}
  run bench_$name.json $B --cpu-images 0 $args
done
for d in late_round saturated peaked late_round+saturated+peaked plateau; do
  run bench_data_$d.json $B --cpu-images 4 --data $d
done
run bench_ripu_peaked.json $B --cpu-images 2 --branch ripu --data peaked
# the reference's DEFAULT purity under the same stress: near-tie-dense maps, each oracle-checked on 2 images
for d in gaussian peaked late_round+saturated+peaked; do
  run bench_hyper_data_$d.json $B --cpu-images 2 --branch hyper --data $d
done
for br in ripu hyper; do
  prof trace_$br --kernel-trace --stats --output-format csv -d $OUT/trace_$br -- python3 $R/bench.py --branch $br --cpu-images 0 --steps 8 --warmup 2
done
prof trace_f32 --kernel-trace --stats --output-format csv -d $OUT/trace_f32 -- python3 $R/bench.py --feat-dtype f32 --cpu-images 0 --steps 8 --warmup 2
prof trace_lowres --kernel-trace --stats --output-format csv -d $OUT/trace_lowres -- python3 $R/bench.py --source lowres --lr-mode exact --cpu-images 0 --steps 8 --warmup 2
prof trace_lowres_gram --kernel-trace --stats --output-format csv -d $OUT/trace_lowres_gram -- python3 $R/bench.py --source lowres --lr-mode gram --cpu-images 0 --steps 8 --warmup 2
# the default line four more times, consecutive processes WITHOUT the settle wait: they alternate between two plateaus
for rep in 1 2 3 4; do
  $B --cpu-images 0 --settle 0 2> /dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('run $rep: %.1f images/s  ms_per_step %.3f  k_feat_reduce %.3f ms (frac %.4f)  rest of the step %.3f ms  flat read of the same tensors %.0f GB/s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], d['ms_per_step'] - r['avg_launch_ms'], r['flat_read']['GB/s']))" >> $OUT/bench_repeats.txt
done
# (built in the container, travels with the snapshot: hipcc --offload-arch=gfx950 -O3 tools/micro/coissue.hip -o tools/micro/coissue)
[ -x $R/tools/micro/coissue ] || hipcc --offload-arch=gfx950 -O3 $R/tools/micro/coissue.hip -o $R/tools/micro/coissue
run coissue.txt $R/tools/micro/coissue
# ---- A/B tools (each ASSERTS that its variants agree bit for bit)
run ab_feat_map.txt python3 $R/tools/ab_feat_map.py
run ab_lowres_dma.txt python3 $R/tools/ab_lowres_dma.py
run gram_ab.txt python3 $R/tools/ab_gram.py
run ab_mlr_epilogue.txt python3 $R/tools/ab_mlr_epilogue.py
# (the logarithm's table from LDS against a -DHALO_LOGF_LDS=0 build reading it from device memory; the variant library travels when it
#  was built in the container: python -c "from halo_amd import _build; _build.FLAGS.append('-DHALO_LOGF_LDS=0'); _build._build_locked(False, objdir='/tmp/obj_nolds', so='halo_amd/csrc/variants/libhalo_hip_logf_global.so')")
[ -f $R/halo_amd/csrc/variants/libhalo_hip_logf_global.so ] && (cd $R && run ab_logf_table.txt bash $R/tools/ab_logf_table.sh)
# ---- selection, the RegionSelection driver, the head tail, training ops
METHODS=auto,serial run select_timing.txt python3 $R/tools/time_select.py
METHODS=auto RANGED=1 run select_timing_ranged.txt python3 $R/tools/time_select.py
run region_selection_timing.txt python3 $R/tools/time_region_selection.py
HALO_RS_GRAPH=0 run region_selection_timing_eager_launches.txt python3 $R/tools/time_region_selection.py
HALO_RS_HW_QUEUES=2 run region_selection_timing_two_queues.txt python3 $R/tools/time_region_selection.py
(cd $R && run hw_queues.txt bash $R/tools/hw_queues_ab.sh)
run head_timing.txt python3 $R/tools/time_head.py
run secondary_kernels.txt python3 $R/tools/time_secondary.py
run branches.txt python3 $R/tools/time_branches.py
run feat_alone.txt python3 $R/tools/time_feat.py
run lowres_timing.txt python3 $R/tools/time_lowres.py
run training_ops.txt python3 $R/tools/time_training_ops.py
run mlr_backward.txt python3 $R/tools/time_mlr_bwd.py
HALO_RS_FLOOR_ONLY=1 HALO_RS_TMP=/dev/shm run region_selection_host_floor_tmpfs.txt python3 $R/tools/time_region_selection.py
HALO_RS_FLOOR_ONLY=1 HALO_RS_NO_INDICATOR=1 run region_selection_host_floor_mask_only.txt python3 $R/tools/time_region_selection.py
HALO_RS_FLOOR_ONLY=1 HALO_RS_FRESH=1 run region_selection_host_floor_fresh_files.txt python3 $R/tools/time_region_selection.py
run fuzz_head.txt python3 $R/tests/fuzz_head.py 3000 21
run fuzz_parity.txt python3 $R/tests/fuzz_parity.py 3000 801
run fuzz_select.txt python3 $R/tests/fuzz_select.py 1500 41
prof trace_head_bwd --kernel-trace --stats --output-format csv -d $OUT/trace_head_bwd -- python3 $R/tools/prof_head_bwd.py
prof trace_feat_alone --kernel-trace --stats --output-format csv -d $OUT/trace_feat_alone -- python3 $R/tools/time_feat.py
prof trace_head --kernel-trace --stats --output-format csv -d $OUT/trace_head -- python3 $R/tools/time_head.py
prof trace_select --kernel-trace --stats --output-format csv -d $OUT/trace_select -- python3 $R/tools/time_select.py
# ---- counters of the low-res kernels and of HyperMLR (separate passes per counter group; --pmc only with --kernel-trace)
prof pmc_lowres --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_lowres -- python3 $R/tools/prof_lowres.py
prof pmc_lowres_clk --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lowres_clk -- python3 $R/tools/prof_lowres.py
prof pmc_lowres_insts --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_lowres_insts -- python3 $R/tools/prof_lowres.py
prof pmc_mlr --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_mlr -- python3 $R/tools/prof_mlr.py
prof pmc_mlr_clk --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mlr_clk -- python3 $R/tools/prof_mlr.py
# ---- the N > 1 code on the one GPU of the box (never a measurement: the ranks share the device over gloo)
HALO_BENCH_BACKEND=gloo HALO_BENCH_SHARE_GPU=1 run bench_world8_one_gpu_tiny.json $B --gpus 8 --height 64 --width 128 --channels 8 --batch 8 --ring 16 --warmup 2 --pool-images 2975 --cpu-images 0
run bench_pool96_one_rank.json $B --cpu-images 0 --ring 16 --pool-images 96 --dump-tables $OUT/pool96_one.npz
HALO_BENCH_BACKEND=gloo HALO_BENCH_SHARE_GPU=1 run bench_pool96_two_ranks_one_gpu.json $B --gpus 2 --cpu-images 0 --ring 16 --pool-images 96 --dump-tables $OUT/pool96_two.npz
run two_ranks_one_gpu.txt python3 - <<PY
import json, numpy as np
a, b = np.load("$OUT/pool96_one.npz"), np.load("$OUT/pool96_two.npz")
same = bool(np.array_equal(a["tables"].view(np.int64), b["tables"].view(np.int64)) and np.array_equal(a["counts"], b["counts"]))
d = json.loads([l for l in open("$OUT/bench_pool96_two_ranks_one_gpu.json") if l.startswith("{")][-1])
print("bench.py --gpus 2 (gloo, both ranks on the one GPU), 96 full-size images, 48 per rank:")
print("  gathered pool tables bit-identical to the one-rank run:", same)
print("  exchange:", d["exchange"], " sharding:", d["config"]["sharding"])
print("  per-rank roofline:", d["roofline"].get("per_rank_frac"), d["roofline"].get("per_rank_avg_launch_ms"))
assert same
PY
rm -f $OUT/pool96_one.npz $OUT/pool96_two.npz
# keep the merged output small: the per-dispatch traces are only needed for the summaries computed above / by the distiller
find $OUT -name "*kernel_trace.csv" -size +12M -delete
ls $OUT | head -120
if [ -f $OUT/FAILED ]; then echo "==== FAILED steps"; cat $OUT/FAILED; exit 1; fi
echo "collection complete, no failed step"
