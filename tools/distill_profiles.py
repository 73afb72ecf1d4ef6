"""Turn gpurun_out/<round>/ (tools/collect_profiles.sh on an MI355X box) into the tracked summaries under profiles/.
    python tools/distill_profiles.py [r03]"""
import re
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC = os.path.join(ROOT, "gpurun_out", RND + "_profiles")
DST = os.path.join(ROOT, "profiles")


def one(pattern):
    """newest match (gpurun merges every call's output into the same directory tree)"""
    f = glob.glob(os.path.join(SRC, pattern))
    return max(f, key=os.path.getmtime) if f else None


def copy_json(name, dst):
    p = os.path.join(SRC, name)
    if not os.path.exists(p):
        return None
    lines = [ln for ln in open(p) if ln.startswith("{")]
    if not lines:
        return None
    d = json.loads(lines[-1])
    json.dump(d, open(os.path.join(DST, dst), "w"), indent=1)
    return d


def stats_csv(pattern, dst, top=40):
    f = one(pattern)
    if not f:
        return
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(DST, dst), "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:top]:
            w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])


def counters(pattern):
    f = one(pattern)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def durations(pattern, key):
    f = one(pattern)
    out = []
    if f:
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                out.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return out


def short(name):
    return name.split("(")[0].replace("void ", "")


def pmc_summary(dst):
    """HBM traffic of the roofline kernel from the two separate --pmc passes (gfx950: FETCH_SIZE doubled) -> dst (JSON).
    `python tools/distill_profiles.py r06 --pmc-only <file>` writes it on the GPU box between the counter passes and the default
    bench run, which then reports it as roofline.traffic (HALO_BENCH_PMC=<file>: same box, same call, same launch shape)."""
    fetch, write = counters("pmc_fetch/*/*counter_collection.csv"), counters("pmc_write/*/*counter_collection.csv")
    per = {}
    feat = None
    for k in fetch:
        per[short(k)] = {"FETCH_SIZE_raw": round(sum(fetch[k]["FETCH_SIZE"]) / max(1, len(fetch[k]["FETCH_SIZE"])), 1)}
        if "k_feat_reduce" in k:
            feat = k
    for k in write:
        per.setdefault(short(k), {})["WRITE_SIZE"] = round(sum(write[k]["WRITE_SIZE"]) / max(1, len(write[k]["WRITE_SIZE"])), 1)
    if not feat:
        return None
    f_kb = sum(fetch[feat]["FETCH_SIZE"]) / len(fetch[feat]["FETCH_SIZE"])
    wk = [k for k in write if "k_feat_reduce" in k][0]
    w_kb = min(write[wk]["WRITE_SIZE"])
    B, H, W, C, O = 16, 1024, 2048, 256, 19
    kern = B * H * W * (C * 8 + 8 + O * 4 + 4)          # what the kernel itself moves (radius + entropy maps written)
    alg = B * H * W * (C * 8 + O * 4 + 8)               # features + logits read, one float64 score written
    hbm = int(2 * f_kb * 1024 + w_kb * 1024)
    rec = {"round": int(re.match(r'r(\d+)', RND).group(1)), "kernel": short(feat), "batch": B, "dtype": "f64", "shape_HWCO": [H, W, C, O],
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
                      "--cpu-images 0 --ring 16 (two separate passes)",
           "FETCH_SIZE_KB_avg_per_launch": f_kb, "WRITE_SIZE_KB_min_per_launch": w_kb,
           "correction": "gfx950: FETCH_SIZE reports 1/2 of a 16-B/lane coalesced streaming read (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE exact",
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "kernel_bytes_per_launch": kern,
           "traffic_over_algorithmic": round(hbm / alg, 4), "traffic_over_kernel_bytes": round(hbm / kern, 4), "per_kernel_KB": per}
    json.dump(rec, open(dst, "w"), indent=1)
    return rec


def main():
    os.makedirs(DST, exist_ok=True)
    d = copy_json("bench_default.json", RND + "_bench_default.json")
    copy_json("bench_under_rocprof.json", RND + "_bench_under_rocprof.json")
    for v in ("f32", "lowres", "lowres_exact", "lowres_gram", "c512", "ripu", "hyper", "pool2975", "resets_kernel", "resets_fills",
              "pool96_one_rank", "pool96_two_ranks_one_gpu", "r02_equivalent", "world8_one_gpu_tiny", "f32_selprio0", "hyper_inline_tail",
              "data_late_round", "data_saturated", "data_peaked", "data_late_round+saturated+peaked", "data_plateau", "ripu_peaked",
              "hyper_data_gaussian", "hyper_data_peaked", "hyper_data_late_round+saturated+peaked"):
        copy_json("bench_%s.json" % v, RND + "_bench_%s.json" % v)
    stats_csv("trace/*/*_kernel_stats.csv", RND + "_kernel_stats.csv")
    stats_csv("trace_ripu/*/*_kernel_stats.csv", RND + "_kernel_stats_ripu.csv", 25)
    stats_csv("trace_hyper/*/*_kernel_stats.csv", RND + "_kernel_stats_hyper.csv", 25)
    stats_csv("trace_select/*/*_kernel_stats.csv", RND + "_kernel_stats_select_tool.csv", 25)
    stats_csv("trace_feat_alone/*/*_kernel_stats.csv", RND + "_kernel_stats_feat_alone.csv", 12)
    stats_csv("trace_lowres/*/*_kernel_stats.csv", RND + "_kernel_stats_lowres.csv", 25)
    stats_csv("trace_f32/*/*_kernel_stats.csv", RND + "_kernel_stats_f32.csv", 25)
    stats_csv("trace_lowres_gram/*/*_kernel_stats.csv", RND + "_kernel_stats_lowres_gram.csv", 25)
    stats_csv("trace_head/*/*_kernel_stats.csv", RND + "_kernel_stats_head.csv", 25)
    stats_csv("trace_head_bwd/*/*_kernel_stats.csv", RND + "_kernel_stats_head_bwd.csv", 25)
    lr = counters("pmc_lowres/*/*counter_collection.csv")
    if lr:
        clk = counters("pmc_lowres_clk/*/*counter_collection.csv")
        ins = counters("pmc_lowres_insts/*/*counter_collection.csv")
        per = {}
        for k, c in lr.items():
            if "_lr" not in k and "gram" not in k:
                continue
            rec = {n: sum(v) / len(v) for n, v in c.items()}
            if k in ins:
                rec.update({n: sum(v) / len(v) for n, v in ins[k].items()})
            if k in clk:
                gui = sum(clk[k]["GRBM_GUI_ACTIVE"]) / len(clk[k]["GRBM_GUI_ACTIVE"]) / 8          # summed over the 8 XCDs
                d_ms = durations("pmc_lowres_clk/*/*kernel_trace.csv", short(k))
                if d_ms:
                    ms = sorted(d_ms)[len(d_ms) // 2]
                    rec["duration_ms_clock_pass"] = ms
                    rec["effective_clock_GHz"] = gui / (ms * 1e-3) / 1e9
                    # one wave's VALU instruction keeps it "active" for 4 cycles; a float64 instruction also occupies the SIMD's
                    # FP64 pipe for 4 cycles (16 lanes per cycle), so this ratio is the pipe's busy fraction for float64 kernels
                    # (float32 kernels: two waves overlap, the ratio tops out at 2)
                    rec["valu_active_cycles_per_SIMD_cycle"] = rec["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * gui)
            per[short(k)] = rec
        json.dump({"round": int(re.match(r'r(\d+)', RND).group(1)), "command": "rocprofv3 --pmc SQ_* --kernel-trace -- python3 tools/prof_lowres.py (three passes: SQ activity, "
                                                      "GRBM_GUI_ACTIVE for the clock, instruction counts)",
                   "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over all waves; GRBM_GUI_ACTIVE / 8 = shader cycles",
                   "per_kernel_avg_per_launch": per}, open(os.path.join(DST, RND + "_pmc_lowres.json"), "w"), indent=1)
    for t in ("select_timing.txt", "select_timing_mrad3.txt", "region_selection_timing.txt", "secondary_kernels.txt", "branches.txt", "training_ops.txt", "feat_alone.txt", "lowres_timing.txt", "tail_timeline.txt", "ab_lowres_dma.txt", "two_ranks_one_gpu.txt", "select_timing_ranged.txt",
              "select16_breakdown.txt", "bench_repeats.txt", "ab_feat_map.txt", "region_selection_timing_device_staging.txt",
              "region_selection_timing_python_writer.txt", "host_pieces.txt", "hw_queues.txt", "gram_ab.txt", "op_rate.txt",
              "lowres_overlap_probe.txt", "head_timing.txt", "region_selection_timing_eager_launches.txt", "region_selection_timing_two_queues.txt", "ab_mlr_epilogue.txt", "mlr_backward.txt",
              "region_selection_host_floor_tmpfs.txt", "region_selection_host_floor_mask_only.txt", "region_selection_host_floor_fresh_files.txt", "fuzz_head.txt", "fuzz_parity.txt", "fuzz_select.txt", "coissue.txt", "pmc_summary.txt", "ab_logf_table.txt", "FAILED"):
        p = os.path.join(SRC, t)
        if os.path.exists(p):
            keep = [ln for ln in open(p) if "amdgpu.ids" not in ln]
            if t == "fuzz_head.txt":
                keep = keep[-2:]
            open(os.path.join(DST, RND + "_" + t), "w").writelines(keep)
    pmc_summary(os.path.join(DST, RND + "_pmc_summary.json"))
    # ---- HyperMLR on the matrix cores
    mlr = counters("pmc_mlr/*/*counter_collection.csv")
    clk = counters("pmc_mlr_clk/*/*counter_collection.csv")
    for k in mlr:
        if "hypermlr" in k:
            c = {n: sum(v) / len(v) for n, v in mlr[k].items()}
            dur = durations("pmc_mlr/*/*kernel_trace.csv", "hypermlr")
            gui = [sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"]) for kk, v in clk.items() if "hypermlr" in kk]
            dur_clk = durations("pmc_mlr_clk/*/*kernel_trace.csv", "hypermlr")
            cycles = gui[0] / 8 if gui else None          # summed over the 8 XCDs
            util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cycles if cycles else None
            json.dump({"round": int(re.match(r'r(\d+)', RND).group(1)), "kernel": short(k), "shape": "x (1,256,1024,2048) f64, 19 classes, float32 logits",
                       "command": "rocprofv3 --pmc SQ_* --kernel-trace -- python3 tools/prof_mlr.py ; second pass --pmc GRBM_GUI_ACTIVE",
                       "counters_avg_per_launch": c, "duration_ms": dur, "duration_ms_clock_pass": dur_clk,
                       "shader_cycles_per_launch": cycles, "effective_clock_GHz": (cycles / (sum(dur_clk) / len(dur_clk) * 1e-3) / 1e9) if cycles and dur_clk else None,
                       "mfma_busy_cycles_per_SIMD": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024,
                       "mfma_utilisation": util, "note": "SQ_VALU_MFMA_BUSY_CYCLES counts cycles (64 per v_mfma_f64_16x16x4_f64) summed over 1024 SIMDs; "
                                                         "f64 MFMA and f64 VALU share the FP64 units, so the epilogue's VALU time adds to, not overlaps with, the MFMA time"},
                      open(os.path.join(DST, RND + "_mfma_head.json"), "w"), indent=1)
    print("profiles written:", sorted(f for f in os.listdir(DST) if f.startswith(RND + "_")))
    if d:
        print("default bench:", d["value"], d["unit"], "roofline", d["roofline"])


if __name__ == "__main__":
    if "--pmc-only" in sys.argv:
        r = pmc_summary(sys.argv[sys.argv.index("--pmc-only") + 1])
        print("pmc summary:", None if r is None else {k: r[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic")})
    else:
        main()
