#!/bin/bash
# SQ counters + clock of the low-res kernels (separate --pmc passes, kernel trace only)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/pmc_lowres $OUT/pmc_lowres_clk $OUT/pmc_lowres_mem
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_lowres -- python3 $R/tools/prof_lowres.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lowres_clk -- python3 $R/tools/prof_lowres.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_lowres_mem -- python3 $R/tools/prof_lowres.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r03"
def agg(d):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/" + d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            res[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return res
def dur(d):
    res = collections.defaultdict(list)
    for f in glob.glob(out + "/" + d + "/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            res[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return res
a, c, m = agg("pmc_lowres"), agg("pmc_lowres_clk"), agg("pmc_lowres_mem")
dc = dur("pmc_lowres_clk")
for k in a:
    if "_lr" not in k and "gram" not in k:
        continue
    v = {n: sum(x) / len(x) for n, x in a[k].items()}
    line = "%-44s" % k[:44]
    if k in c:
        gui = sum(c[k]["GRBM_GUI_ACTIVE"]) / len(c[k]["GRBM_GUI_ACTIVE"]) / 8
        ms = sorted(dc[k])[len(dc[k]) // 2]
        clk = gui / (ms * 1e-3) / 1e9
        line += " ms %.3f clock %.2f GHz  VALU-active/SIMD-cycles %.2f" % (ms, clk, v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * gui))
    line += "  wave: valu %.2f lds %.2f wait_any %.2f wait_inst %.2f" % (v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], v["SQ_ACTIVE_INST_LDS"] / v["SQ_WAVE_CYCLES"],
                                                                     v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"])
    if k in m:
        mm = {n: sum(x) / len(x) for n, x in m[k].items()}
        line += "  insts valu %.0fM lds %.0fM salu %.0fM  lds bank-conflict/active %.3f" % (mm.get("SQ_INSTS_VALU", 0) / 1e6, mm.get("SQ_INSTS_LDS", 0) / 1e6, mm.get("SQ_INSTS_SALU", 0) / 1e6,
                                                                                        mm.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, mm.get("SQ_LDS_IDX_ACTIVE", 1)))
    print(line)
PY
find $OUT/pmc_lowres $OUT/pmc_lowres_clk $OUT/pmc_lowres_mem -name "*.csv" -size +8M -delete
