"""Register / LDS / occupancy table of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py halo_score.hip [filter]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halo_amd import _build

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = [_build._hipcc()] + _build.FLAGS + _build.EXTRA_FLAGS.get(src, []) + ["-Rpass-analysis=kernel-resource-usage", "-c",
       os.path.join(_build.CSRC, src), "-o", "/tmp/_kr.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, d = None, {}
for ln in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); d[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); d[cur][k.strip()] = v.strip()
names = subprocess.run(["c++filt"] + list(d), capture_output=True, text=True).stdout.splitlines()
for (k, v), name in zip(d.items(), names):
    name = name.split("(")[0].replace("void ", "")
    if flt in name:
        print("%-64s VGPR %4s AGPR %3s SGPR %4s occ %s LDS %6s scratch %s" % (name[:64], v.get("VGPRs"), v.get("AGPRs"), v.get("TotalSGPRs"),
              v.get("Occupancy [waves/SIMD]"), v.get("LDS Size [bytes/block]"), v.get("ScratchSize [bytes/lane]")))
