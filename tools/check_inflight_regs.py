"""Static check of the device assembly of the low-res kernels: between a hand-issued ds_read_b64 and the s_waitcnt that covers it, no
VALU instruction may touch the read's destination registers.  The compiler takes an inline-asm output for a value that exists when
the statement ends, so it is free to copy or spill such a register while the LDS data is still in flight (it did once, round 4:
phi copies of the first channel's registers in front of the wait).  Linear scan per function (it does not follow branches: a
conservative approximation that flags the pattern that occurred).
    python tools/check_inflight_regs.py [halo_score.s]      # without an argument: compiles halo_amd/csrc/halo_score.hip to assembly first
Exit code 1 if anything is flagged."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(text, pattern=r"k_feat_reduce_lr"):
    total = 0
    for m in re.finditer(r"^(_ZN4halo\w+):\s*;", text, re.M):
        name = m.group(1)
        if not re.search(pattern, name):
            continue
        tail = text[m.end():]
        body = tail[:tail.index(".Lfunc_end")].splitlines()
        pending, bad = [], 0
        for ln in body:
            t = ln.strip()
            if not t or t[0] in ";.":
                continue
            op = t.split()[0]
            if op == "ds_read_b64":
                pending.append(regs(t.split()[1]))
                continue
            w = re.match(r"s_waitcnt.*lgkmcnt\((\d+)\)", t)
            if w:
                n = int(w.group(1))
                pending = [] if n == 0 else pending[max(0, len(pending) - n):]
                continue
            if op.startswith("v_") or op.startswith("scratch_") or op.startswith("global_store") or op.startswith("ds_write"):
                used = set()
                for o in re.split(r"[ ,]+", t)[1:]:
                    used |= regs(o)
                if pending and used & set().union(*pending):
                    bad += 1
                    if bad <= 3:
                        print("  %s: touches a register whose LDS read is in flight: %s" % (name, t))
        total += bad
        print("%-110s %d" % (name, bad))
    return total


if __name__ == "__main__":
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        sys.path.insert(0, ROOT)
        from halo_amd import _build
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "halo_score.s")
            cmd = [_build._hipcc()] + [f for f in _build.FLAGS if f != "-fPIC"] + _build.EXTRA_FLAGS.get("halo_score.hip", []) + \
                  ["--cuda-device-only", "-S", "-o", out, os.path.join(_build.CSRC, "halo_score.hip")]
            subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            text = open(out).read()
    n = check(text)
    print("flagged:", n)
    sys.exit(1 if n else 0)
