"""Build recipe for libhalo_hip.so (hand-written HIP for gfx950, no torch extension machinery).

    python -m halo_amd._build          # or halo_amd._build.build()

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the
resulting .so sits in-tree (git-ignored) and travels to the GPU box with the snapshot.
Flags that matter for correctness: -ffp-contract=off (the numeric contract writes every fma
explicitly) and correctly rounded float32 divide/sqrt (hipcc's default, stated anyway).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "libhalo_hip.so")
SOURCES = ["halo_api.hip", "halo_score.hip", "halo_select.hip", "halo_select_binned.hip", "halo_hyperbolic.hip", "halo_loss.hip", "halo_pool.hip"]
HOST_SO = os.path.join(CSRC, "libhalo_host.so")          # plain C host helpers (PNG writer of the persistence step), built with gcc
HOST_SOURCES = ["halo_host.c"]
HEADERS = ["halo_common.hpp", "halo_devmath.hpp", "halo_select_common.hpp", "halo_select_plan.hpp", os.path.join("..", "..", "include", "halo_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function"]
# halo_score.hip: no SLP vectorisation.  Packing the per-class float32 chains of the entropy code into v_pk_* saves 10 % of
# its instructions but costs 40 VGPRs (operand pairs, constants held in registers): 116-120 instead of 79, i.e. 4 instead
# of 6 waves per SIMD for the fused feature kernel, and the stand-alone logit kernel runs 8 % slower packed.
EXTRA_FLAGS = {"halo_score.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libhalo_hip.so cannot be built")


def is_stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__), os.path.join(HERE, "_asmcheck.py")]
    return any(os.path.getmtime(d) > t for d in deps)


def host_is_stale():
    if not os.path.exists(HOST_SO):
        return True
    t = os.path.getmtime(HOST_SO)
    return any(os.path.getmtime(p) > t for p in [os.path.join(CSRC, f) for f in HOST_SOURCES] + [os.path.join(HERE, "..", "include", "halo_host.h")])


def build_host(force=False):
    """libhalo_host.so: the CPU-side helpers (no HIP), compiled with the system C compiler."""
    if not force and not host_is_stale():
        return HOST_SO
    import fcntl
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    with open(os.path.join(CSRC, "build", ".lock_host"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not host_is_stale():
                return HOST_SO
            cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc") or _hipcc()
            tmp = HOST_SO + ".tmp.%d" % os.getpid()
            cmd = [cc, "-O3", "-std=c11", "-fPIC", "-shared", "-pthread", "-Wall"] + [os.path.join(CSRC, f) for f in HOST_SOURCES] + ["-o", tmp]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError("host library build failed:\n%s" % r.stdout)
            os.replace(tmp, HOST_SO)
            return HOST_SO
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def build(force=False, verbose=False):
    """Compile every .hip translation unit for gfx950 and link libhalo_hip.so.

    Safe under concurrent callers (one process per GPU under torch.distributed.run all import the
    package at once): the build runs under an exclusive file lock and re-checks staleness inside it."""
    if not force and not is_stale():
        return SO
    import fcntl
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    with open(os.path.join(CSRC, "build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():
                return SO
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


class AsmCheckError(RuntimeError):
    """the emitted device assembly breaks a contract of the hand-scheduled inline assembly (halo_amd/_asmcheck.py)"""


def device_asm_path(objdir, src):
    """where `-save-temps=obj` leaves the device assembly of the compile that produced the object file"""
    return os.path.join(objdir, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def check_device_asm(path, src):
    """The gate between compile and link: a translation unit that hand-issues LDS reads / LDS-DMA in inline assembly is scanned on
    the assembly hipcc JUST emitted for it (registers with a read in flight, counted waits of the DMA double buffer); a hit refuses
    the build -- another hipcc, other flags or HALO_LR8_WAVES can bring the round-4 race back silently.
    HALO_ASMCHECK=warn reports and carries on, =off skips (debugging aids; never set by the package)."""
    mode = os.environ.get("HALO_ASMCHECK", "enforce")
    if mode == "off":
        return None
    import io
    from . import _asmcheck
    if not os.path.exists(path):
        raise AsmCheckError("%s: the compiler left no device assembly at %s to check" % (src, path))
    buf = io.StringIO()
    with open(path) as f:
        checked, flagged = _asmcheck.check_text(f.read(), out=buf)
    if checked == 0:
        raise AsmCheckError("%s hand-issues LDS reads / LDS-DMA but no function of %s carries them: the scan is looking at the wrong file" % (src, path))
    if flagged:
        msg = "%s: %d violation(s) in %d hand-scheduled function(s) of the emitted assembly:\n%s" % (src, flagged, checked, buf.getvalue()[:6000])
        if mode == "warn":
            sys.stderr.write("halo_amd._build: WARNING, " + msg + "\n")
        else:
            raise AsmCheckError(msg + "\nlibhalo_hip.so was NOT linked (HALO_ASMCHECK=warn overrides)")
    return checked, flagged


def _build_locked(verbose, objdir=None, so=None):
    hipcc = _hipcc()
    objdir = objdir or os.path.join(CSRC, "build")
    so = so or SO
    os.makedirs(objdir, exist_ok=True)
    from . import _asmcheck
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        with open(os.path.join(CSRC, src)) as f:
            scan = _asmcheck.source_needs_check(f.read()) and os.environ.get("HALO_ASMCHECK", "enforce") != "off"
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-save-temps=obj"] if scan else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, obj, scan, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for src, obj, scan, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
        if verbose and out.strip():
            print(out)
        if scan:
            res = check_device_asm(device_asm_path(objdir, src), src)
            if verbose and res:
                print("%s: %d hand-scheduled function(s) scanned, %d flagged" % (src, res[0], res[1]))
            stem = src.replace(".hip", "")
            for fn in os.listdir(objdir):              # the temporaries of -save-temps (bitcode, preprocessed source, 10 MB of assembly);
                if fn.startswith((stem + "-", stem + ".hip-")):      # a flagged .s stays behind for inspection (the raise above)
                    os.remove(os.path.join(objdir, fn))
        objs.append(obj)
    tmp = so + ".tmp.%d" % os.getpid()
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout)
    os.replace(tmp, so)
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv))
