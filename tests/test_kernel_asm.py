"""The assembly gate of the build (halo_amd/_asmcheck.py, run by halo_amd/_build.py between compile and link; CPU only).

The low-res kernels issue LDS reads and LDS-DMA by inline assembly and wait for them in LATER statements; the compiler knows nothing
about those operations being asynchronous.  Round 4 shipped a race for a day (phi copies of registers whose LDS data was still in
flight, commit 3ae5865).  The check follows every path through the emitted code; a hit refuses the link."""
import gzip
import os
import stat
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "fixtures", "inflight_phi_copies_3ae5865.s.gz")


def _fixture_text():
    with gzip.open(FIXTURE, "rt") as f:
        return f.read()


def test_the_round4_race_is_flagged():
    """one function of the broken build (hipcc's own output for halo_score.hip at 3ae5865, comments stripped): the six v_mov_b64
    phi copies in front of the counted wait are found"""
    from halo_amd import _asmcheck
    checked, flagged = _asmcheck.check_text(_fixture_text(), out=open(os.devnull, "w"))
    assert checked == 1 and flagged >= 6
    msgs = []
    for name, body in _asmcheck._functions(_fixture_text()):
        msgs += _asmcheck.check_function(name, body) or []
    assert sum("v_mov_b64" in m and "in flight" in m for m in msgs) >= 6


def _snippet(body):
    return "f:\n" + body + "\n.Lfunc_end0:\n"


def test_checker_follows_branches_and_counts_every_lds_operation():
    from halo_amd._asmcheck import check_text
    null = open(os.devnull, "w")
    issue = "\t;;#ASMSTART\n\tds_read_b64 v[10:11], v1\n\tds_read_b64 v[12:13], v1 offset:8\n\t;;#ASMEND\n"
    # a use behind a counted wait that covers the reads: fine
    assert check_text(_snippet(issue + "\ts_waitcnt lgkmcnt(0)\n\tv_add_f64 v[2:3], v[10:11], v[12:13]\n\ts_endpgm"), out=null) == (1, 0)
    # the same use in front of the wait
    assert check_text(_snippet(issue + "\tv_add_f64 v[2:3], v[10:11], v[12:13]\n\ts_waitcnt lgkmcnt(0)\n\ts_endpgm"), out=null)[1] == 1
    # a compiler ds_write AFTER the reads is counted: lgkmcnt(1) retires both reads, lgkmcnt(2) only the first
    tail = "\tds_write_b32 v4, v5\n\ts_waitcnt lgkmcnt(%d)\n\tv_mov_b64 v[2:3], v[12:13]\n\ts_endpgm"
    assert check_text(_snippet(issue + tail % 1), out=null)[1] == 0
    assert check_text(_snippet(issue + tail % 2), out=null)[1] == 1
    # the round-4 shape: the copy sits on ONE arm of a branch between issue and wait (a linear scan of the layout order would see
    # the wait of the other arm first)
    arms = issue + "\ts_cbranch_scc1 .LBB0_2\n\ts_waitcnt lgkmcnt(0)\n\ts_branch .LBB0_3\n.LBB0_2:\n\tv_mov_b64 v[20:21], v[10:11]\n" \
                   "\ts_waitcnt lgkmcnt(0)\n.LBB0_3:\n\tv_add_f64 v[2:3], v[10:11], v[12:13]\n\ts_endpgm"
    assert check_text(_snippet(arms), out=null)[1] == 1
    # a loop whose back edge carries a pending read into the header
    loop = ".LBB0_1:\n\tv_add_f64 v[2:3], v[10:11], v[2:3]\n" + issue + "\ts_cbranch_scc1 .LBB0_1\n\ts_waitcnt lgkmcnt(0)\n\ts_endpgm"
    assert check_text(_snippet(loop), out=null)[1] >= 1


def test_counted_dma_wait_needs_a_clean_vector_memory_queue():
    from halo_amd._asmcheck import check_text
    null = open(os.devnull, "w")
    dma = "\t;;#ASMSTART\n\tglobal_load_lds_dwordx4 v[2:3], off\n\t;;#ASMEND\n"
    wait = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(2)\n\t;;#ASMEND\n"
    assert check_text(_snippet(dma * 2 + dma * 2 + wait + "\ts_endpgm"), out=null) == (1, 0)
    # a compiler load between the block and its wait shifts the count
    assert check_text(_snippet(dma * 2 + dma * 2 + "\tglobal_load_dword v9, v[4:5], off\n" + wait + "\ts_endpgm"), out=null)[1] == 1
    # a spill store in front of the blocks is still outstanding at the wait
    assert check_text(_snippet("\tscratch_store_dwordx2 off, v[8:9], off\n" + dma * 4 + wait + "\ts_endpgm"), out=null)[1] == 1
    # a second counted wait without a new block in between would leave the previous block in flight
    assert check_text(_snippet(dma * 4 + wait + wait + "\ts_endpgm"), out=null)[1] == 1
    # steady state of the double buffer: block, wait, block, wait
    assert check_text(_snippet(dma * 2 + dma * 2 + wait + dma * 2 + wait + "\ts_waitcnt vmcnt(0)\n\ts_endpgm"), out=null) == (1, 0)


def test_a_build_whose_assembly_is_flagged_does_not_link(tmp_path, monkeypatch):
    """halo_amd._build with a stand-in compiler that emits the round-4 assembly for halo_score.hip: build() raises AsmCheckError and no
    library appears (the gate sits between compile and link, on the files of the compile that would have been linked)."""
    from halo_amd import _build
    asm = tmp_path / "broken.s"
    asm.write_text(_fixture_text())
    fake = tmp_path / "hipcc"
    fake.write_text("#!%s\nimport shutil, sys\na = sys.argv[1:]\nout = a[a.index('-o') + 1]\nopen(out, 'w').write('x')\n"
                    "if '-save-temps=obj' in a:\n    src = a[a.index('-c') + 1]\n"
                    "    import os\n    stem = os.path.basename(src)[:-4]\n"
                    "    shutil.copy(%r, os.path.join(os.path.dirname(out), stem + '-hip-amdgcn-amd-amdhsa-gfx950.s'))\n"
                    % (sys.executable, str(asm)))
    fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("HIPCC", str(fake))
    monkeypatch.delenv("HALO_ASMCHECK", raising=False)
    so = tmp_path / "libhalo_hip.so"
    with pytest.raises(_build.AsmCheckError) as ei:
        _build._build_locked(False, objdir=str(tmp_path / "obj"), so=str(so))
    assert "halo_score.hip" in str(ei.value) and "in flight" in str(ei.value) and not so.exists()
    assert os.path.exists(_build.device_asm_path(str(tmp_path / "obj"), "halo_score.hip"))      # kept for inspection
    monkeypatch.setenv("HALO_ASMCHECK", "warn")                   # the documented override links anyway, loudly
    assert _build._build_locked(False, objdir=str(tmp_path / "obj"), so=str(so)) == str(so) and so.exists()


def test_the_sources_that_need_the_scan_are_the_ones_scanned():
    from halo_amd import _asmcheck, _build
    need = [s for s in _build.SOURCES if _asmcheck.source_needs_check(open(os.path.join(_build.CSRC, s)).read())]
    assert need == ["halo_score.hip"]


def test_the_current_build_passed_the_gate():
    """the in-tree library was produced by a build that ran the scan (build() is the only producer); re-scan when the compile's
    assembly is still around, otherwise compile halo_score.hip to assembly here (about a minute)"""
    import subprocess
    import tempfile
    from halo_amd import _asmcheck, _build
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "halo_score.s")
        cmd = [_build._hipcc()] + [f for f in _build.FLAGS if f != "-fPIC"] + _build.EXTRA_FLAGS.get("halo_score.hip", []) + \
              ["--cuda-device-only", "-S", "-o", out, os.path.join(_build.CSRC, "halo_score.hip")]
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        checked, flagged = _asmcheck.check_file(out)
    assert checked >= 10 and flagged == 0
