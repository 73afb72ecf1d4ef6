"""Static checks on the device assembly hipcc produces for the hand-scheduled kernels (CPU only: hipcc cross-compiles)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_instruction_touches_a_register_whose_lds_read_is_in_flight():
    """k_feat_reduce_lr_dmaf issues the LDS reads of channel c + 1 by inline asm, interpolates channel c, and only then waits
    (counted s_waitcnt).  The compiler knows nothing about the reads being asynchronous: any copy, spill or use of a destination
    register it places between the issue and the wait reads stale data -- a race no parity test is guaranteed to see.  Round 4
    shipped one for a day (phi copies of the first channel's registers where the full-chunk / partial-chunk branch met the
    issue).  tools/check_inflight_regs.py scans the assembly of every low-res kernel for the pattern."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight_regs.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "flagged: 0" in r.stdout and "k_feat_reduce_lr_dmaf" in r.stdout
