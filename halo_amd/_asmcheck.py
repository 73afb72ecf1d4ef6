"""Static check of the device assembly hipcc produced for a translation unit that contains hand-scheduled inline assembly.

Two contracts the C++ source cannot express are verified on the instructions that were actually emitted; `halo_amd._build`
runs this on the very compile that produces libhalo_hip.so and refuses to link when anything is flagged.

1. **Registers with an LDS read in flight.**  A hand-issued `ds_read*` (inside `;;#ASMSTART ... ;;#ASMEND`) returns its data some
   cycles later; the matching `s_waitcnt lgkmcnt(n)` sits in a LATER asm statement.  For the compiler the read's outputs exist when
   the first statement ends, so it may copy, spill or overwrite them before the wait (round 4 shipped exactly that: phi copies at a
   control-flow merge, 144-60 000 wrong pixels per image).  Rule: from a hand-issued LDS read until a wait retires it, NO instruction
   may name one of its destination registers -- on every path through the function.
2. **Counted waits on the LDS-DMA double buffer.**  `global_load_lds_dwordx4` blocks are issued per chunk and a hand-written
   `s_waitcnt vmcnt(N)`, N > 0, lets the newest block stay in flight.  That is only right if the N newest vector-memory operations
   outstanding at the wait ARE that block.  Rule: at a hand-written counted vmcnt wait, every possibly-outstanding vector-memory
   operation is a hand-issued LDS-DMA (no compiler load or store in between), at least N of them on every path.

Method: basic blocks and the control-flow graph are rebuilt from the labels and branches of each function, and a forward data-flow
analysis runs to a fixed point.  LGKM state = the ordered queue of outstanding LGKM operations (hand LDS reads with their destination
registers, other LDS operations, scalar-memory operations -- which return out of order, so a queue that holds one is only retired by
lgkmcnt(0)); VM state = the ordered queue of outstanding vector-memory operations (dma / other).  Where the queues of two
predecessors differ the merge is conservative: pending hand registers become "sticky" (only a wait for zero retires them) and the
VM queue becomes "unknown" (a counted hand wait reached in that state is flagged).  Waits the compiler inserts are honoured like
hand-written ones.

    python -m halo_amd._asmcheck file.s [...]         # exit code 1 if anything is flagged
"""
import re
import sys

_FUNC = re.compile(r"^([A-Za-z_][\w$.]*):\s*(?:;.*)?$")
_LABEL = re.compile(r"^(\.L[\w$.]+):")
_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_WAIT_LGKM = re.compile(r"lgkmcnt\((\d+)\)")
_WAIT_VM = re.compile(r"vmcnt\((\d+)\)")
_HAND_SOURCE = re.compile(r"ds_read|global_load_lds|buffer_load[^\n]*\blds\b")


def source_needs_check(text):
    """does this C++ / HIP source hand-issue LDS reads or LDS-DMA in inline assembly?"""
    for m in re.finditer(r"asm\s*(?:volatile)?\s*\(", text):
        if _HAND_SOURCE.search(text[m.end():m.end() + 1200].split(");")[0]):
            return True
    return False


def _vregs(operands):
    out = set()
    for m in _VREG.finditer(operands):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


class _Insn(object):
    __slots__ = ("op", "text", "hand", "regs", "dst")

    def __init__(self, text, hand):
        self.text = text
        self.op = text.split()[0]
        self.hand = hand
        ops = text[len(self.op):]
        self.regs = _vregs(ops)
        first = ops.split(",")[0]
        self.dst = _vregs(first)


def _functions(text):
    """yield (name, [lines]) for every function body: from its label to .Lfunc_end"""
    lines = text.splitlines()
    i, n = 0, len(lines)
    while i < n:
        m = _FUNC.match(lines[i])
        if m and not lines[i].startswith(".L") and i + 1 < n:
            name = m.group(1)
            j = i + 1
            body = []
            while j < n and not lines[j].lstrip().startswith(".Lfunc_end"):
                body.append(lines[j])
                j += 1
            if j < n:
                yield name, body
                i = j
        i += 1


def _blocks(body):
    """-> (blocks: list of (label or None, [Insn]), succ: list of lists of block indices)"""
    blocks = [[None, []]]
    hand = False
    for raw in body:
        t = raw.strip()
        if not t:
            continue
        if t.startswith(";;#ASMSTART"):
            hand = True
            continue
        if t.startswith(";;#ASMEND"):
            hand = False
            continue
        m = _LABEL.match(t)
        if m:
            blocks.append([m.group(1), []])
            continue
        if t[0] in ";." or t.startswith("//"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        blocks[-1][1].append(_Insn(t, hand))
        if t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            blocks.append([None, []])
    index = {lab: k for k, (lab, _) in enumerate(blocks) if lab}
    succ = []
    for k, (_, insns) in enumerate(blocks):
        s = []
        last = insns[-1] if insns else None
        fall = True
        if last is not None:
            if last.op == "s_branch":
                fall = False
            if last.op in ("s_endpgm",) or last.op.startswith("s_setpc"):
                fall = False
            if last.op.startswith(("s_cbranch", "s_branch")):
                tgt = last.text.split()[-1]
                if tgt in index:
                    s.append(index[tgt])
        if fall and k + 1 < len(blocks):
            s.append(k + 1)
        succ.append(s)
    return blocks, succ


_LGKM_OPS = ("ds_", "s_load", "s_buffer_load", "s_scratch_load", "s_memtime", "s_memrealtime", "s_sendmsg", "s_atomic", "s_dcache", "s_store", "s_buffer_store")
_VM_OPS = ("global_", "buffer_", "flat_", "scratch_", "tbuffer_", "image_")


def _is_lgkm(op):
    return op.startswith(_LGKM_OPS)


def _is_vm(op):
    return op.startswith(_VM_OPS) and not op.startswith(("buffer_wbl2", "buffer_inv", "buffer_gl", "global_wb", "global_inv"))


_TOP = ("TOP",)            # "anything may be outstanding": reached only when a bound of the analysis is exceeded
_MAX_QUEUE, _MAX_STATES = 96, 512


def _step_lgkm(q, ins, report):
    """one instruction on one possible LGKM queue.  Entries, oldest first: ('h', regs) a hand-issued LDS read and its destination
    registers, 'o' any other LDS operation.  Scalar-memory operations are ignored: they return out of order, and the worst case for
    a counted wait is that they are all back already -- which is the queue without them.  Entries older than the oldest 'h' are
    dropped (LDS operations return in order, so they never outlive it)."""
    if q is _TOP:
        if ins.hand and ins.op.startswith("ds_read"):
            report("hand-issued `%s` in a state the analysis could not bound" % ins.text)
        return q
    op = ins.op
    if op.startswith("s_waitcnt"):
        m = _WAIT_LGKM.search(ins.text)
        n = int(m.group(1)) if m else None
        numeric = re.match(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", ins.text)
        if numeric:                                  # raw immediate, gfx9 layout: vmcnt 3:0 + 15:14, expcnt 6:4, lgkmcnt 11:8
            n = (int(numeric.group(1), 0) >> 8) & 0xf
            n = None if n == 0xf else n
        if n is None:
            return q
        q = q[max(0, len(q) - n):] if n else ()
    else:
        pend = set()
        for e in q:
            if e != "o":
                pend |= e[1]
        if pend and ins.regs & pend:
            report("`%s` names v%s while a hand-issued LDS read into it is in flight" % (ins.text, sorted(ins.regs & pend)))
        if op.startswith("ds_"):
            q = q + ((("h", frozenset(ins.dst)),) if (ins.hand and op.startswith("ds_read")) else ("o",))
    k = 0
    while k < len(q) and q[k] == "o":                # nothing older than the oldest hand read matters
        k += 1
    q = q[k:]
    return q if len(q) <= _MAX_QUEUE else _TOP


def _step_vm(q, ins, report):
    """one instruction on one possible vector-memory queue.  Entries, oldest first: 'd' a hand-issued LDS-DMA of the block being
    issued, 'D' one of an earlier block (a counted hand wait has passed since), 'o' any other vector-memory operation.  A
    hand-written `s_waitcnt vmcnt(N)`, N > 0, must find nothing but LDS-DMA outstanding (loads return in order among themselves; a
    compiler load or store in between would shift the count or return out of order) and at least N fresh 'd' at the young end (a
    block was issued since the previous counted wait): then "all but the newest N have landed" means "every earlier block has"."""
    op = ins.op
    if op.startswith("s_waitcnt"):
        m = _WAIT_VM.search(ins.text)
        n = int(m.group(1)) if m else None
        numeric = re.match(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", ins.text)
        if numeric:
            imm = int(numeric.group(1), 0)
            n = (imm & 0xf) | (((imm >> 14) & 0x3) << 4)
            n = None if n == 0x3f else n
        if n is None:
            return q
        if n == 0:
            return ()
        if ins.hand:
            if q is _TOP:
                report("counted wait `%s` in a state the analysis could not bound" % ins.text)
                return q
            fresh = 0
            while fresh < len(q) and q[len(q) - 1 - fresh] == "d":
                fresh += 1
            if fresh < n or any(e not in ("d", "D") for e in q):
                report("counted wait `%s`: outstanding vector-memory operations are [%s] (d = LDS-DMA issued since the last counted "
                       "wait, D = before it, o = anything else); expected only LDS-DMA, the newest %d of them fresh" % (ins.text, "".join(q), n))
            return tuple("D" for _ in q[max(0, len(q) - n):])
        return q if q is _TOP else q[max(0, len(q) - n):]
    if q is _TOP:
        return q
    if _is_vm(op):
        q = q + ("d" if (ins.hand and "lds" in op) else "o",)
    return q if len(q) <= _MAX_QUEUE else _TOP


def _run(blocks, succ, step):
    """forward data flow over SETS of possible queues (path-insensitive merges lose the correlations compilers create between a
    flag register and the state: a loop header reached with and without a pending block, then a branch on the flag)."""
    entry = [None] * len(blocks)
    entry[0] = frozenset([()])
    work = [0]
    budget = 400000
    while work:
        k = work.pop()
        budget -= 1
        if budget < 0:
            return None
        cur = entry[k]
        for ins in blocks[k][1]:
            cur = frozenset(step(q, ins, lambda m_: None) for q in cur)
        if len(cur) > _MAX_STATES:
            cur = frozenset([_TOP])
        for s in succ[k]:
            merged = cur if entry[s] is None else (entry[s] | cur)
            if len(merged) > _MAX_STATES:
                merged = frozenset([_TOP])
            if entry[s] is None or merged != entry[s]:
                entry[s] = merged
                work.append(s)
    return entry


def check_function(name, body):
    """-> list of messages, or None when the function holds no hand-scheduled LDS read / LDS-DMA / wait"""
    blocks, succ = _blocks(body)
    if not any(i.hand and (i.op.startswith("ds_read") or "lds" in i.op) for _, insns in blocks for i in insns):
        return None
    msgs = []
    for step in (_step_lgkm, _step_vm):
        entry = _run(blocks, succ, step)
        if entry is None:
            msgs.append("data-flow analysis did not converge")
            continue
        seen = set()
        for k, (_, insns) in enumerate(blocks):
            if entry[k] is None:
                continue
            cur = entry[k]
            for ins in insns:
                nxt = set()
                for q in cur:
                    nxt.add(step(q, ins, lambda m_: (m_ not in seen) and (seen.add(m_) or msgs.append(m_))))
                cur = nxt
    return msgs


def check_text(text, verbose=False, out=None):
    """-> (functions checked, total flagged); prints one line per checked function when verbose"""
    checked = flagged = 0
    for name, body in _functions(text):
        msgs = check_function(name, body)
        if msgs is None:
            continue
        checked += 1
        flagged += len(msgs)
        if verbose or msgs:
            print("%-120s %d" % (name, len(msgs)), file=out or sys.stdout)
            for m_ in msgs[:4]:
                print("    " + m_, file=out or sys.stdout)
    return checked, flagged


def check_file(path, verbose=False):
    with open(path) as f:
        return check_text(f.read(), verbose)


if __name__ == "__main__":
    total = 0
    for p in sys.argv[1:]:
        c, n = check_file(p, verbose=True)
        print("%s: %d hand-scheduled function(s), %d flagged" % (p, c, n))
        total += n
    sys.exit(1 if total else 0)
