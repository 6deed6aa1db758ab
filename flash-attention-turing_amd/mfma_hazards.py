"""Static scan for the one hazard class hipcc cannot see in these kernels: the accumulating MFMAs are inline asm ("+v" / "+a" tiles, fa_device.hpp), so the
compiler's hazard recogniser does not know their results land passes later, and it is free to schedule a register-only VALU instruction that reads such a
result directly behind the MFMA - above the `asm volatile("s_nop ...")` pad, whose "memory" clobber orders memory operations only.  Round 5 met exactly that
in an experiment build (three query columns per wave, profiles/r5_fwd_qb3_ab.log: v_fma_f32 two instructions behind the MFMA that produced its input, values
wrong by ~0.1); since then every pad names the registers it protects, and this scan walks the ISA of every instance.

The scan is conservative: every instruction counts one wait state (s_nop N: N + 1), labels are fall-through, branches keep the pending set - and since round 6
 a BACKWARD branch carries the pending set to its target: the code from the loop's top label is walked a second time with the MFMAs still in flight at the
loop's tail, so a reader at a loop top behind an MFMA at the loop tail is seen.  A result register touched (read or written) by a non-MFMA instruction, or read as
A / B by another MFMA, fewer than NEED wait states behind the MFMA that writes it, is reported.

NEED: the documented requirement + 1.  "XDL write VGPR -> VALU read / write of that VGPR" needs passes + 3 wait states on the gfx940 family that gfx950 extends
(LLVM GCNHazardRecognizer, GFX940_XDL_N_PassWriteVgprVALUReadWaitStates: 2-pass 5, 4-pass 7, 8-pass 11, 16-pass 19; the CDNA guide's section 5.7 states the 8-pass
case as "12 states = s_nop 11"): v_mfma_f32_16x16x32_{f16,bf16} is 4 passes (16 cycles, tools/ubench: 16.3 cycles back to back) -> 7, v_mfma_f32_32x32x16 is
8 passes -> 11.  The same numbers are what hipcc itself leaves behind its own (builtin) MFMAs in these files: a dependent VALU sits 8 wait states behind a
v_mfma_f32_16x16x32 (5 instructions + s_nop 2 in fa_bwd_dq16) and 12 behind a v_mfma_f32_32x32x16 (s_nop 11 in fa_bwd)."""
import re

NEED = {"16x16x32": 8, "32x32x16": 12}      # = documented 7 / 11 + 1; other shapes: 20 (none is issued from inline asm here)
HORIZON = 48                                # wait states after which an MFMA result has landed for every shape


def _regs(tok):
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        if m.group(1):
            out.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def _walk(lines, begin, end, pending, viol, labels, back_edges):
    """one straight-line pass over lines[begin:end]; `pending`: (dst regs, wait states elapsed, needed, line no, text) of the MFMAs still in flight.
    back_edges: None, or a list collecting (target line, pending set at the branch) of every backward branch met.  Stops early once nothing is pending
    and no back edges are being collected (the second pass of a loop top)."""
    for i in range(begin, end):
        l = lines[i].split(";")[0].strip()
        if not l or l.endswith(":") or (l.startswith(".") and not l.startswith(".LBB")):
            continue
        op = l.split()[0]
        ws = 1
        if op == "s_nop":
            ws = int(l.split()[1]) + 1
        elif op.startswith("v_mfma") or op.startswith("v_smfmac"):
            need_here = next((n for s, n in NEED.items() if s in op), 20)
            ops = l[len(op):].split(",")
            dst, srcab = _regs(ops[0]), _regs(ops[1]) | _regs(ops[2])
            for d, e, need, ln, t in pending:
                if d & srcab and e < need:
                    viol.append((i + 1, l, ln + 1, t, e, need))
            pending = [(d, e + 1, need, ln, t) for d, e, need, ln, t in pending if e + 1 < HORIZON]
            pending.append((dst, 0, need_here, i, l))
            continue
        elif op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pending = [(d, e + 1, need, ln, t) for d, e, need, ln, t in pending]
            if back_edges is not None and op != "s_endpgm" and pending:
                tgt = labels.get(l.split()[-1])
                if tgt is not None and tgt <= i:
                    back_edges.append((tgt, list(pending)))
            continue
        if not op.startswith("s_"):
            r = _regs(l[len(op):])
            for d, e, need, ln, t in pending:
                if d & r and e < need:
                    viol.append((i + 1, l, ln + 1, t, e, need))
        pending = [(d, e + ws, need, ln, t) for d, e, need, ln, t in pending if e + ws < HORIZON]
        if back_edges is None and not pending:
            return


def scan_kernel(lines, start, end):
    labels = {}
    for i in range(start + 1, end):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            labels[m.group(1)] = i
    viol, back = [], []
    _walk(lines, start + 1, end, [], viol, labels, back)
    seen = set(viol)
    for tgt, pend in back:                      # the loop's top again, behind the MFMAs in flight at its tail
        v2 = []
        _walk(lines, tgt, end, pend, v2, labels, None)
        for x in v2:
            if x not in seen:
                seen.add(x)
                viol.append(x)
    return viol


def scan_file(path):
    """{kernel name: [violations]} for every kernel of a --cuda-device-only -S listing"""
    lines = open(path).read().split("\n")
    out = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", l)
        if not m:
            continue
        end = next((j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end")), len(lines))
        out[m.group(1)] = scan_kernel(lines, i, end)
    return out


if __name__ == "__main__":
    import sys

    total = 0
    for k, v in scan_file(sys.argv[1]).items():
        print(k, len(v))
        for x in v[:12]:
            print("   line %d: %s   <- line %d: %s   (%d of %d wait states)" % x)
        total += len(v)
    print("violations:", total)
