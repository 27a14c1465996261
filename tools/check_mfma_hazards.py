"""Static check of the MFMA data hazards that inline asm hides from hipcc, on the compiler's assembly of ANY unit of the library.

Root cause of round 4's mis-scheduled decode-attention instantiations ("asm rule"): on gfx950 an MFMA that reads a
VGPR written by a (non-MFMA) VALU instruction needs TWO wait states between the two.  hipcc's hazard recogniser inserts them
(`s_nop`) for instructions it can see -- it cannot see a VALU write inside an `asm` statement, and it cannot see that an `asm`
statement is an MFMA.  Two rules, checked for every kernel of every file given:

  A. VALU -> MFMA: between a VALU instruction that writes register r and an MFMA that reads r (SrcA / SrcB / SrcC) there are at
     least two wait states (any instruction counts one, `s_nop N` counts N + 1).  Holds by construction for compiler-emitted
     pairs, so any violation involves an asm statement; both members of the pair are printed.
  B. MFMA inside an asm statement: the statement itself pads its LAST MFMA with at least 11 wait states (16x16) / 19 (32x32)
     before it ends, so that no compiler-scheduled reader or writer of its result can be too early.
  C. VALU -> SGPR -> vector memory: a vector-memory instruction that reads an SGPR (base address, descriptor, soffset) which a
     VALU instruction wrote (v_readfirstlane / v_readlane / a compare) comes at least five wait states later (the loads and
     LDS-DMA pieces of the GEMM and attention kernels are issued from asm with scalar bases the compiler computes).

The walk is per basic block (a label starts a new one with empty history: a hazard across a taken branch costs at least the
branch's own issue cycles; fall-through edges are covered conservatively by keeping the history across a label that the
previous instruction falls into).
usage: hipcc ... --cuda-device-only -S x.hip -o x.s; python tools/check_mfma_hazards.py x.s [y.s ...]   (exit 1 on a violation)"""
import re
import sys

VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")
VMEM = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)")


def sregs(tok):
    out = set()
    for m in SREG.finditer(tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    if re.search(r"\bvcc\b", tok):
        out.add(-1)
    return out


def vregs(tok):
    out = set()
    for m in VREG.finditer(tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def split_ops(l):
    parts = l.split(None, 1)
    return parts[0], ([o.strip() for o in parts[1].split(",")] if len(parts) > 1 else [])


# VALU instructions whose first operand is NOT a VGPR destination
NO_VDST = ("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop")


def check_kernel(name, lines):
    """lines: (text, in_asm) of one kernel.  Returns (n_mfma, violations)."""
    bad, n_mfma = [], 0
    hist = []  # (wait states this entry is worth, written vregs if VALU non-MFMA, text, in_asm)
    shist = []  # (wait states, SGPRs written by a VALU instruction, text)
    asm_tail = None  # wait states behind the last MFMA inside the current asm statement
    prev_in_asm = False
    for text, in_asm in lines:
        if prev_in_asm and not in_asm and asm_tail is not None:
            need, got, mf = asm_tail
            if got < need:
                bad.append(f"B: asm statement ends {got} wait states behind `{mf}` (needs {need})")
            asm_tail = None
        prev_in_asm = in_asm
        if text.endswith(":"):
            if hist and hist[-1][2].split()[0] in ("s_branch", "s_endpgm", "s_setpc_b64"):
                hist = []  # not reached by fall-through
            continue
        op, ops = split_ops(text)
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            n_mfma += 1
            srcs = vregs(",".join(ops[1:4]))
            ws = 0
            for w, wr, t, a in reversed(hist):
                if ws >= 2:
                    break
                if wr & srcs:
                    bad.append(f"A: `{t}`{' [asm]' if a else ''} -> `{text}`{' [asm]' if in_asm else ''}: {ws} wait states, registers "
                               f"{sorted(wr & srcs)[:4]}")
                ws += w
            hist.append((1, set(), text, in_asm))
            shist.append((1, set(), text))
            if in_asm:
                asm_tail = (19 if "32x32" in op else 11, 0, text)
            continue
        w = 1
        if op == "s_nop":
            w = int(ops[0], 0) + 1
        if VMEM.match(op):
            use = sregs(",".join(ops))
            ws = 0
            for w2, swr, t in reversed(shist):
                if ws >= 5:
                    break
                if swr & use:
                    bad.append(f"C: `{t}` -> `{text}`{' [asm]' if in_asm else ''}: {ws} wait states, SGPRs {sorted(swr & use)[:4]}")
                ws += w2
        swr = set()
        if op.startswith("v_") and ops and (op.startswith(("v_readfirstlane", "v_readlane", "v_cmp")) or "vcc" in ops[0] or ops[0].startswith("s")):
            swr = sregs(ops[0]) if not op.startswith("v_cmp") or ops[0].startswith(("s", "vcc")) else {-1}
        shist.append((w, swr, text))
        if len(shist) > 10:
            shist.pop(0)
        wr = set()
        if op.startswith("v_") and not op.startswith(NO_VDST) and ops:
            wr = vregs(ops[0])
        hist.append((w, wr, text, in_asm))
        if len(hist) > 8:
            hist.pop(0)
        if in_asm and asm_tail is not None:
            need, got, mf = asm_tail
            asm_tail = (need, got + w, mf)
    return n_mfma, bad


def kernels_of(path):
    s = open(path).read()
    out = []
    for m in re.finditer(r"^(_Z\w+):\s*(?:;.*)?$", s, re.M):
        end = s.find(".end_amdhsa_kernel", m.end())
        nxt = s.find("\n\t.section", m.end())
        stop = min(x for x in (end, nxt, len(s)) if x > 0)
        lines, in_asm = [], False
        for raw in s[m.end():stop].split("\n"):
            l = raw.strip()
            if l.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if l.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not l or l.startswith((";", "//", ".")) and not re.match(r"^\.LBB\w+:", l):
                continue
            t = l.split(";")[0].strip()
            if t:
                lines.append((t, in_asm))
        if lines:
            out.append((m.group(1), lines))
    return out


def main(paths):
    total_bad = 0
    for p in paths:
        n_k, n_m, n_b = 0, 0, 0
        for name, lines in kernels_of(p):
            n, bad = check_kernel(name, lines)
            n_k += 1
            n_m += n
            if bad:
                n_b += len(bad)
                print(f"{p}: {name}: {len(bad)} hazard(s)")
                for b in bad[:6]:
                    print("    " + b)
        print(f"{p}: {n_k} kernels, {n_m} MFMAs, {n_b} hazards")
        total_bad += n_b
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
