"""Static check of the per-wave decode attention kernels (k_attn_decode_wave, k_attn_decode_wave_long) in the compiler's
assembly: their Q / K loads are issued from inline asm and retired by COUNTED waits inside the asm statement that consumes
them, so hipcc does not know the loads are in flight -- a copy, a spill or a reuse of a destination register that it schedules
between a load and the wait that retires it would read or clobber garbage (seen in round 3 with tied operands; the kernels are
written straight-line for that reason).  This walker proves the property for the code that was actually generated: it walks
the control-flow graph of each kernel with the FIFO of in-flight vector-memory operations as state (loads with their
destination registers, LDS-DMA pieces and stores with none; `s_waitcnt vmcnt(N)` retires all but the N youngest) and reports
every instruction that touches a register whose load has not been retired.
usage: hipcc ... --save-temps -c ze_attn_batch.hip; python tools/check_attn_asm.py ze_attn_batch-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re
import sys


def vregs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse_blocks(body):
    """-> {label: [instruction text, ...]}, label order.  An instruction that hipcc emitted itself (outside ;APP ... ;NO_APP) carries
    the prefix "\x01": its destination registers are the compiler's own business (it waits before it uses them) -- the walker counts
    such an operation in the FIFO without registers, which keeps the state space of compiler-generated loops (the merges of the tail,
    two of them since round 6) from multiplying."""
    blocks, order, cur = {"entry": []}, ["entry"], "entry"
    in_app = False
    for raw in body.split("\n"):
        l = raw.strip()
        if l.startswith(";APP"):
            in_app = True
        elif l.startswith(";NO_APP"):
            in_app = False
        if not l or l.startswith((";", "//")):
            continue
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        if l.startswith("."):
            continue
        blocks[cur].append(("" if in_app else "\x01") + l.split(";")[0].strip())
    return blocks, order


VMEM = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)")


def check(body, name, limit=400000):
    blocks, order = parse_blocks(body)
    nxt = {b: (order[i + 1] if i + 1 < len(order) else None) for i, b in enumerate(order)}
    seen, work, bad, n_loads, n_waits, steps = set(), [("entry", ())], {}, 0, 0, 0
    while work:
        b, pend = work.pop()
        if (b, pend) in seen:
            continue
        seen.add((b, pend))
        steps += 1
        if steps > limit:
            print(f"  {name}: state limit reached")
            break
        pending = [set(p) for p in pend]
        ins = blocks[b]
        succ = [nxt[b]]
        stop = False
        for k, l in enumerate(ins):
            own = l.startswith("\x01")   # emitted by the compiler, not by an asm statement
            if own:
                l = l[1:]
            if VMEM.match(l):
                op = l.split()[0]
                is_load = "_load_" in op and "_lds_" not in op
                dest = vregs(l.split(",")[0]) if (is_load and not own) else set()
                live = set().union(*pending) if pending else set()
                src = vregs(",".join(l.split(",")[1:])) if is_load else vregs(l)
                if live & (src | dest):
                    bad[(b, l)] = sorted(live & (src | dest))[:8]
                pending.append(dest)
                if len(pending) > 63:   # (vmcnt is a six-bit counter: nothing older can still be counted)
                    pending.pop(0)
                n_loads += 1
                continue
            m = re.search(r"vmcnt\((\d+)\)", l) if l.startswith("s_waitcnt") else None
            if m:
                n = int(m.group(1))
                n_waits += 1
                while len(pending) > n:
                    pending.pop(0)
                continue
            if l.startswith("s_endpgm"):
                stop = True
                break
            m = re.match(r"s_branch (\.LBB\w+)", l)
            if m:
                succ = [m.group(1)]
                break
            m = re.match(r"s_cbranch_\w+ (\.LBB\w+)", l)
            if m:
                succ = [m.group(1), nxt[b]] if k == len(ins) - 1 else succ + [m.group(1)]
                continue
            live = set().union(*pending) if pending else set()
            if live and vregs(l) & live:
                bad[(b, l)] = sorted(vregs(l) & live)[:8]
        if stop:
            continue
        for s_ in succ:
            if s_:
                work.append((s_, tuple(frozenset(p) for p in pending)))
    for (b, l), r in list(bad.items())[:6]:
        print(f"  {name}: block {b}: `{l[:90]}` touches in-flight registers {r}")
    return n_loads, n_waits, len(bad)


def main(path):
    s = open(path).read()
    total_bad = 0
    found = 0
    for m in re.finditer(r"^(_Z\d+k_attn_decode_wave\w*):", s, re.M):
        a = m.start()
        b = s.index(".Lfunc_end", a)
        n, w, bad = check(s[a:b], m.group(1))
        found += 1
        print(f"{m.group(1)}: {n} vector-memory operations and {w} waits walked, {bad} early touches")
        total_bad += bad
    if not found:
        print("no k_attn_decode_wave kernel in", path)
        return 2
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
