"""Static check of k_gemm_wstream's weight path in the compiler's assembly (hipcc -S): between the inline-asm buffer load of
a weight piece and the counted s_waitcnt that retires it, no instruction on any path from the load may touch the load's
destination registers (hipcc does not know the load is in flight: a copy or a reuse it schedules early would read or
clobber garbage).  Walks the control-flow graph from every weight load with the FIFO of in-flight destinations as state.
usage: hipcc ... --cuda-device-only -S ze_gemm.hip -o ze_gemm.s; python tools/check_wstream_asm.py ze_gemm.s"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse_blocks(body):
    blocks, order, cur = {}, [], "entry"
    blocks[cur] = []
    order.append(cur)
    for raw in body.split("\n"):
        l = raw.strip()
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
        blocks[cur].append(l.split(";")[0].strip())
    return blocks, order


def check(body, name):
    blocks, order = parse_blocks(body)
    nxt = {b: (order[i + 1] if i + 1 < len(order) else None) for i, b in enumerate(order)}
    start = next((b for b in order if any(i.startswith("buffer_load_dwordx4") and " nt" in i for i in blocks[b])), None)
    if start is None:
        return 0, 0, 0
    seen, work, bad, n_loads, n_waits = set(), [(start, ())], [], 0, 0
    while work:
        b, pend = work.pop()
        if (b, pend) in seen:
            continue
        seen.add((b, pend))
        pending = [set(p) for p in pend]
        ins = blocks[b]
        succ = [nxt[b]]
        stop = False
        for k, l in enumerate(ins):
            if l.startswith("buffer_load_dwordx4") and " nt" in l:
                pending.append(regs(l.split(",")[0]))
                n_loads += 1
                continue
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", l)
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
            if live and regs(l) & live:
                bad.append((b, l, sorted(regs(l) & live)[:8]))
        if stop or not pending:
            continue  # nothing in flight: the rest of the program is the compiler's own business
        for s_ in succ:
            if s_:
                work.append((s_, tuple(frozenset(p) for p in pending)))
    for b, l, r in bad[:5]:
        print(f"  {name}: block {b}: `{l}` touches in-flight registers {r}")
    return n_loads, n_waits, len(bad)


def main(path):
    s = open(path).read()
    total_bad = 0
    for m in re.finditer(r"^(_Z14k_gemm_wstream\w+):", s, re.M):
        a = m.start()
        b = s.index(".Lfunc_end", a)
        n, w, bad = check(s[a:b], m.group(1))
        print(f"{m.group(1)}: {n} weight loads and {w} waits walked, {bad} early touches")
        total_bad += bad
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
