"""The overlap probe of tools/probe_overlap.py inside ONE process: two engines (own weights, own workspaces) on two HIP
streams of the same GPU, driven by two Python threads (ctypes releases the GIL during a call).  One loops 64-chain decode
bursts, the other 16-chain prefill passes; each is timed alone, then both together.
usage: python tools/probe_overlap_inproc.py [seconds=5]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from zoomearth_amd.config import ModelConfig  # noqa: E402
from zoomearth_amd.engine import Engine  # noqa: E402
from zoomearth_amd.synth import uniform_ints  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0


def build(B):
    e = Engine(ModelConfig.zoomearth_3b(), max_seqs=B, max_ctx=2048, max_patches=2048, max_tile_side=1024,
               max_prefill_rows=16 * 832)
    e.fill_synthetic(0)
    ids = [uniform_ints(100 + s, 802, 1000, 150000).tolist() for s in range(B)]
    pl = [e.rope_index(i, []) for i in ids]

    def prefill(gs):
        for s in gs:
            e.seq_reset(s)
        e.prefill_batch(gs, [ids[s] for s in gs], [None] * len(gs), [pl[s][0] for s in gs], [pl[s][1] for s in gs])
    return e, prefill


dec, dec_prefill = build(64)
pre, pre_prefill = build(16)
for g0 in range(0, 64, 16):
    dec_prefill(list(range(g0, g0 + 16)))
p = dec.gen_params(ignore_eos=True)
for s in range(64):
    dec.chain_begin(s, p)
dec.decode_burst(list(range(64)), 4, p)
pre_prefill(list(range(16)))
torch.cuda.synchronize()
streams = {"decode": torch.cuda.Stream(), "prefill": torch.cuda.Stream()}
result = {}


def run(role, start, stop_at):
    with torch.cuda.stream(streams[role]):
        st = torch.cuda.current_stream()
        if role == "decode":  # the graph of the step is captured per stream: warm up on this one
            for s in range(64):
                dec.seq_truncate(s, 802)
            dec.decode_burst(list(range(64)), 8, p)
        else:
            pre_prefill(list(range(16)))
        st.synchronize()
        start.wait()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() < stop_at[0]:
            if role == "decode":
                for s in range(64):
                    dec.seq_truncate(s, 802)
                dec.decode_burst(list(range(64)), 16, p)
                n += 16
            else:
                pre_prefill(list(range(16)))
                n += 1
            st.synchronize()
        dt = time.perf_counter() - t0
        result[role] = (n / dt, 1000 * dt / max(n, 1))


for roles in (["decode"], ["prefill"], ["decode", "prefill"]):
    result.clear()
    start = threading.Barrier(len(roles) + 1)
    stop_at = [0.0]
    ts = [threading.Thread(target=run, args=(r, start, stop_at)) for r in roles]
    for t in ts:
        t.start()
    stop_at[0] = time.perf_counter() + SECS + 1.0
    start.wait()
    stop_at[0] = time.perf_counter() + SECS
    for t in ts:
        t.join()
    print("--- " + " + ".join(roles) + ": " + ", ".join(
        f"{r} {result[r][0]:.2f} {'steps' if r == 'decode' else 'passes'}/s ({result[r][1]:.2f} ms each)" for r in roles), flush=True)
dec.close()
pre.close()
