"""Does a memory-bound decode stream overlap with a compute-bound prefill stream on one MI355X?  Two PROCESSES on the
same GPU (separate HW queues, like two HIP streams): one loops 64-chain decode bursts, the other 16-chain prefill passes;
each is timed alone and both together.  usage: python tools/probe_overlap.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(role, secs, gate):
    import torch
    from zoomearth_amd.config import ModelConfig
    from zoomearth_amd.engine import Engine
    from zoomearth_amd.synth import uniform_ints
    B = 64 if role == "decode" else 16
    e = Engine(ModelConfig.zoomearth_3b(), max_seqs=B, max_ctx=2048, max_patches=2048, max_tile_side=1024, max_prefill_rows=16 * 832)
    e.fill_synthetic(0)
    ids = [uniform_ints(100 + s, 802, 1000, 150000).tolist() for s in range(B)]
    pl = [e.rope_index(i, []) for i in ids]

    def prefill(gs):
        for s in gs:
            e.seq_reset(s)
        e.prefill_batch(gs, [ids[s] for s in gs], [None] * len(gs), [pl[s][0] for s in gs], [pl[s][1] for s in gs])

    if role == "decode":
        for g0 in range(0, B, 16):
            prefill(list(range(g0, g0 + 16)))
        p = e.gen_params(ignore_eos=True)
        for s in range(B):
            e.chain_begin(s, p)
        e.decode_burst(list(range(B)), 4, p)
    else:
        prefill(list(range(16)))
    torch.cuda.synchronize()
    open(gate + "." + role, "w").close()
    while not os.path.exists(gate):
        time.sleep(0.005)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < secs:
        if role == "decode":
            for s in range(B):
                e.seq_truncate(s, 802)
            e.decode_burst(list(range(B)), 16, p)
            n += 16
        else:
            prefill(list(range(16)))
            torch.cuda.synchronize()
            n += 1
    dt = time.perf_counter() - t0
    print(f"{role}: {n / dt:.2f} {'steps' if role == 'decode' else 'passes'}/s ({1000 * dt / n:.2f} ms each)", flush=True)
    e.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(sys.argv[1], float(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    for roles in (["decode"], ["prefill"], ["decode", "prefill"]):
        gate = f"/tmp/ze_gate_{os.getpid()}_{'_'.join(roles)}"
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), r, "6", gate]) for r in roles]
        while not all(os.path.exists(gate + "." + r) for r in roles):
            time.sleep(0.05)
            if any(p.poll() is not None for p in ps):
                break
        open(gate, "w").close()
        print("---", "+".join(roles), flush=True)
        for p in ps:
            p.wait()
