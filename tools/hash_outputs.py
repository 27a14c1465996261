"""Deterministic workload over every kernel family, printing a hash per output: used to A/B two builds of the library
(ZE_LIB_PATH) bit for bit -- e.g. with and without hipcc's SLP vectoriser."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine
from zoomearth_amd.synth import uniform_ints, synthetic_tile

def h(name, t):
    a = t.detach().cpu().contiguous().view(torch.uint8).numpy() if isinstance(t, torch.Tensor) else np.ascontiguousarray(t)
    print(f"{name:28s} {hashlib.sha256(a.tobytes()).hexdigest()[:16]}", flush=True)

import dataclasses
for label, cfg in (("tiny", ModelConfig.tiny()),
                   ("3b-shape", dataclasses.replace(ModelConfig.zoomearth_3b(), text=dataclasses.replace(ModelConfig.zoomearth_3b().text, num_hidden_layers=2),
                                                    vision=dataclasses.replace(ModelConfig.zoomearth_3b().vision, depth=2, fullatt_block_indexes=(1,))))):
    e = Engine(cfg, device=0, max_seqs=3, max_ctx=2048, max_patches=4096, max_tile_side=2048, max_prefill_rows=4096)
    e.fill_synthetic(seed=1, std=0.02, matrix_gain=4.0, bias_std=0.02, norm_jitter=0.1)
    tile = torch.from_numpy(synthetic_tile(77, 900, 1100)).cuda()
    view = e.crop_resize(tile, (0, 0, 1100, 900), (512, 418))
    h(label + " view", view)
    pv, g = e.preprocess_image(view)
    h(label + " pixel_values", pv)
    emb = e.vit_forward(pv, [g])
    h(label + " vit", emb)
    n_img = g[1] * g[2] // 4
    vocab = cfg.text.vocab_size
    ids = uniform_ints(5, 20, 10, vocab - 100).tolist() + [cfg.vision_start_token_id] + [cfg.image_token_id] * n_img + [cfg.vision_end_token_id] + uniform_ints(6, 200, 10, vocab - 100).tolist()
    pos, delta = e.rope_index(ids, [g])
    e.seq_reset(0)
    lg = e.prefill(0, ids, emb, pos, delta)
    h(label + " prefill logits", lg)
    for i, t in enumerate(uniform_ints(7, 3, 10, vocab - 100).tolist()):
        h(label + f" decode {i}", e.decode_step(0, int(t)))
    e.seq_reset(0)
    e.prefill(0, ids, emb, pos, delta, want_logits=False)
    e.mark_seen(0, ids)
    print(label, "greedy", e.generate(0, 12, repetition_penalty=1.1, ignore_eos=True))
    e.seq_reset(0)
    e.prefill(0, ids, emb, pos, delta, want_logits=False)
    print(label, "sampled", e.generate(0, 12, ignore_eos=True, do_sample=True, temperature=0.9, seed=3))
    for s in (1, 2):
        e.seq_reset(s)
    for s, n in ((1, 150), (2, 97)):
        tid = uniform_ints(9 + s, n, 10, vocab - 100).tolist()
        p2, d2 = e.rope_index(tid, [])
        e.seq_reset(s)
        e.prefill(s, tid, None, p2, d2, want_logits=False)
    h(label + " decode_batch", e.decode_batch([1, 2], [11, 12]))
    e.close()
    torch.cuda.empty_cache()
