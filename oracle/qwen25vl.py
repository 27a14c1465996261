"""Oracle for the Qwen2.5-VL forward pass and greedy generation.

TEST INFRASTRUCTURE (see oracle/__init__.py).  numpy restatement of the arithmetic the
reference executes through `transformers` (pinned ==4.49.0 at /root/reference/requirements.txt:14;
restated from the 5.15.0 copy in this image, `HF:` = site-packages/transformers/).  Call sites in
the reference: src/eval/infer.py:102-115,147-157 and src/demo.py:7-19,127-128.

Two numeric modes:
  * "fp32": plain float32 everywhere (the yardstick).
  * "bf16": float32 storage, rounded to bfloat16 (round-to-nearest-even) at the points where the HF
    model running in torch.bfloat16 materialises a bf16 tensor.  GEMMs accumulate in float32 and
    round once.  This is what the HIP engine is designed to follow cast point for cast point.

Functions cite the HF lines they restate:
  RMSNorm                     HF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:64-79
  SwiGLU MLP                  :85-96 (vision, bias) / :541-554 (text, no bias)
  patch embed (conv3d=GEMM)   :99-122
  vision rotary table         :125-134, :441-446
  patch merger                :137-150
  rotate_half / vision rope   :153-171
  vision attention (varlen)   :211-291
  vision block / transformer  :294-321, :408-471
  text rotary (M-RoPE)        :486-538, :557-599
  text attention (GQA,causal) :602-689 ; eager reference :174-208
  decoder layer / text model  :692-757, :761-872
  embed + image scatter       :1185-1253
  lm_head, fp32 logits        :1386-1387 ; HF:generation/utils.py:2894
  repetition penalty          HF:generation/logits_process.py:373-413
  greedy loop / EOS / pad     HF:generation/utils.py:2783-2936
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from . import fp8, indices, prng


# ----------------------------------------------------------------------------- numerics helpers
def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest-even bfloat16 -> float32 (NaN preserved)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    with np.errstate(over="ignore"):
        y = ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)
    return np.where(np.isnan(x), x, y)


def _ident(x):
    return np.asarray(x, dtype=np.float32)


def _silu(x):
    return x / (1.0 + np.exp(-x, dtype=np.float32))


def _erf(x: np.ndarray) -> np.ndarray:
    # float64 erf (exact-GELU is only used on [N/4, 5120] in the merger)
    try:
        from scipy.special import erf
        return erf(x.astype(np.float64))
    except ImportError:  # pragma: no cover
        return np.vectorize(math.erf, otypes=[np.float64])(x.astype(np.float64))


def _gelu_erf(x):
    return (0.5 * x.astype(np.float64) * (1.0 + _erf(x / math.sqrt(2.0)))).astype(np.float32)


def _softmax(s):
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m, dtype=np.float32)
    return e / e.sum(axis=-1, keepdims=True, dtype=np.float32)


# ----------------------------------------------------------------------------- config
@dataclass
class VisionConfig:
    depth: int = 32
    hidden_size: int = 1280
    num_heads: int = 16
    intermediate_size: int = 3420
    out_hidden_size: int = 2048
    patch_size: int = 14
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    window_size: int = 112
    in_channels: int = 3
    fullatt_block_indexes: tuple = (7, 15, 23, 31)


@dataclass
class TextConfig:
    hidden_size: int = 2048
    num_hidden_layers: int = 36
    num_attention_heads: int = 16
    num_key_value_heads: int = 2
    intermediate_size: int = 11008
    vocab_size: int = 151936
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    mrope_section: tuple = (16, 24, 24)
    tie_word_embeddings: bool = True


@dataclass
class Config:
    vision: VisionConfig = field(default_factory=VisionConfig)
    text: TextConfig = field(default_factory=TextConfig)
    image_token_id: int = 151655
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653
    eos_token_ids: tuple = (151645, 151643)
    pad_token_id: int = 151643

    @property
    def head_dim(self):
        return self.text.hidden_size // self.text.num_attention_heads


def tiny_config() -> Config:
    """The parity-fixture config: real head dims (ViT 80, LLM 128), real GQA/M-RoPE structure,
    awkward MLP width (220 % 16 == 12 like 3420), small everything else."""
    return Config(
        vision=VisionConfig(depth=4, hidden_size=160, num_heads=2, intermediate_size=220, out_hidden_size=512,
                            fullatt_block_indexes=(1, 3)),
        text=TextConfig(hidden_size=512, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                        intermediate_size=1376, vocab_size=2048, tie_word_embeddings=True),
        image_token_id=2005, vision_start_token_id=2002, vision_end_token_id=2003,
        eos_token_ids=(2045, 2043), pad_token_id=2043,
    )


def heads_config() -> Config:
    """The 3B model's HEAD STRUCTURE at small depth (VERDICT r4 missing #4): text 16 query / 2 key-value heads x 128 (GQA group
    8, hidden 2048 -- `repeat_kv` at the real group size), ViT 16 heads x 80 (hidden 1280), two layers / two blocks (one of
    them full attention), small MLPs and vocabulary.  tests/golden/heads_chain.npz pins the oracle to transformers on it."""
    return Config(
        vision=VisionConfig(depth=2, hidden_size=1280, num_heads=16, intermediate_size=220, out_hidden_size=2048,
                            fullatt_block_indexes=(1,)),
        text=TextConfig(hidden_size=2048, num_hidden_layers=2, num_attention_heads=16, num_key_value_heads=2,
                        intermediate_size=1376, vocab_size=2048, tie_word_embeddings=True),
        image_token_id=2005, vision_start_token_id=2002, vision_end_token_id=2003,
        eos_token_ids=(2045, 2043), pad_token_id=2043,
    )


def weight_shapes(cfg: Config) -> dict:
    """HF 5.x checkpoint keys -> shapes."""
    v, t = cfg.vision, cfg.text
    s = {}
    s["model.visual.patch_embed.proj.weight"] = (v.hidden_size, v.in_channels, v.temporal_patch_size,
                                                 v.patch_size, v.patch_size)
    for i in range(v.depth):
        p = f"model.visual.blocks.{i}."
        s[p + "norm1.weight"] = (v.hidden_size,)
        s[p + "norm2.weight"] = (v.hidden_size,)
        s[p + "attn.qkv.weight"] = (3 * v.hidden_size, v.hidden_size)
        s[p + "attn.qkv.bias"] = (3 * v.hidden_size,)
        s[p + "attn.proj.weight"] = (v.hidden_size, v.hidden_size)
        s[p + "attn.proj.bias"] = (v.hidden_size,)
        for n in ("gate_proj", "up_proj"):
            s[p + f"mlp.{n}.weight"] = (v.intermediate_size, v.hidden_size)
            s[p + f"mlp.{n}.bias"] = (v.intermediate_size,)
        s[p + "mlp.down_proj.weight"] = (v.hidden_size, v.intermediate_size)
        s[p + "mlp.down_proj.bias"] = (v.hidden_size,)
    m = v.hidden_size * v.spatial_merge_size ** 2
    s["model.visual.merger.ln_q.weight"] = (v.hidden_size,)
    s["model.visual.merger.mlp.0.weight"] = (m, m)
    s["model.visual.merger.mlp.0.bias"] = (m,)
    s["model.visual.merger.mlp.2.weight"] = (v.out_hidden_size, m)
    s["model.visual.merger.mlp.2.bias"] = (v.out_hidden_size,)
    hd = cfg.head_dim
    s["model.language_model.embed_tokens.weight"] = (t.vocab_size, t.hidden_size)
    for i in range(t.num_hidden_layers):
        p = f"model.language_model.layers.{i}."
        s[p + "input_layernorm.weight"] = (t.hidden_size,)
        s[p + "post_attention_layernorm.weight"] = (t.hidden_size,)
        s[p + "self_attn.q_proj.weight"] = (t.num_attention_heads * hd, t.hidden_size)
        s[p + "self_attn.q_proj.bias"] = (t.num_attention_heads * hd,)
        s[p + "self_attn.k_proj.weight"] = (t.num_key_value_heads * hd, t.hidden_size)
        s[p + "self_attn.k_proj.bias"] = (t.num_key_value_heads * hd,)
        s[p + "self_attn.v_proj.weight"] = (t.num_key_value_heads * hd, t.hidden_size)
        s[p + "self_attn.v_proj.bias"] = (t.num_key_value_heads * hd,)
        s[p + "self_attn.o_proj.weight"] = (t.hidden_size, t.num_attention_heads * hd)
        s[p + "mlp.gate_proj.weight"] = (t.intermediate_size, t.hidden_size)
        s[p + "mlp.up_proj.weight"] = (t.intermediate_size, t.hidden_size)
        s[p + "mlp.down_proj.weight"] = (t.hidden_size, t.intermediate_size)
    s["model.language_model.norm.weight"] = (t.hidden_size,)
    if not t.tie_word_embeddings:
        s["lm_head.weight"] = (t.vocab_size, t.hidden_size)
    return s


def synthetic_weights(cfg: Config, seed: int = 0, std: float = 0.02, matrix_gain: float = 1.0,
                      bias_std: float = 0.0, norm_jitter: float = 0.0) -> dict:
    """Synthetic checkpoint from the repo PRNG (identical to the engine's ze_weights_fill_synthetic).

    matrices ~ N(0, (std*matrix_gain)^2) except embed_tokens / lm_head ~ N(0, std^2);
    norm weights = 1 + N(0, norm_jitter^2); biases ~ N(0, bias_std^2) (0 -> zeros).
    Values are rounded to bf16 (what the engine stores).  SURVEY.md 8 c.2: matrix_gain=4 gives
    diverse greedy output on random weights.
    """
    from . import prng

    out = {}
    for name, shape in weight_shapes(cfg).items():
        n = int(np.prod(shape))
        s = prng.tensor_seed(seed, name)
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("ln_q.weight") \
                or name.endswith("layernorm.weight") or name.endswith("language_model.norm.weight"):
            w = 1.0 + (prng.normal_ih4(s, n, norm_jitter) if norm_jitter > 0 else np.zeros(n, np.float32))
        elif name.endswith(".bias"):
            w = prng.normal_ih4(s, n, bias_std) if bias_std > 0 else np.zeros(n, np.float32)
        elif name.endswith("embed_tokens.weight") or name == "lm_head.weight":
            w = prng.normal_ih4(s, n, std)
        else:
            w = prng.normal_ih4(s, n, std * matrix_gain)
        out[name] = bf16_round(np.asarray(w, dtype=np.float32)).reshape(shape)
    return out


# ----------------------------------------------------------------------------- model
class Qwen25VLOracle:
    def __init__(self, cfg: Config, weights: dict, mode: str = "bf16", act_fp8: bool = False, share_weights: bool = False):
        """share_weights: `weights` are float32 arrays that already hold bf16-representable values (what the engine
        stores) and are used as they are, without the per-oracle copy -- the full-depth checks keep ONE 14-GB set for
        the fp32 and the bf16 oracle."""
        assert mode in ("bf16", "fp32")
        self.cfg = cfg
        self.mode = mode
        # act_fp8: the engine's ze_set_fp8_activations mode (include/zoomearth.h) -- the two RMSNorm outputs of every
        # decoder layer (the inputs of q/k/v and of gate/up) are replaced by their per-row E4M3 quantisation q * 2^k
        # (oracle/fp8.py).  No counterpart in the reference, whose checkpoints run bf16.
        self.act_fp8 = act_fp8
        self.r = bf16_round if mode == "bf16" else _ident
        if share_weights:
            assert all(v.dtype == np.float32 for v in weights.values())
            self.w = dict(weights)
        else:
            self.w = {k: (bf16_round(v.astype(np.float32)) if mode == "bf16" else v.astype(np.float32))
                      for k, v in weights.items()}
        if "lm_head.weight" not in self.w:
            self.w["lm_head.weight"] = self.w["model.language_model.embed_tokens.weight"]
        self.reset()

    # -------- primitives
    def linear(self, x, wname, bname=None):
        y = x @ self.w[wname].reshape(self.w[wname].shape[0], -1).T
        if bname is not None:
            y = y + self.w[bname]
        return self.r(y)

    def rmsnorm(self, x, wname, eps):
        # HF:...modeling_qwen2_5_vl.py:64-79: fp32 normalise, cast to input dtype, THEN multiply by weight
        xf = x.astype(np.float32)
        var = np.mean(xf * xf, axis=-1, keepdims=True, dtype=np.float32)
        xn = self.r(xf * (1.0 / np.sqrt(var + np.float32(eps), dtype=np.float32)))
        return self.r(self.w[wname] * xn)

    def swiglu(self, x, prefix, bias):
        b = (lambda n: prefix + n + ".bias") if bias else (lambda n: None)
        g = self.linear(x, prefix + "gate_proj.weight", b("gate_proj"))
        u = self.linear(x, prefix + "up_proj.weight", b("up_proj"))
        a = self.r(self.r(_silu(g)) * u)
        return self.linear(a, prefix + "down_proj.weight", b("down_proj"))

    # -------- vision tower
    def vision_rope_tables(self, grid_thw):
        v = self.cfg.vision
        hd = v.hidden_size // v.num_heads
        dim = hd // 2
        inv_freq = (1.0 / (np.float32(10000.0) ** (np.arange(0, dim, 2, dtype=np.float32) / np.float32(dim)))
                    ).astype(np.float32)
        pos = indices.vision_position_ids(grid_thw, v.spatial_merge_size)  # [N, 2]
        rot = (pos[:, :, None].astype(np.float32) * inv_freq[None, None, :]).reshape(pos.shape[0], -1)  # [N, dim]
        return rot.astype(np.float32)

    def vit_forward(self, pixel_values, grid_thw, return_blocks=False):
        """pixel_values float32 [N, C*T*P*P]; returns merged embeddings [N/4, out_hidden] in HF order."""
        v = self.cfg.vision
        unit = v.spatial_merge_size ** 2
        grid_thw = [tuple(int(a) for a in g) for g in grid_thw]
        cu_full = indices.vision_cu_seqlens(grid_thw)
        widx, cu_win = indices.vision_window_index(grid_thw, v.spatial_merge_size, v.window_size, v.patch_size)
        x = self.r(np.asarray(pixel_values, dtype=np.float32))
        h = self.linear(x, "model.visual.patch_embed.proj.weight")
        n = h.shape[0]
        h = h.reshape(n // unit, unit, -1)[widx].reshape(n, -1)
        rot = self.vision_rope_tables(grid_thw)
        rot = rot.reshape(n // unit, unit, -1)[widx].reshape(n, -1)
        emb = np.concatenate([rot, rot], axis=-1)
        cos, sin = np.cos(emb, dtype=np.float32), np.sin(emb, dtype=np.float32)
        nh = v.num_heads
        hd = v.hidden_size // nh
        scale = np.float32(hd ** -0.5)
        blocks = []
        for li in range(v.depth):
            p = f"model.visual.blocks.{li}."
            cu = cu_full if li in v.fullatt_block_indexes else cu_win
            y = self.rmsnorm(h, p + "norm1.weight", 1e-6)
            qkv = self.linear(y, p + "attn.qkv.weight", p + "attn.qkv.bias").reshape(n, 3, nh, hd)
            q, k, vv = qkv[:, 0], qkv[:, 1], qkv[:, 2]

            def rope(t):
                t1, t2 = t[..., : hd // 2], t[..., hd // 2:]
                rh = np.concatenate([-t2, t1], axis=-1)
                return self.r(t * cos[:, None, :] + rh * sin[:, None, :])

            q, k = rope(q), rope(k)
            o = np.empty_like(q)
            for s0, s1 in zip(cu[:-1], cu[1:]):
                qs = q[s0:s1].transpose(1, 0, 2)
                ks = k[s0:s1].transpose(1, 0, 2)
                vs = vv[s0:s1].transpose(1, 0, 2)
                pr = _softmax((qs @ ks.transpose(0, 2, 1)) * scale)
                o[s0:s1] = (pr @ vs).transpose(1, 0, 2)
            o = self.r(o).reshape(n, -1)
            h = self.r(h + self.linear(o, p + "attn.proj.weight", p + "attn.proj.bias"))
            y = self.rmsnorm(h, p + "norm2.weight", 1e-6)
            h = self.r(h + self.swiglu(y, p + "mlp.", bias=True))
            if return_blocks:
                blocks.append(h.copy())
        y = self.rmsnorm(h, "model.visual.merger.ln_q.weight", 1e-6).reshape(n // unit, -1)
        y = self.linear(y, "model.visual.merger.mlp.0.weight", "model.visual.merger.mlp.0.bias")
        y = self.r(_gelu_erf(y))
        y = self.linear(y, "model.visual.merger.mlp.2.weight", "model.visual.merger.mlp.2.bias")
        out = y[np.argsort(widx)]
        return (out, blocks) if return_blocks else out

    # -------- language model
    def reset(self):
        self.k_cache = [None] * self.cfg.text.num_hidden_layers
        self.v_cache = [None] * self.cfg.text.num_hidden_layers
        self.rope_delta = 0
        self.ctx = 0

    def text_rope(self, pos3):
        """pos3 int [3, T] -> (cos, sin) [T, head_dim] with M-RoPE section selection, cast to model dtype."""
        t = self.cfg.text
        hd = self.cfg.head_dim
        inv_freq = (1.0 / (np.float32(t.rope_theta) ** (np.arange(0, hd, 2, dtype=np.float32) / np.float32(hd)))
                    ).astype(np.float32)
        freqs = pos3[:, :, None].astype(np.float32) * inv_freq[None, None, :]  # [3, T, hd/2]
        emb = np.concatenate([freqs, freqs], axis=-1)  # [3, T, hd]
        cos, sin = np.cos(emb, dtype=np.float32), np.sin(emb, dtype=np.float32)
        sec = list(t.mrope_section) * 2
        sel_c, sel_s, o = [], [], 0
        for i, n in enumerate(sec):
            sel_c.append(cos[i % 3, :, o:o + n])
            sel_s.append(sin[i % 3, :, o:o + n])
            o += n
        return self.r(np.concatenate(sel_c, axis=-1)), self.r(np.concatenate(sel_s, axis=-1))

    def text_forward(self, h, pos3, return_layers=False):
        """h [T, hidden] new-token embeddings; pos3 [3, T]; appends to the KV cache; returns final-norm states."""
        t = self.cfg.text
        hd = self.cfg.head_dim
        nq, nkv = t.num_attention_heads, t.num_key_value_heads
        grp = nq // nkv
        cos, sin = self.text_rope(np.asarray(pos3))
        tn = h.shape[0]
        past = self.ctx
        scale = np.float32(hd ** -0.5)
        layers = []
        for li in range(t.num_hidden_layers):
            p = f"model.language_model.layers.{li}."
            y = self.rmsnorm(h, p + "input_layernorm.weight", t.rms_norm_eps)
            if self.act_fp8:
                y = fp8.quantize_rows(y)[2]
            q = self.linear(y, p + "self_attn.q_proj.weight", p + "self_attn.q_proj.bias").reshape(tn, nq, hd)
            k = self.linear(y, p + "self_attn.k_proj.weight", p + "self_attn.k_proj.bias").reshape(tn, nkv, hd)
            vv = self.linear(y, p + "self_attn.v_proj.weight", p + "self_attn.v_proj.bias").reshape(tn, nkv, hd)

            def rope(x):
                x1, x2 = x[..., : hd // 2], x[..., hd // 2:]
                rh = np.concatenate([-x2, x1], axis=-1)
                return self.r(self.r(x * cos[:, None, :]) + self.r(rh * sin[:, None, :]))

            q, k = rope(q), rope(k)
            if self.k_cache[li] is None:
                self.k_cache[li], self.v_cache[li] = k, vv
            else:
                self.k_cache[li] = np.concatenate([self.k_cache[li], k], axis=0)
                self.v_cache[li] = np.concatenate([self.v_cache[li], vv], axis=0)
            kk, vk = self.k_cache[li], self.v_cache[li]
            ctx = kk.shape[0]
            o = np.empty((tn, nq, hd), dtype=np.float32)
            mask = np.arange(ctx)[None, :] > (past + np.arange(tn))[:, None]
            for hq in range(nq):
                s = (q[:, hq] @ kk[:, hq // grp].T) * scale
                s = np.where(mask, np.float32(-np.inf), s)
                o[:, hq] = _softmax(s) @ vk[:, hq // grp]
            o = self.r(o).reshape(tn, -1)
            h = self.r(h + self.linear(o, p + "self_attn.o_proj.weight"))
            y = self.rmsnorm(h, p + "post_attention_layernorm.weight", t.rms_norm_eps)
            if self.act_fp8:
                y = fp8.quantize_rows(y)[2]
            h = self.r(h + self.swiglu(y, p + "mlp.", bias=False))
            if return_layers:
                layers.append(h.copy())
        self.ctx = past + tn
        hn = self.rmsnorm(h, "model.language_model.norm.weight", t.rms_norm_eps)
        return (hn, layers) if return_layers else hn

    def embed(self, input_ids, image_embeds=None):
        e = self.w["model.language_model.embed_tokens.weight"][np.asarray(input_ids)]
        e = np.array(e, dtype=np.float32)
        if image_embeds is not None:
            m = np.asarray(input_ids) == self.cfg.image_token_id
            if int(m.sum()) != image_embeds.shape[0]:
                raise ValueError(
                    f"Image features and image tokens do not match, tokens: {int(m.sum())}, "
                    f"features: {image_embeds.shape[0]}")
            e[m] = image_embeds
        return e

    def logits(self, hn_last):
        # lm_head in model dtype, then the fp32 copy (HF:generation/utils.py:2894)
        return self.r(hn_last @ self.w["lm_head.weight"].T).astype(np.float32)

    def _prefill_hidden(self, input_ids, pixel_values=None, grid_thw=None, image_embeds=None, return_layers=False):
        self.reset()
        ids = np.asarray(input_ids, dtype=np.int64)
        if image_embeds is None and pixel_values is not None:
            image_embeds = self.vit_forward(pixel_values, grid_thw)
        if grid_thw is not None:
            pos, delta = indices.rope_index(ids[None, :], grid_thw, self.cfg.image_token_id,
                                            merge=self.cfg.vision.spatial_merge_size)
            pos3 = pos[:, 0]
            self.rope_delta = int(delta[0, 0])
        else:
            pos3 = np.tile(np.arange(len(ids))[None, :], (3, 1))
            self.rope_delta = 0
        h = self.embed(ids, image_embeds)
        return self.text_forward(h, pos3, return_layers=return_layers)

    def prefill(self, input_ids, pixel_values=None, grid_thw=None, image_embeds=None, return_layers=False):
        """Single sequence (no padding).  Returns fp32 logits [vocab] of the last position."""
        out = self._prefill_hidden(input_ids, pixel_values, grid_thw, image_embeds, return_layers)
        hn = out[0] if return_layers else out
        lg = self.logits(hn[-1])
        return (lg, out[1]) if return_layers else lg

    def per_token_logps(self, input_ids, pixel_values=None, grid_thw=None, image_embeds=None):
        """Log-probability of every next id, fp32 [len - 1]: logits[:-1].log_softmax(-1).gather(ids[1:]) as
        `_get_per_token_logps` computes it (src/train/RL/src/open-r1-multimodal/src/open_r1/trainer/
        grpo_trainer.py:494-504); the logits are the lm_head output in model dtype, the log-softmax is fp32."""
        ids = np.asarray(input_ids, dtype=np.int64)
        hn = self._prefill_hidden(ids, pixel_values, grid_thw, image_embeds)
        lg = self.r(hn[:-1] @ self.w["lm_head.weight"].T).astype(np.float32)
        m = lg.max(axis=-1)
        lse = m + np.log(np.exp(lg - m[:, None]).sum(axis=-1, dtype=np.float32))
        return (lg[np.arange(len(ids) - 1), ids[1:]] - lse).astype(np.float32)

    def decode_step(self, token: int):
        p = self.ctx + self.rope_delta
        pos3 = np.full((3, 1), p, dtype=np.int64)
        h = self.embed(np.array([token]))
        hn = self.text_forward(h, pos3)
        return self.logits(hn[-1])


def apply_repetition_penalty(logits: np.ndarray, seen_ids, penalty: float) -> np.ndarray:
    """HF:generation/logits_process.py:409-413 on an fp32 row; every id present in the sequence
    (prompt included) is penalised once."""
    out = logits.copy()
    idx = np.unique(np.asarray(list(seen_ids), dtype=np.int64))
    sc = out[idx]
    out[idx] = np.where(sc < 0, sc * np.float32(penalty), sc / np.float32(penalty))
    return out


SAMPLE_BLOCKS = 128


def sample_uniform(seed: int, slot: int, index: int) -> np.float32:
    """The draw u in [0, 1) of generated token `index` of the chain in row `slot` of the generate call (0 for a
    single-chain call; include/zoomearth.h, ze_op_sample_temperature): 24 high bits of
    stream64(mix64(seed ^ mix64(row + 1)), index)."""
    m = (1 << 64) - 1
    inner = int(prng.mix64(np.array([(slot + 1) & m], dtype=np.uint64))[0])
    key = int(prng.mix64(np.array([(seed ^ inner) & m], dtype=np.uint64))[0])
    h = int(prng.stream64(key, index, 1)[0])
    return np.float32(h >> 40) * np.float32(2.0 ** -24)


def sample_temperature(logits: np.ndarray, seen_ids, penalty: float, temperature: float, seed: int, slot: int,
                       index: int):
    """Temperature sampling as the reference runs it (src/eval/infer.py:109-115: do_sample=True, temperature=0.01,
    top_k = top_p = None -> HF TemperatureLogitsWarper + softmax + multinomial, HF:generation/utils.py:2894-2916),
    with this repo's counter-based uniform instead of torch's generator and the inverse-CDF rule / fp32 summation
    order of zoomearth_amd/csrc/ze_sample.hip.  Returns (token, gap): gap = distance of the target to the nearest
    boundary of the chosen token's CDF interval, relative to the total mass -- draws with a gap below ~1e-5 may
    legitimately differ between implementations whose expf differs in the last bit."""
    f32 = np.float32
    sc = apply_repetition_penalty(np.asarray(logits, dtype=f32), seen_ids, penalty) if penalty != 1.0 else \
        np.asarray(logits, dtype=f32)
    vocab = sc.shape[0]
    t = f32(temperature)
    zmax = f32(sc.max()) / t
    e = np.exp((sc / t - zmax).astype(f32)).astype(f32)
    chunk = -(-vocab // SAMPLE_BLOCKS)
    run = -(-chunk // 256)
    pad = np.zeros(SAMPLE_BLOCKS * 256 * run, dtype=f32)
    # element i sits in chunk i // chunk, run (i % chunk) // run, position (i % chunk) % run
    idx = np.arange(vocab)
    pos = (idx // chunk) * (256 * run) + (idx % chunk)
    pad[pos] = e
    grid = pad.reshape(SAMPLE_BLOCKS, 256, run)
    run_cum = np.cumsum(grid, axis=2, dtype=f32)            # sequential fp32 adds inside a run
    run_sum = run_cum[:, :, -1]
    thr_cum = np.cumsum(run_sum, axis=1, dtype=f32)          # the 256 run sums in order
    chunk_sum = thr_cum[:, -1]
    blk_cum = np.cumsum(chunk_sum, dtype=f32)                # the chunk sums in order
    total = blk_cum[-1]
    u = sample_uniform(seed, slot, index)
    target = f32(u * total)
    hit = np.nonzero(blk_cum > target)[0]
    if hit.size == 0:
        blk = int(np.nonzero(chunk_sum > 0)[0][-1])
        tgt_in = f32(-np.inf)
    else:
        blk = int(hit[0])
        tgt_in = f32(target - (blk_cum[blk - 1] if blk else f32(0)))
    token = -1
    cum = f32(0)
    for th in range(256):
        if f32(cum + run_sum[blk, th]) > tgt_in:
            for j in range(run):
                i = blk * chunk + th * run + j
                if i >= min(vocab, (blk + 1) * chunk):
                    break
                cum = f32(cum + grid[blk, th, j])
                if cum > tgt_in:
                    token = i
                    break
            if token >= 0:
                break
        else:
            cum = f32(cum + run_sum[blk, th])
    if token < 0:
        nz = np.nonzero(e[blk * chunk:min(vocab, (blk + 1) * chunk)] > 0)[0]
        token = blk * chunk + int(nz[-1])
    # gap in float64 against the exact CDF
    cdf = np.cumsum(e.astype(np.float64))
    tot = cdf[-1]
    x = float(u) * tot
    lo = cdf[token - 1] if token else 0.0
    gap = min(abs(x - lo), abs(cdf[token] - x)) / tot
    return token, float(gap)


def greedy_generate(model: Qwen25VLOracle, input_ids, pixel_values=None, grid_thw=None, max_new_tokens=16,
                    repetition_penalty: float = 1.0, eos_token_ids=None, forced_tokens=None,
                    return_logits=True):
    """Greedy loop for one sequence.  forced_tokens (teacher forcing) feeds the given ids instead of
    the argmax while still recording this model's own raw logits / argmax at every step.

    Returns dict(tokens=[...chosen/fed...], argmax=[...], logits=[steps, vocab] raw fp32 logits,
    margins=[top1-top2 of the PROCESSED scores]).
    """
    eos = set(eos_token_ids if eos_token_ids is not None else model.cfg.eos_token_ids)
    seq = [int(t) for t in input_ids]
    lg = model.prefill(seq, pixel_values=pixel_values, grid_thw=grid_thw)
    toks, amax, logs, margins = [], [], [], []
    for step in range(max_new_tokens):
        sc = apply_repetition_penalty(lg, seq, repetition_penalty) if repetition_penalty != 1.0 else lg
        a = int(np.argmax(sc))
        top2 = np.partition(sc, -2)[-2:]
        margins.append(float(top2[1] - top2[0]))
        amax.append(a)
        if return_logits:
            logs.append(lg.copy())
        nxt = int(forced_tokens[step]) if forced_tokens is not None else a
        toks.append(nxt)
        seq.append(nxt)
        if forced_tokens is None and nxt in eos:
            break
        if step + 1 < max_new_tokens:
            lg = model.decode_step(nxt)
    return dict(tokens=toks, argmax=amax, logits=np.stack(logs) if logs else None, margins=margins)
