"""Tokenizer shim with the slice of the HF tokenizer API the entry points use
(`processor.tokenizer.padding_side = "left"`, `processor.tokenizer.decode(ids, skip_special_tokens=True)`;
/root/reference/src/eval/infer.py:122,157; src/demo.py:26).

Tokenisation is host-side and not part of the hot path: this wraps the `tokenizers` (Rust) library that HF itself
uses for Qwen2TokenizerFast, loading the checkpoint's tokenizer.json.
"""
from __future__ import annotations

import json
import os


class ZoomEarthTokenizer:
    def __init__(self, tok, pad_token: str = "<|endoftext|>", padding_side: str = "right"):
        self._tok = tok
        self.padding_side = padding_side
        self.pad_token = pad_token
        self.pad_token_id = tok.token_to_id(pad_token)
        if self.pad_token_id is None:
            raise ValueError(f"pad token {pad_token!r} is not in the vocabulary")

    @classmethod
    def from_pretrained(cls, path: str, **kw):
        from tokenizers import Tokenizer

        tok = Tokenizer.from_file(os.path.join(path, "tokenizer.json"))
        pad = "<|endoftext|>"
        cfgp = os.path.join(path, "tokenizer_config.json")
        if os.path.exists(cfgp):
            with open(cfgp, encoding="utf-8") as f:
                c = json.load(f)
            p = c.get("pad_token")
            if isinstance(p, dict):
                p = p.get("content")
            pad = p or pad
        return cls(tok, pad_token=pad)

    def convert_tokens_to_ids(self, token: str):
        return self._tok.token_to_id(token)

    def encode(self, text: str):
        return self._tok.encode(text, add_special_tokens=False).ids

    def __call__(self, text, padding=False, return_tensors=None, **kw):
        texts = [text] if isinstance(text, str) else list(text)
        rows = [self.encode(t) for t in texts]
        width = max(len(r) for r in rows) if padding in ("longest", True) else None
        ids, mask = [], []
        for r in rows:
            pad = (width - len(r)) if width is not None else 0
            if self.padding_side == "left":
                ids.append([self.pad_token_id] * pad + r)
                mask.append([0] * pad + [1] * len(r))
            else:
                ids.append(r + [self.pad_token_id] * pad)
                mask.append([1] * len(r) + [0] * pad)
        if return_tensors == "pt":
            import torch
            return {"input_ids": torch.tensor(ids, dtype=torch.long), "attention_mask": torch.tensor(mask, dtype=torch.long)}
        return {"input_ids": ids, "attention_mask": mask}

    def decode(self, ids, skip_special_tokens: bool = False, **kw) -> str:
        if hasattr(ids, "tolist"):
            ids = ids.tolist()
        return self._tok.decode([int(i) for i in ids], skip_special_tokens=skip_special_tokens)

    def batch_decode(self, batch, skip_special_tokens: bool = False, **kw):
        return [self.decode(r, skip_special_tokens=skip_special_tokens) for r in batch]
