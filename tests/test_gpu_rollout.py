"""GPU: the two-stage rollout driver (zoomearth_amd/rollout.py; the generation half of the reference's GRPO step,
/root/reference/src/train/RL/src/open-r1-multimodal/src/open_r1/trainer/grpo_trainer.py:561-683) on the tiny config:
G sampled chains per prompt advance together, stage 2 zooms into each chain's own box, samples without a reference box
skip stage 2, runs are reproducible from the seed, chains of one prompt differ, and the returned log-probabilities are
`model.per_token_logps` of the final sequence from the stage-1 prompt length on."""
import numpy as np
import pytest
import torch

from gpu_util import CHAIN_W
from oracle import prng
from test_gpu_infer_e2e import word
from zoomearth_amd import hostloop as H
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.image import DeviceImage
from zoomearth_amd.modeling import ZoomEarthForConditionalGeneration
from zoomearth_amd.processor import ZoomEarthProcessor
from zoomearth_amd.rollout import rollout_two_stage
from zoomearth_amd.tokenizer import ZoomEarthTokenizer

pytestmark = pytest.mark.gpu


def bbox_tokenizer():
    from tokenizers import AddedToken, Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import WhitespaceSplit
    specials = {"<|endoftext|>": 2043, "<|im_end|>": 2045, "<|im_start|>": 2044, "<|vision_start|>": 2002,
                "<|vision_end|>": 2003, "<|image_pad|>": 2005, "<unk>": 2047}
    vocab = {word(i): i for i in range(2000)}
    vocab.update(specials)
    tok = Tokenizer(WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = WhitespaceSplit()
    tok.add_special_tokens([AddedToken(t, special=True) for t in specials if t != "<unk>"])
    return ZoomEarthTokenizer(tok, pad_token="<|endoftext|>")


def test_two_stage_rollout_g_chains_per_prompt():
    model = ZoomEarthForConditionalGeneration.from_synthetic(ModelConfig.tiny(), **CHAIN_W, max_seqs=8, max_ctx=2048,
                                                            max_patches=8192, max_tile_side=2048, max_prefill_rows=8192)
    try:
        proc = ZoomEarthProcessor(bbox_tokenizer(), min_pixels=3136, max_pixels=128 * 128 * 28 * 28)
        tiles = [DeviceImage.from_numpy(prng.synthetic_tile(90 + t, 700, 900), model.engine) for t in range(2)]
        samples = []
        for i in range(3):
            q = " ".join(word(int(v)) for v in prng.uniform_ints(70 + i, 5, 0, 1999))
            samples.append(dict(prompt=H.stage1_prompt(q), image=tiles[i % 2], bbox=[1, 2, 3, 4] if i != 1 else []))
        G = 4
        a = rollout_two_stage(model, proc, samples, num_generations=G, temperature=0.9, max_new_tokens=10, seed=11)
        b = rollout_two_stage(model, proc, samples, num_generations=G, temperature=0.9, max_new_tokens=10, seed=11, with_logps=False)
        c = rollout_two_stage(model, proc, samples, num_generations=G, temperature=0.9, max_new_tokens=10, seed=12, with_logps=False)
        assert len(a) == 3 * G and [(r.sample, r.generation) for r in a] == [(i, g) for i in range(3) for g in range(G)]
        assert all(r.error is None for r in a)
        assert [(r.completion1_ids, r.completion2_ids) for r in a] == [(r.completion1_ids, r.completion2_ids) for r in b]
        assert [r.completion1_ids for r in a] != [r.completion1_ids for r in c]           # the seed matters
        for i in range(3):
            firsts = {tuple(r.completion1_ids) for r in a if r.sample == i}
            assert len(firsts) > 1, "the G chains of a prompt must not all coincide at T = 0.9"
        for r in a:
            if r.sample == 1:                                                             # no reference box: stage 2 skipped
                assert r.prompt2 is None and r.completion2_ids == [] and len(r.images) == 1
                tail = r.completion1_ids
            else:
                assert r.prompt2 == H.stage2_prompt(r.prompt1, r.completion1) and len(r.images) == 2
                assert len(r.bbox) == 4 and r.scale == 900 / 512 and r.images[1].size[0] <= 512
                tail = r.completion2_ids
            assert 1 <= len(tail) <= 10
            # scoring = model.per_token_logps on the final sequence, from the stage-1 prompt length on
            prompt = r.prompt2 if r.prompt2 is not None else r.prompt1
            inp = proc(text=[prompt], images=list(r.images), return_tensors="pt")
            ids = torch.cat([inp["input_ids"], torch.tensor([tail])], dim=1)
            want = model.per_token_logps(ids, torch.ones_like(ids), inp["pixel_values"], inp["image_grid_thw"])[0, r.n_prompt1 - 1:]
            assert r.logps.shape == want.shape and r.logps.shape[0] == ids.shape[1] - r.n_prompt1
            assert torch.equal(r.logps.cpu(), want.cpu())
            assert np.isfinite(r.logps.cpu().numpy()).all() and (r.logps.cpu().numpy() <= 0).all()
    finally:
        model.engine.close()
