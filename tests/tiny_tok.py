"""A tiny in-memory tokenizer for the tiny config (no checkpoint files offline): word-level vocab w0..w1999 plus
the Qwen special tokens at the ids ModelConfig.tiny() declares."""
from zoomearth_amd.tokenizer import ZoomEarthTokenizer

SPECIALS = {"<|endoftext|>": 2043, "<|im_end|>": 2045, "<|im_start|>": 2044, "<|vision_start|>": 2002,
            "<|vision_end|>": 2003, "<|image_pad|>": 2005, "<unk>": 2047}


def make_tokenizer() -> ZoomEarthTokenizer:
    from tokenizers import AddedToken, Tokenizer
    from tokenizers.decoders import Fuse  # noqa: F401
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import WhitespaceSplit

    vocab = {f"w{i}": i for i in range(2000)}
    vocab.update(SPECIALS)
    tok = Tokenizer(WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = WhitespaceSplit()
    tok.add_special_tokens([AddedToken(t, special=True) for t in SPECIALS if t != "<unk>"])
    return ZoomEarthTokenizer(tok, pad_token="<|endoftext|>")


# ----------------------------------------------------------------------------- a REAL byte-level BPE
# The reference decodes the stage-1 output, strips it and tokenises it AGAIN inside the stage-2 prompt
# (/root/reference/src/eval/infer.py:118-123, 153-157, 222-225).  With the word-level tokenizer above that round trip is the
# identity; with a byte-level BPE it is not -- pieces re-merge across what the model emitted as separate tokens, bytes that are no
# valid UTF-8 come back as U+FFFD -- so the scheduler's "keep the rows the decode steps wrote while the re-tokenised ids repeat the
# generated ones" really stops early.  Trained here, in the test set-up, with `tokenizers.trainers.BpeTrainer` on the prompt text and
# synthetic answers (no file of the reference is read): ByteLevel alphabet (256 byte tokens) + merges up to id 2001, then the Qwen
# special tokens at the ids ModelConfig.tiny() / the test checkpoints declare (2002 .. 2047, unused ids reserved).
def bpe_word(i: int, three_number: bool = True) -> str:
    """The vocabulary of tests/test_gpu_infer_e2e.py: plain words and `"bbox_2d":[...]` fragments (some with three numbers)."""
    if i % 3 == 0:
        return f"w{i}"
    x, y = (i * 37) % 400, (i * 91) % 300
    if i % 11 == 1 and three_number:
        return f'"bbox_2d":[{x},{y},{x + 40}]'
    return f'"bbox_2d":[{x},{y},{x + 30 + i % 200},{y + 20 + i % 150}]'


def train_bpe(n_words: int = 260, base_vocab: int = 2002, total_vocab: int = 2048, three_number: bool = True):
    """-> tokenizers.Tokenizer (byte-level BPE, ByteLevel decoder, specials at the tiny config's ids)."""
    import json

    import numpy as np
    from tokenizers import AddedToken, Regex, Tokenizer
    from tokenizers.decoders import ByteLevel as ByteLevelDecoder
    from tokenizers.models import BPE
    from tokenizers.pre_tokenizers import ByteLevel, Sequence, Split
    from tokenizers.trainers import BpeTrainer

    from zoomearth_amd import hostloop

    tok = Tokenizer(BPE())
    # pre-tokens = a word with its leading space (so that frequent whole fragments become ONE token and rare ones stay in pieces)
    tok.pre_tokenizer = Sequence([Split(Regex(r" ?[^ ]+"), behavior="isolated"), ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.decoder = ByteLevelDecoder()
    words = [bpe_word(i, three_number) for i in range(n_words)]
    rng = np.random.default_rng(0)
    corpus = [" ".join(words[j] for j in rng.integers(0, n_words, 20)) for _ in range(400)]
    prompts = [hostloop.stage1_prompt("which w3 is next to the w6 ?"), hostloop.stage2_prompt(hostloop.stage1_prompt("w9 w12"), words[4])]
    for sp in SPECIALS:   # (added tokens are cut out of the text before the model sees it: they must not become merges)
        prompts = [p.replace(sp, " ") for p in prompts]
    corpus += prompts * 8
    tok.train_from_iterator(corpus, BpeTrainer(vocab_size=base_vocab, special_tokens=[], initial_alphabet=ByteLevel.alphabet(), show_progress=False))
    j = json.loads(tok.to_str())
    vocab = j["model"]["vocab"]
    for k in range(len(vocab), base_vocab):   # (a short corpus may run out of pairs: unused ids up to the specials)
        vocab[f"<|fill{k}|>"] = k
    tok = Tokenizer.from_str(json.dumps(j))
    by_id = {v: k for k, v in SPECIALS.items() if k != "<unk>"}
    tok.add_special_tokens([AddedToken(by_id.get(i, f"<|reserved{i}|>"), special=True) for i in range(base_vocab, total_vocab)])
    for k, v in by_id.items():
        assert tok.token_to_id(v) == k, (v, tok.token_to_id(v))
    return tok


def make_bpe_tokenizer(**kw) -> ZoomEarthTokenizer:
    return ZoomEarthTokenizer(train_bpe(**kw), pad_token="<|endoftext|>")
