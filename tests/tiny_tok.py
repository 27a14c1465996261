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
