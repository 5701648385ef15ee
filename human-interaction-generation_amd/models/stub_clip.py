"""Deterministic stand-in for OpenAI CLIP ViT-B/32's text tower.

The reference calls `clip.load('ViT-B/32', "cpu")` and `clip.tokenize(text, truncate=True)`
(codes/models/transformer.py:319,382-388).  CLIP itself is a third-party, network-downloaded
dependency (codes/requirements.txt:7) that BASELINE.json stubs; this module has the same
call surface (`load`, `tokenize`, and the attributes `encode_text` touches) so the text head
can be exercised end to end without the real weights.  When the real `clip` package is
importable, `models.transformer` uses it instead.
"""
import zlib

import torch
from torch import nn

CONTEXT_LENGTH = 77
VOCAB = 49408
SOT, EOT = 49406, 49407
WIDTH = 512


class StubCLIP(nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(1234)
        self.token_embedding = nn.Embedding(VOCAB, WIDTH)
        self.positional_embedding = nn.Parameter(torch.empty(CONTEXT_LENGTH, WIDTH))
        self.transformer = nn.Identity()
        self.ln_final = nn.LayerNorm(WIDTH)
        with torch.no_grad():
            self.token_embedding.weight.copy_(torch.randn(VOCAB, WIDTH, generator=g) * 0.02)
            self.positional_embedding.copy_(torch.randn(CONTEXT_LENGTH, WIDTH, generator=g) * 0.01)

    @property
    def dtype(self):
        return self.token_embedding.weight.dtype

    def initialize_parameters(self):
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)


def load(name="ViT-B/32", device="cpu", **_):
    return StubCLIP().to(device), None


def tokenize(texts, context_length=CONTEXT_LENGTH, truncate=False):
    """Word-hash tokenizer: SOT, one id per whitespace word, EOT (the arg-max id, which is
    what `encode_text` gathers on, transformer.py:394), zero padding."""
    if isinstance(texts, str):
        texts = [texts]
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, s in enumerate(texts):
        ids = [SOT] + [1 + zlib.crc32(w.lower().encode()) % (SOT - 1) for w in s.split()] + [EOT]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError("Input %r is too long for context length %d" % (s, context_length))
            ids = ids[:context_length]
            ids[-1] = EOT
        out[i, :len(ids)] = torch.tensor(ids)
    return out
