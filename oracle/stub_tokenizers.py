"""Deterministic stand-ins for the Llama text tokenizer and the Mimi codec (neither is available offline,
SURVEY.md section 0 finding 5).  TEST INFRASTRUCTURE: used by oracle/make_golden.py to drive the REFERENCE
processor and by tests/ to drive ours with identical inputs."""
import torch
import torch.nn as nn


class StubTextTokenizer:
    bos, eos = 128000, 128001

    def encode(self, text, add_special_tokens=True):
        ids = [3 + (ord(c) * 131 + i * 7) % 1000 for i, c in enumerate(text)]
        return [self.bos] + ids + [self.eos] if add_special_tokens else ids


class StubAudioTokenizer(nn.Module):
    """`encode(wav [1,1,T]) -> [1, 32, T // 1920]` integer codes (24 kHz / 12.5 Hz = 1920 samples per frame)."""
    sample_rate = 24000

    def __init__(self):
        super().__init__()
        self.dummy = nn.Parameter(torch.zeros(1))

    def encode(self, wav):
        F = wav.shape[-1] // 1920
        x = wav[0, 0, : F * 1920].reshape(F, 1920)
        base = (x.abs().sum(-1) * 1000).long()                   # [F]
        cb = torch.arange(32)[:, None]
        return ((base[None, :] + cb * 37) % 2050 + 1).unsqueeze(0).float()   # float like Mimi's int codes cast
