"""CPU suite: CSMProcessor against the reference processor's outputs on stub
tokenizers (tests/golden/processor.npz, produced by oracle/make_golden.py --only processor)."""
import random

import numpy as np
import pytest
import torch

from csm_hf_amd.processor import CSMProcessor
from oracle.stub_tokenizers import StubTextTokenizer, StubAudioTokenizer


def cases():
    g = torch.Generator().manual_seed(0)
    wavs = [torch.rand(1920 * 5 + 100, generator=g), torch.rand(1920 * 3, generator=g), torch.rand(1920 * 9, generator=g)]
    convo1 = [{"role": "speaker_0", "content": [{"type": "text", "text": "Hello there"}, {"type": "audio"}]},
              {"role": "speaker_1", "content": [{"type": "text", "text": "Hi"}, {"type": "audio"}]},
              {"role": "speaker_0", "content": [{"type": "text", "text": "How are you today?"}]}]
    convo2 = [{"role": "speaker_3", "content": [{"type": "text", "text": "Short"}, {"type": "audio"}]}]
    return {
        "single": dict(messages=convo1, audios=wavs[:2]),
        "single_noamort": dict(messages=convo1, audios=wavs[:2], amortize_decoder_training=False, messages_training_mask=[1, 0, 1]),
        "trunc": dict(messages=convo1, audios=wavs[:2], max_length=20, amortize_decoder_training=False),
        "batch": dict(messages=[convo1, convo2], audios=[wavs[:2], [wavs[2]]], amortization_ratio=4),
    }


@pytest.fixture(scope="module")
def proc():
    return CSMProcessor(StubTextTokenizer(), StubAudioTokenizer())


@pytest.mark.parametrize("name", ["single", "single_noamort", "trunc", "batch"])
def test_processor_matches_reference(gold, proc, name):
    g = gold("processor")
    random.seed(5)
    out = proc(**cases()[name])
    for k in ("input_ids", "attention_mask", "labels"):
        assert np.array_equal(out[k].long().numpy(), g[f"{name}.{k}"].astype(np.int64)), (name, k)
    assert out["input_ids"].dtype == torch.long and out["labels"].dtype == torch.long
    assert out["attention_mask"].dtype == torch.int32          # also in the padded path (documented deviation)


def test_processor_layout_and_fixed_modes(proc):
    out = proc(**cases()["single"], amortize_decoder_training=False)
    ids, m = out["input_ids"][0], out["attention_mask"][0]
    text_rows = m[:, 32] == 1
    assert bool((m[text_rows, :32] == 0).all()) and bool((ids[text_rows, :32] == 0).all())
    audio_rows = ~text_rows
    assert bool((m[audio_rows, :32] == 1).all()) and bool((ids[audio_rows, 32] == 0).all())
    # each audio message ends with one all-zero, fully masked-in EOS frame
    eos = audio_rows & (ids[:, :32] == 0).all(-1)
    assert int(eos.sum()) == 2
    # left padding of the shorter conversation
    b = proc(**cases()["batch"])
    pad = (b["attention_mask"][1].sum(-1) == 0)
    n_pad = int(pad.sum())
    assert n_pad > 0 and bool(pad[:n_pad].all()) and not bool(pad[n_pad:].any())
    # text=/speaker_id= mode works (the reference raises "Unsupported return format: True")
    t = proc(text="hi", speaker_id=2)
    assert t["input_ids"].shape == (1, 7, 33) and int(t["attention_mask"][0, :, 32].sum()) == 7
    with pytest.raises(ValueError):
        proc()
    with pytest.raises(ValueError):
        proc(messages=cases()["single"]["messages"], return_tensors="np")
