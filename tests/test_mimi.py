"""Mimi decode.  CPU: the oracle (oracle/mimi_oracle.py) against the waveforms transformers'
MimiModel.decode produced for the committed codes (tests/golden/mimi_*.npz, oracle/make_golden_mimi.py).  GPU: the HIP path
(csm_mimi_decode through the C ABI) against the same fixtures and the oracle."""
import os

import numpy as np
import pytest
import torch

from csm_hf_amd.mimi import MimiDecodeConfig, synth_mimi_state_dict, mimi_state_dict_spec
from oracle import mimi_oracle as MO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {"tiny": MimiDecodeConfig.tiny(), "full": MimiDecodeConfig()}


def rel_max(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize("name", ["tiny", "full"])
def test_oracle_vs_transformers_golden(name):
    cfg = CASES[name]
    g = np.load(os.path.join(GOLD, f"mimi_{name}.npz"))
    sd = synth_mimi_state_dict(cfg, seed=0)
    assert set(sd) == {k for k, _, _ in mimi_state_dict_spec(cfg)}
    codes = torch.from_numpy(g["codes"])
    out = MO.decode(sd, cfg, codes)
    assert out.shape == g["audio"].shape == (codes.shape[0], 1, codes.shape[2] * cfg.samples_per_frame)
    assert rel_max(out, torch.from_numpy(g["audio"])) < 1e-5
    assert float(g["oracle_max_rel_err"]) < 1e-5
    if name == "tiny":
        # causality: a later frame does not change earlier samples (causal convolutions, causal attention)
        c2 = codes.clone()
        c2[:, :, -1] = (c2[:, :, -1] + 1) % cfg.codebook_size
        out2 = MO.decode(sd, cfg, c2)
        keep = (codes.shape[2] - 1) * cfg.samples_per_frame
        assert torch.equal(out[..., :keep], out2[..., :keep]) and not torch.equal(out, out2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "full"])
def test_hip_decode_vs_transformers_golden_and_oracle(name):
    """fp32 path: every product on the exact-fp32 MFMA, tolerance 1e-4 of the waveform's peak (summation order)."""
    from csm_hf_amd import MimiDecoder
    cfg = CASES[name]
    g = np.load(os.path.join(GOLD, f"mimi_{name}.npz"))
    sd = synth_mimi_state_dict(cfg, seed=0)
    dec = MimiDecoder(cfg, sd, "cuda:0", max_frames=64)
    codes = torch.from_numpy(g["codes"])
    out = dec.decode(codes.to("cuda:0")).cpu()
    assert out.shape == g["audio"].shape
    assert rel_max(out, torch.from_numpy(g["audio"])) < 1e-4
    # other lengths and a batch, against the oracle: one frame, a length that is not the fixture's, window wrap (tiny: 6)
    gen = torch.Generator().manual_seed(5)
    for B, T in ((1, 1), (3, 17 if name == "tiny" else 5)):
        c = torch.randint(0, cfg.codebook_size, (B, cfg.num_quantizers, T), generator=gen)
        want = MO.decode(sd, cfg, c)
        got = dec.decode(c.to("cuda:0")).cpu()
        assert rel_max(got, want) < 1e-4, (B, T)
    with pytest.raises(ValueError):
        dec.decode(torch.zeros(1, cfg.num_quantizers, 65, dtype=torch.long))          # beyond max_frames
    with pytest.raises(ValueError):
        dec.decode(torch.full((1, cfg.num_quantizers, 2), cfg.codebook_size, dtype=torch.long))
    dec.close()
