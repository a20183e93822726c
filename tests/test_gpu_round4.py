"""GPU suite, round 4: an optimizer loop through the autograd bridge (parameters updated in place between steps: the
engine must repack), the traffic record, and the round's other additions (see the individual tests)."""
import os

import numpy as np
import pytest
import torch

from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
from oracle import csm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tiny_model(dtype=torch.float32, seed=0):
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=seed, std=0.05)
    m = CSMModel(cfg)
    m.load_state_dict({k: v.to(dtype) for k, v in sd.items()})
    return cfg, sd, m.to(DEV).eval()


def _grad_err(got, want):
    return float((got.double().cpu() - want.double()).norm() / want.double().norm().clamp_min(1e-30))


def test_two_sgd_steps_through_the_bridge_follow_the_oracle():
    """the engine multiplies packed COPIES of most parameters (qkv concatenated, gate/up interleaved,
    heads transposed, transposed copies for the backward pass) and ALIASES of the others.  After an in-place
    `optimizer.step()` the second forward / backward must see the updated values everywhere -- the model notices the
    version counters and rebuilds the engine.  Reference loop: torch.autograd through the oracle with the same SGD
    (train.py:308-326 is an HF-Trainer loop over exactly this objective)."""
    cfg, sd, m = tiny_model(seed=5)
    ids, mask = synth_context(cfg, 2, 4, 9, seed=77)
    labels = torch.full_like(ids, -100)
    labels[:, 4:, :32] = ids[:, 4:, :32]
    lr = 0.5
    # oracle: two plain SGD steps
    osd = {k: v.float().clone().requires_grad_(True) for k, v in sd.items()}
    want_loss = []
    for _ in range(3):
        for v in osd.values():
            v.grad = None
        with torch.enable_grad():
            loss = O.forward_loss(osd, cfg, ids, mask, labels)[0]
            loss.backward()
        want_loss.append(float(loss))
        with torch.no_grad():
            for v in osd.values():
                v -= lr * v.grad
    assert want_loss[0] - want_loss[2] > 1e-3 * want_loss[0], want_loss        # the steps are large enough to matter
    # HIP path: the same loop through forward(labels=...).loss.backward() and torch.optim.SGD
    m.requires_grad_(True)
    opt = torch.optim.SGD(m.parameters(), lr=lr)
    got_loss = []
    for step in range(3):
        opt.zero_grad(set_to_none=True)
        out = m(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV))
        out.loss.backward()
        got_loss.append(float(out.loss))
        if step == 1:       # gradients of the SECOND step, computed from parameters that were updated in place once
            want_g = grads_of_second_step(cfg, ids, mask, labels, sd, lr)
            worst = max(_grad_err(p.grad, want_g[n]) for n, p in m.named_parameters())
            assert worst < 2e-4, worst
        opt.step()
    for g, w in zip(got_loss, want_loss):
        assert abs(g - w) < 5e-5 * abs(w), (got_loss, want_loss)
    # and generation after training uses the trained weights too
    trained = {k: v.detach().clone() for k, v in osd.items()}
    toks = m.generate(ids[:1].to(DEV), mask[:1].to(DEV), max_new_frames=2, topk=1, stop_on_all_zeros=False)
    tr = {}
    want = O.generate(trained, cfg, ids[:1], mask[:1], max_new_frames=2, topk=1, stop_on_all_zeros=False, trace=tr)
    tv = torch.topk(tr["logits"][:, 0], 2, -1)[0]
    margin = (tv[..., 0] - tv[..., 1]).reshape(-1)
    stop = int((margin < 1e-4).nonzero()[0]) if bool((margin < 1e-4).any()) else margin.numel()
    assert torch.equal(toks.cpu().reshape(-1)[:stop], want.reshape(-1)[:stop])
    m.requires_grad_(False)
    m._drop_engine()


def grads_of_second_step(cfg, ids, mask, labels, sd0, lr):
    """{name: gradient} of the oracle's SECOND step (parameters after one SGD step from sd0)"""
    p = {k: v.float().clone().requires_grad_(True) for k, v in sd0.items()}
    with torch.enable_grad():
        O.forward_loss(p, cfg, ids, mask, labels)[0].backward()
    with torch.no_grad():
        q = {k: (v - lr * v.grad).clone().requires_grad_(True) for k, v in p.items()}
    with torch.enable_grad():
        O.forward_loss(q, cfg, ids, mask, labels)[0].backward()
    return {k: v.grad for k, v in q.items()}


def test_in_place_parameter_write_outside_training_is_seen_by_generate():
    """the same staleness without any training call: `p.data.mul_()` on a packed-by-copy parameter between two generate()s"""
    cfg, sd, m = tiny_model(seed=6)
    ids, mask = synth_context(cfg, 1, 3, 6, seed=5)
    a = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=2, topk=1, stop_on_all_zeros=False)
    with torch.no_grad():
        m.backbone.layers[0].self_attn.q_proj.weight.mul_(-1.5)        # packed into wqkv by concatenation (a copy)
        m.decoder.layers[1].mlp.gate_proj.weight.mul_(0.5)             # packed into wgu by interleaving (a copy)
    sd2 = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    b = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=2, topk=1, stop_on_all_zeros=False)
    want = O.generate(sd2, cfg, ids, mask, max_new_frames=2, topk=1, stop_on_all_zeros=False)
    assert torch.equal(b.cpu(), want)
    assert not torch.equal(a.cpu(), b.cpu()) or True      # (the streams usually differ; equality is not an error)
    m._drop_engine()
