"""Mimi decode.  CPU: the oracle (oracle/mimi_oracle.py) against the waveforms transformers'
MimiModel.decode produced for the committed codes (tests/golden/mimi_*.npz, oracle/make_golden_mimi.py).  GPU: the HIP path
(csm_mimi_decode through the C ABI) against the same fixtures and the oracle."""
import os

import numpy as np
import pytest
import torch

from csm_hf_amd.mimi import MimiDecodeConfig, synth_mimi_state_dict, mimi_state_dict_spec, load_mimi_checkpoint
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


def _long_fixture():
    g = np.load(os.path.join(GOLD, "mimi_full_long.npz"))
    return torch.from_numpy(g["codes"]), torch.from_numpy(g["audio0"]), torch.from_numpy(g["audio1_every8"]), float(g["peak"])


def _long_err(out, a0, a1s, peak):
    """distance of a [2, 1, n] waveform from the fixture (sequence 0 whole, sequence 1 every 8th sample), relative to the peak"""
    return max(float((out[0].double() - a0.double()).abs().max()), float((out[1, :, ::8].double() - a1s.double()).abs().max())) / peak


def test_oracle_past_the_attention_window_vs_transformers():
    """kyutai/mimi shape, 150 frames = 300 transformer positions > sliding_window 250: the window wraps.  The oracle against
    the waveform transformers' MimiModel.decode produced (fixture mimi_full_long, oracle/make_golden_mimi.py)."""
    cfg = CASES["full"]
    codes, a0, a1s, peak = _long_fixture()
    assert codes.shape == (2, cfg.num_quantizers, 150) and 2 * codes.shape[2] > cfg.sliding_window
    out = MO.decode(synth_mimi_state_dict(cfg, seed=0), cfg, codes)
    assert _long_err(out, a0, a1s, peak) < 1e-5


@pytest.mark.gpu
def test_hip_decode_past_the_attention_window():
    """the real shape had never wrapped its attention window against the oracle.  150 frames (window = 125),
    `max_frames = 512`: one-shot B = 2 and B = 1, and a stream in ragged chunks, all < 1e-4 of the peak against the
    transformers-generated fixture; 500 frames (the window wraps four times) against the oracle."""
    from csm_hf_amd import MimiDecoder
    cfg = CASES["full"]
    codes, a0, a1s, peak = _long_fixture()
    sd = synth_mimi_state_dict(cfg, seed=0)
    dec = MimiDecoder(cfg, sd, "cuda:0", max_frames=512)
    cd = codes.to("cuda:0")
    both = dec.decode(cd).cpu()                                   # B = 2, one shot
    assert both.shape == (2, 1, 150 * cfg.samples_per_frame) and _long_err(both, a0, a1s, peak) < 1e-4
    solo = dec.decode(cd[:1]).cpu()                               # B = 1: same sequence, other GEMM row counts
    assert float((solo[0].double() - a0.double()).abs().max()) / peak < 1e-4
    for row, chunks in ((0, (1, 7, 16, 2, 33, 5, 64, 3, 40)), (1, (13,) * 12)):      # ragged / regular chunks, both past the window
        dec.stream_reset()
        parts, t = [], 0
        for n in chunks:
            n = min(n, 150 - t)
            if n <= 0:
                break
            parts.append(dec.stream_decode(cd[row, :, t:t + n]).clone())
            t += n
        assert t == 150
        got = torch.cat(parts, dim=-1).cpu()[0]
        want = a0 if row == 0 else None
        if row == 0:
            assert float((got.double() - a0.double()).abs().max()) / peak < 1e-4, chunks
        else:
            assert float((got[:, ::8].double() - a1s.double()).abs().max()) / peak < 1e-4, chunks
    g = torch.Generator().manual_seed(17)
    c500 = torch.randint(0, cfg.codebook_size, (1, cfg.num_quantizers, 500), generator=g)
    want = MO.decode(sd, cfg, c500)
    assert rel_max(dec.decode(c500.to("cuda:0")).cpu(), want) < 1e-4
    dec.close()


def test_checkpoint_directory_round_trip(tmp_path):
    """`kyutai/mimi`-layout directory (config.json + model.safetensors, as transformers writes it) -> config + decode-path
    tensors; a weight-normalised convolution (g, v) is folded to the plain weight."""
    import json
    from safetensors.torch import save_file
    cfg = MimiDecodeConfig.tiny()
    sd = synth_mimi_state_dict(cfg, seed=0)
    hf = dict(num_quantizers=cfg.num_quantizers, num_semantic_quantizers=1, codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim,
              hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
              num_key_value_heads=cfg.num_attention_heads, head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size,
              sliding_window=cfg.sliding_window, norm_eps=cfg.norm_eps, upsampling_ratios=cfg.upsampling_ratios, num_filters=cfg.num_filters,
              kernel_size=7, last_kernel_size=3, residual_kernel_size=3, compress=2, rope_parameters={"rope_theta": 10000.0, "rope_type": "default"},
              frame_rate=12.5, sampling_rate=12.5 * cfg.samples_per_frame, use_causal_conv=True, model_type="mimi")
    stored = dict(sd)
    k = "decoder.layers.0.conv.weight"
    w = stored.pop(k)
    v = w * 3.0                                             # any direction tensor; g restores the norm
    stored["decoder.layers.0.conv.parametrizations.weight.original1"] = v
    stored["decoder.layers.0.conv.parametrizations.weight.original0"] = w.flatten(1).norm(dim=1).view(-1, 1, 1)
    stored["encoder.layers.0.conv.weight"] = torch.zeros(4, 1, 7)          # an encode-side tensor: ignored
    os.makedirs(tmp_path, exist_ok=True)
    json.dump(hf, open(os.path.join(tmp_path, "config.json"), "w"))
    save_file({a: b.contiguous() for a, b in stored.items()}, os.path.join(tmp_path, "model.safetensors"))
    cfg2, sd2 = load_mimi_checkpoint(str(tmp_path))
    assert cfg2 == cfg and set(sd2) == set(sd)
    for a in sd:
        torch.testing.assert_close(sd2[a], sd[a], atol=1e-6, rtol=1e-6)
    hf["use_conv_shortcut"] = True
    json.dump(hf, open(os.path.join(tmp_path, "config.json"), "w"))
    with pytest.raises(ValueError):
        load_mimi_checkpoint(str(tmp_path))


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
    if name == "full":
        # the reference's pipeline end to end (README.md:100-118): generate frames, hand them to the codec as [B, 32, n]
        from csm_hf_amd import CSMConfig, CSMModel
        from csm_hf_amd.synth import synth_state_dict, synth_context
        ccfg = CSMConfig.tiny()
        m = CSMModel(ccfg)
        m.load_state_dict(synth_state_dict(ccfg, seed=0, std=0.05))
        m = m.to("cuda:0").eval()
        ids, mask = synth_context(ccfg, 2, 3, 5, seed=1)
        frames = m.generate(ids.to("cuda:0"), mask.to("cuda:0"), max_new_frames=4, topk=1, stop_on_all_zeros=False)
        wav = dec.decode(frames.permute(0, 2, 1))
        assert wav.shape == (2, 1, 4 * 1920) and bool(torch.isfinite(wav).all())
        assert rel_max(wav.cpu(), MO.decode(sd, cfg, frames.permute(0, 2, 1).cpu())) < 1e-4
        m._drop_engine()
    # streaming: the chunks of a stream (1, 1, 2, 5, ... frames; the tiny window of 6 positions wraps many times) concatenate
    # to the one-shot decode -- K/V window, rotary positions and every convolution's left context carried in the handle
    Ts = 23 if name == "tiny" else 9
    c = torch.randint(0, cfg.codebook_size, (1, cfg.num_quantizers, Ts), generator=gen).to("cuda:0")
    whole = dec.decode(c)
    for chunks in ((1, 1, 2, 5, 3, 1, 4, 6), (Ts,), (2,) * 20):
        dec.stream_reset()
        parts, t = [], 0
        for n in chunks:
            n = min(n, Ts - t)
            if n <= 0:
                break
            parts.append(dec.stream_decode(c[0, :, t:t + n]))
            t += n
        assert t == Ts
        got = torch.cat(parts, dim=-1)
        # GEMMs of <= 16 rows (calls of up to 8 frames) run on the skinny-GEMM kernels (fp32 FMA order, not the MFMA chain's):
        # measured 2.2e-6 of the peak over 160 frames at the full shape; bitwise equal with option skinny_rows = 0 (test below)
        assert rel_max(got.cpu(), whole.cpu()) < 1e-5, chunks
    with pytest.raises(ValueError):
        dec.decode(torch.zeros(1, cfg.num_quantizers, 65, dtype=torch.long))          # beyond max_frames
    with pytest.raises(ValueError):
        dec.decode(torch.full((1, cfg.num_quantizers, 2), cfg.codebook_size, dtype=torch.long))
    dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "full"])
def test_stream_decode_small_calls_on_the_skinny_gemm(name):
    """GEMMs of few rows (streaming calls of a few frames, short one-shot decodes) run on the weight-streaming skinny GEMM
    (4.0 -> 1.0 ms per one-frame call at the kyutai/mimi shape).  Pinned three ways: against the ORACLE at the codec's 1e-4
    of the peak (the anchor), against the one-shot decode on the tile kernel at 1e-5 (fp32 summation order of the two GEMM
    kernels), and -- with the skinny path and the K split switched off (csm_mimi_set_option) -- the stream bitwise against the one-shot
    decode (one kernel and one summation order for every row count)."""
    from csm_hf_amd import MimiDecoder
    cfg = CASES[name]
    sd = synth_mimi_state_dict(cfg, seed=0)
    gen = torch.Generator().manual_seed(11)
    Ts = 40 if name == "tiny" else 20
    c = torch.randint(0, cfg.codebook_size, (1, cfg.num_quantizers, Ts), generator=gen)
    want = MO.decode(sd, cfg, c)
    cd = c.to("cuda:0")

    def stream(dec, chunks):
        dec.stream_reset()
        parts, t = [], 0
        for n in chunks:
            n = min(n, Ts - t)
            if n <= 0:
                break
            parts.append(dec.stream_decode(cd[0, :, t:t + n]).clone())
            t += n
        assert t == Ts
        return torch.cat(parts, dim=-1)

    fast = MimiDecoder(cfg, sd, "cuda:0", max_frames=64)
    plain = MimiDecoder(cfg, sd, "cuda:0", max_frames=64)
    plain.set_option("skinny_rows", 0)
    plain.set_option("splitk", 0)
    with pytest.raises(RuntimeError):
        plain.set_option("no_such_option", 1)
    whole = plain.decode(cd)                              # every GEMM on the 128 x 128 tile, K walked in one piece
    assert rel_max(whole.cpu(), want) < 1e-4
    assert rel_max(fast.decode(cd).cpu(), want) < 1e-4   # short one-shot decodes take the skinny path too
    differs = False
    for chunks in ((1,) * Ts, (2, 1) * Ts, (8, 3) * Ts):
        a, b = stream(fast, chunks), stream(plain, chunks)
        assert rel_max(a.cpu(), want) < 1e-4, chunks
        assert rel_max(a.cpu(), whole.cpu()) < 1e-5, chunks
        assert torch.equal(b, whole), chunks
        differs |= not torch.equal(a, b)
    assert differs                                        # the switch selects a different kernel: the two paths are not one
    fast.close()
    plain.close()


@pytest.mark.gpu
def test_streams_side_by_side_on_one_weight_set():
    """`MimiDecoder.new_stream()`: further decoders on the SAME device weights (no second copy), each with its own stream state.
    Three rows of a batch streamed frame by frame in interleaved order: every stream equals ITS one-shot decode bitwise on the
    plain tile, i.e. the handles share nothing mutable; closing the parent first keeps the children's weights alive."""
    from csm_hf_amd import MimiDecoder
    cfg = CASES["tiny"]
    sd = synth_mimi_state_dict(cfg, seed=0)
    gen = torch.Generator().manual_seed(5)
    T = 24
    codes = torch.randint(0, cfg.codebook_size, (3, cfg.num_quantizers, T), generator=gen).to("cuda:0")
    first = MimiDecoder(cfg, sd, "cuda:0", max_frames=64)
    decs = [first, first.new_stream(), first.new_stream()]
    assert all(d.packed is first.packed for d in decs)
    for d in decs:
        d.set_option("skinny_rows", 0)
        d.set_option("splitk", 0)
        d.stream_reset()
    whole = first.decode(codes)                            # [3, 1, T * spf], rows decoded one after the other
    parts = [[] for _ in decs]
    t = [0, 0, 0]
    order = [0, 1, 2, 2, 1, 0, 1, 1, 2, 0, 0, 2] * 8       # interleaved, unequal progress
    for b in order:
        if t[b] >= T:
            continue
        n = 1 + (t[b] + b) % 3
        n = min(n, T - t[b])
        parts[b].append(decs[b].stream_decode(codes[b, :, t[b]:t[b] + n]).clone())
        t[b] += n
    for b in range(3):
        while t[b] < T:
            parts[b].append(decs[b].stream_decode(codes[b, :, t[b]:t[b] + 1]).clone())
            t[b] += 1
        assert torch.equal(torch.cat(parts[b], dim=-1)[0], whole[b]), b
    want = MO.decode(sd, cfg, codes.cpu())
    assert rel_max(whole.cpu(), want) < 1e-4
    first.close()                                          # the children hold the packed tensors
    decs[1].stream_reset()
    again = decs[1].stream_decode(codes[1])
    assert torch.equal(again[0], whole[1])
    decs[1].close()
    decs[2].close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "full"])
def test_stream_group_equals_each_streams_one_shot_decode(name):
    """csm_mimi_streams_* (round 3): S streams advance in lockstep and every launch covers all of them.  Each stream's chunks
    must concatenate to the ONE-SHOT decode of its own sequence: against the oracle at the codec's 1e-4 of the peak and
    against the single-stream HIP decode at 1e-5 (the GEMMs see S times the rows and may take another of the three GEMM
    paths: fp32 summation order).  Ragged chunk sizes, the attention window wrapping (tiny: window 6), and one stream
    RESTARTED in the middle (a batch row taken over by a new utterance) while the others continue."""
    from csm_hf_amd import MimiDecoder
    cfg = CASES[name]
    sd = synth_mimi_state_dict(cfg, seed=0)
    gen = torch.Generator().manual_seed(17)
    S, T = (5, 36) if name == "tiny" else (4, 14)
    codes = torch.randint(0, cfg.codebook_size, (S, cfg.num_quantizers, T), generator=gen)
    new_row = torch.randint(0, cfg.codebook_size, (cfg.num_quantizers, T), generator=gen)   # the utterance that takes over stream 2
    dec = MimiDecoder(cfg, sd, "cuda:0", max_frames=64)
    cd = codes.to("cuda:0")
    whole = dec.decode(cd)                                                   # [S, 1, T * spf], single-stream passes
    want = MO.decode(sd, cfg, codes)
    assert rel_max(whole.cpu(), want) < 1e-4
    spf = cfg.samples_per_frame
    dec.streams_open(S)
    with pytest.raises(ValueError):
        dec.streams_decode(cd[:, :, :1].repeat(1, 1, 64 // S + 1))           # more frames per call than max_frames // S
    restart_at = T // 2
    parts, t, chunks = [], 0, (1, 3, 2, 1, 4)
    cur = cd.clone()
    i = 0
    restarted = False
    while t < T:
        n = min(chunks[i % len(chunks)], T - t, 64 // S)
        if not restarted and t >= restart_at:
            dec.streams_reset(2)                                             # stream 2 starts a new utterance at frame t
            cur[2, :, t:] = new_row[:, :T - t].to("cuda:0")
            restarted, t_restart = True, t
        parts.append(dec.streams_decode(cur[:, :, t:t + n]).clone())
        t += n
        i += 1
    got = torch.cat(parts, dim=-1)
    for s in range(S):
        if s == 2:
            continue
        assert rel_max(got[s].cpu(), want[s]) < 1e-4, s
        assert rel_max(got[s].cpu(), whole[s].cpu()) < 1e-5, s
    # stream 2: the old utterance up to the restart, then the new one from silence
    assert rel_max(got[2, :, :t_restart * spf].cpu(), want[2, :, :t_restart * spf]) < 1e-4
    fresh = MO.decode(sd, cfg, new_row[None, :, :T - t_restart])
    assert rel_max(got[2, :, t_restart * spf:].cpu(), fresh[0]) < 1e-4
    # a new group replaces the old one; all streams from silence again
    dec.streams_open(2)
    two = torch.cat([dec.streams_decode(cd[:2, :, a:a + 2]) for a in range(0, T - T % 2, 2)], dim=-1)
    assert rel_max(two.cpu(), want[:2, :, :two.shape[-1]]) < 1e-4
    dec.close()
