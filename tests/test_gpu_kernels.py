"""GPU suite, per-kernel parity: every HIP kernel called through the C ABI against the CPU oracle's
formula for the same op (oracle/csm_oracle.py) on the same seeded inputs."""
import numpy as np
import pytest
import torch

from csm_hf_amd import CSMConfig
from csm_hf_amd.synth import synth_state_dict, synth_context, hash_uniform
from oracle import csm_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny():
    from csm_hf_amd.engine import Engine
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    eng = Engine(cfg, sd, "cuda:0", torch.float32, max_batch=4, max_len=128, max_frames=8, max_prefill_rows=256)
    yield cfg, sd, eng
    eng.close()


@pytest.fixture(scope="module")
def tiny_bf16():
    from csm_hf_amd.engine import Engine
    cfg = CSMConfig.tiny()
    sd = {k: v.to(torch.bfloat16) for k, v in synth_state_dict(cfg, seed=0, std=0.05).items()}
    eng = Engine(cfg, sd, "cuda:0", torch.bfloat16, max_batch=4, max_len=128, max_frames=8, max_prefill_rows=256,
                 kv_dtype=torch.bfloat16)
    yield cfg, sd, eng
    eng.close()


def rnd(name, *shape, scale=1.0):
    n = int(np.prod(shape))
    return (hash_uniform(name, n, 99) * scale).view(*shape)


def test_embed_sum(tiny, tiny_bf16):
    for cfg, sd, eng in (tiny, tiny_bf16):
        ids, mask = synth_context(cfg, 3, 4, 5, seed=4, eos_frame=True, tail_text=2)
        mask[2, :3] = 0     # a left-padded row
        ref, _ = O.embed_frames({k: v.float() for k, v in sd.items()}, cfg, ids, mask)
        out = eng.k_embed_sum(ids, mask).cpu().view_as(ref)
        torch.testing.assert_close(out, ref, atol=1e-6, rtol=1e-6)
        out2 = eng.k_embed_sum(ids, None).cpu().view_as(ref)   # mask=None: everything live
        ref2, _ = O.embed_frames({k: v.float() for k, v in sd.items()}, cfg, ids, None)
        torch.testing.assert_close(out2, ref2, atol=1e-6, rtol=1e-6)


def test_rmsnorm(tiny):
    _, _, eng = tiny
    for H in (256, 1024, 2048):
        x, w = rnd("x", 5, H, scale=3.0), rnd("w", H) + 1.0
        torch.testing.assert_close(eng.k_rmsnorm(x, w, 1e-5).cpu(), O.rmsnorm(x, w, 1e-5), atol=2e-6, rtol=2e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,K", [(64, 256), (2051, 1024), (3075, 2048), (256, 8192), (1024, 1024)])
def test_gemv(tiny, dtype, N, K):
    _, _, eng = tiny
    W = rnd("W", N, K, scale=0.05).to(dtype)
    for M in (1, 2, 3, 4, 7, 16, 21):
        x = rnd(f"x{M}", M, K)
        ref = x.double() @ W.double().T
        y = eng.k_gemv(W, x).cpu()
        torch.testing.assert_close(y.double(), ref, atol=2e-5, rtol=1e-5)
        ln = rnd("ln", K) + 1.0
        y2 = eng.k_gemv(W, x, ln=ln, eps=1e-5).cpu()
        ref2 = O.rmsnorm(x, ln, 1e-5).double() @ W.double().T
        torch.testing.assert_close(y2.double(), ref2, atol=5e-5, rtol=1e-5)


def test_gemv_rows_are_batch_invariant(tiny):
    """what makes batch-sharding exact: a row's result does not depend on which batch it sits in.
    fp32-FMA kernels (fp32 weights, or M == 1): bit-identical for any M.  Matrix-core kernel (bf16 weights,
    M >= 2): bit-identical across batch sizes/positions within that kernel, and within fp32 round-off of the
    M == 1 kernel (x is carried exactly as three bf16 parts)."""
    _, _, eng = tiny
    Wf = rnd("Wb", 512, 2048, scale=0.05)
    x = rnd("xb", 16, 2048)
    y4 = eng.k_gemv(Wf, x[:4]).cpu()
    for b in range(4):
        assert torch.equal(eng.k_gemv(Wf, x[b:b + 1]).cpu()[0], y4[b])
    W = Wf.to(torch.bfloat16)
    y16 = eng.k_gemv(W, x).cpu()
    y5 = eng.k_gemv(W, x[3:8]).cpu()
    assert torch.equal(y5, y16[3:8])
    y1 = torch.cat([eng.k_gemv(W, x[b:b + 1]).cpu() for b in range(16)])
    ref = x.double() @ W.double().T
    assert float((y16.double() - ref).abs().max()) < 2e-6 and float((y1.double() - ref).abs().max()) < 2e-6


@pytest.mark.parametrize("N,K", [(83, 256), (2051, 1024), (1024, 8192), (3075, 2048)])
def test_gemv_fragment_order_weights(tiny, N, K):
    """batched rows read the weights from the 16-row x 32-k fragment-order copy (tile16_kernel); the row-major variant
    of the same matrix-core kernel must agree to fp32 summation order, ragged last tiles (N % 16 != 0) included."""
    from csm_hf_amd.engine import quantize_fp8_rows
    _, _, eng = tiny
    Wb = rnd("Wt", N, K, scale=0.05).to(torch.bfloat16)
    q, sc = quantize_fp8_rows(rnd("Wt8", N, K, scale=0.05))
    ln = rnd("lnt", K) + 1.0
    try:
        for M in (2, 5, 16, 19):
            x = rnd(f"xt{M}", M, K)
            outs = {}
            for tile in (1, 0):
                eng.set_option("tile_weights", tile)
                outs[tile] = (eng.k_gemv(Wb, x).cpu(), eng.k_gemv(Wb, x, ln=ln, eps=1e-5).cpu(),
                              eng.k_gemv(q, x, scale=sc).cpu())
            for a, b in zip(outs[1], outs[0]):
                torch.testing.assert_close(a, b, atol=3e-6, rtol=3e-6)
            torch.testing.assert_close(outs[1][0].double(), x.double() @ Wb.double().T, atol=2e-5, rtol=1e-5)
    finally:
        eng.set_option("tile_weights", 1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,N,K", [(1, 128, 32), (70, 384, 256), (257, 256, 512), (512, 1024, 256)])
def test_gemm_prefill(tiny, dtype, R, N, K):
    _, _, eng = tiny
    W = rnd("Wg", N, K, scale=0.05).to(dtype)
    A = rnd("Ag", R, K)
    ref = A.double() @ W.double().T
    torch.testing.assert_close(eng.k_gemm(W, A).cpu().double(), ref, atol=2e-5, rtol=1e-5)


def test_gemm_transpose_detecting(tiny):
    """identity activations against an ASYMMETRIC weight: catches a row<->col swap in the MFMA C layout."""
    _, _, eng = tiny
    W = torch.arange(128 * 128, dtype=torch.float32).view(128, 128) / 1000.0
    out = eng.k_gemm(W, torch.eye(128)).cpu()
    torch.testing.assert_close(out, W.T.contiguous(), atol=0, rtol=0)
    Wb = (torch.arange(128 * 128, dtype=torch.float32).view(128, 128) % 251 - 125).to(torch.bfloat16)   # bf16 x3 kernel
    out = eng.k_gemm(Wb, torch.eye(128)).cpu()
    torch.testing.assert_close(out, Wb.float().T.contiguous(), atol=0, rtol=0)
    A = rnd("Asym", 200, 128) * 3.0                                 # exactness of the 3-way split on non-trivial A
    ref = A.double() @ Wb.double().T
    assert float((eng.k_gemm(Wb, A).cpu().double() - ref).abs().max()) < 1e-3 * 1e-2 * float(ref.abs().max()) + 1e-4


def test_sampler_against_reference_vectors(tiny, gold):
    _, _, eng = tiny
    g = gold("sampler")
    logits, noise = torch.from_numpy(g["logits"]), torch.from_numpy(g["noise"])
    for topk in (1, 50):
        for T in (0.7, 1.0):
            idx = eng.k_sample(logits, topk, T, noise=noise).cpu().numpy()
            want = g[f"idx_k{topk}_T{T}"]
            if topk == 1:
                # greedy: lowest index among exact ties (documented deviation); rows 0-7 have no ties
                assert np.array_equal(idx[:8], want[:8])
                lg = g["logits"]
                assert np.all(lg[np.arange(16), idx] == lg.max(-1))
                assert np.array_equal(idx, lg.argmax(-1))
            else:
                assert np.array_equal(idx, want), (topk, T)
    # temperature == 0 is argmax
    assert np.array_equal(eng.k_sample(logits, 50, 0.0).cpu().numpy(), g["logits"].argmax(-1))


def test_sampler_topk_edges_and_distribution(tiny):
    _, _, eng = tiny
    V = 2051
    logits = rnd("lg", 4, V, scale=2.0)
    noise = torch.empty(4, V).exponential_(1, generator=torch.Generator().manual_seed(1))
    for k in (1, 2, 50, 256, 257, V):
        want = O.sample_topk(logits, k, 0.9, noise).squeeze(-1)
        if k == 1:
            want = logits.argmax(-1)
        assert torch.equal(eng.k_sample(logits, k, 0.9, noise=noise).cpu().long(), want.long()), k
    # degenerate value distributions (the fallback paths of the k-th value search): a constant row, a row whose
    # top 600 entries are exactly tied (more than one histogram bin can resolve), a two-valued row, huge outliers
    deg = rnd("lgd", 5, V, scale=2.0)
    deg[0] = 1.25
    deg[1, torch.randperm(V, generator=torch.Generator().manual_seed(3))[:600]] = 7.0
    deg[2] = torch.where(torch.arange(V) % 3 == 0, torch.tensor(2.0), torch.tensor(-2.0))
    deg[3, 5] = 1e30
    deg[3, 77] = -1e30
    deg[4, :300] = deg[4, 300]          # 301 equal entries somewhere in the middle of the range
    noise5 = torch.empty(5, V).exponential_(1, generator=torch.Generator().manual_seed(2))
    for k in (2, 50, 300, 700):
        want = O.sample_topk(deg, k, 0.9, noise5).squeeze(-1)
        assert torch.equal(eng.k_sample(deg, k, 0.9, noise=noise5).cpu().long(), want.long()), k
    with pytest.raises(RuntimeError):
        eng.k_sample(logits, V + 1, 1.0)
    # device Philox race: empirical distribution ~ softmax over the top-k survivors
    lg = torch.tensor([[2.0, 1.0, 0.0, -1.0] + [-30.0] * 60]).repeat(4096, 1)
    idx = eng.k_sample(lg, 3, 1.0, seed=123).cpu().long()
    counts = torch.bincount(idx, minlength=4).float()[:4] / 4096
    want = torch.softmax(torch.tensor([2.0, 1.0, 0.0]), 0)
    assert counts[3] == 0 and torch.allclose(counts[:3], want, atol=0.03)


@pytest.mark.parametrize("which", [0, 1])
def test_rope_scatter_and_attention(tiny, tiny_bf16, which):
    """RoPE + KV append + decode attention vs the oracle's llama3 RoPE / SDPA formulas, incl. a causal
    multi-row pass (prefill semantics), split-KV combine and a left-padded sequence."""
    for cfg, sd, eng in (tiny, tiny_bf16):
        lc = cfg.decoder_config if which else cfg.backbone_config
        nq, nkv, hd = lc.num_attention_heads, lc.num_key_value_heads, lc.head_dim
        L = 32 if which else 100
        B = 2
        qkv = rnd(f"qkv{which}", B * L, (nq + 2 * nkv) * hd)
        row_seq = torch.arange(B).repeat_interleave(L)
        row_pos = torch.arange(L).repeat(B)
        q_rot = eng.k_rope_scatter(which, 1, qkv, row_seq, row_pos).cpu()
        # oracle
        x = qkv.view(B, L, nq + 2 * nkv, hd)
        q, k, v = x[:, :, :nq].transpose(1, 2), x[:, :, nq:nq + nkv].transpose(1, 2), x[:, :, nq + nkv:].transpose(1, 2)
        inv = O.llama3_inv_freq(hd, lc.rope_theta, lc.rope_scaling)
        cos, sin = O.rope_cos_sin(inv, torch.arange(L)[None], torch.float32)
        qr, kr = O.apply_rope(q, k, cos, sin)
        torch.testing.assert_close(q_rot.view(B, L, nq, hd).transpose(1, 2), qr * hd ** -0.5, atol=2e-6, rtol=2e-6)
        if eng.packed and sd["projection.weight"].dtype == torch.bfloat16:
            kr, v = kr.to(torch.bfloat16).float(), v.to(torch.bfloat16).float()   # bf16 KV storage
        ref = torch.nn.functional.scaled_dot_product_attention(qr, kr, v, is_causal=True, enable_gqa=True)
        ref = ref.transpose(1, 2).reshape(B * L, nq * hd)
        out = eng.k_attn(which, 1, q_rot, row_seq, row_pos).cpu()
        torch.testing.assert_close(out, ref, atol=3e-5, rtol=3e-5)
        if which == 0:
            # decode rows with split-KV: last position of each sequence, nsplit 1 vs 4 vs 16
            rs, rp = torch.arange(B), torch.full((B,), L - 1)
            ql = q_rot.view(B, L, -1)[:, -1]
            o1 = eng.k_attn(0, 1, ql, rs, rp, nsplit=1).cpu()
            torch.testing.assert_close(o1, ref.view(B, L, -1)[:, -1], atol=3e-5, rtol=3e-5)
            for ns in (4, 16):
                torch.testing.assert_close(eng.k_attn(0, 1, ql, rs, rp, nsplit=ns).cpu(), o1, atol=2e-6, rtol=2e-6)
            # left padding: keys < kv_start are invisible
            eng.set_kv_start([0, 37, 0, 0])
            op = eng.k_attn(0, 1, ql, rs, rp, nsplit=4).cpu()
            eng.set_kv_start([0, 0, 0, 0])
            refp = torch.nn.functional.scaled_dot_product_attention(qr[1:2, :, -1:], kr[1:2, :, 37:], v[1:2, :, 37:], enable_gqa=True)
            torch.testing.assert_close(op[1], refp.reshape(-1), atol=3e-5, rtol=3e-5)
            torch.testing.assert_close(op[0], o1[0], atol=2e-6, rtol=2e-6)


@pytest.mark.parametrize("N,K", [(64, 256), (2051, 1024), (3075, 2048), (1024, 8192), (16384, 1024)])
def test_gemv_and_gemm_fp8_weights(tiny, N, K):
    """e4m3fn weights + per-row scales (BASELINE config 5): every kernel family against the dequantised fp64
    product -- the widening is exact, so the only error is fp32 summation order."""
    from csm_hf_amd.engine import quantize_fp8_rows, dequantize_fp8_rows
    _, _, eng = tiny
    W = rnd("W8", N, K, scale=0.05)
    q, s = quantize_fp8_rows(W)
    Wd = dequantize_fp8_rows(q, s).double()
    assert float((Wd - W.double()).abs().max()) < 0.05 * 0.07          # e4m3: <= 2^-4 relative per weight
    for M in (1, 3, 16):
        x = rnd(f"x8{M}", M, K)
        torch.testing.assert_close(eng.k_gemv(q, x, scale=s).cpu().double(), x.double() @ Wd.T, atol=3e-5, rtol=1e-5)
        ln = rnd("ln8", K) + 1.0
        ref = O.rmsnorm(x, ln, 1e-5).double() @ Wd.T
        torch.testing.assert_close(eng.k_gemv(q, x, ln=ln, eps=1e-5, scale=s).cpu().double(), ref, atol=6e-5, rtol=1e-5)
    if N % 128 == 0 and K % 32 == 0:
        A = rnd("A8", 70, K)
        torch.testing.assert_close(eng.k_gemm(q, A, scale=s).cpu().double(), A.double() @ Wd.T, atol=3e-5, rtol=1e-5)
