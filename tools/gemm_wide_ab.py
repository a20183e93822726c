"""A/B of the prefill GEMM variants in prefill_precision = bf16 (square tile | wide tile, 1 or 4 weight-fragment sets,\nwith / without the per-workgroup k rotation): time and distance to the square-tile result per context length."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
dev = "cuda:0"
cfg = CSMConfig()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
m = CSMModel(cfg); m.load_state_dict(sd); del sd
for ctx in (512, 1024, 2048):
    ids, mask = synth_context(cfg, 1, ctx // 4, ctx - ctx // 4, seed=5)
    m._drop_engine()
    eng = m._ensure_engine(1, ctx + 8, 4, ctx)
    eng.set_option("prefill_bf16", 1)
    ref = None
    for opts in ({"gemm_wide": 0}, {"gemm_wide": 1, "gemm_wide_depth": 1, "gemm_wide_krot": 0}, {"gemm_wide_krot": 1}, {"gemm_wide_depth": 4, "gemm_wide_krot": 0}, {"gemm_wide_krot": 1}):
        for k, v in opts.items(): eng.set_option(k, v)
        ts = []
        for rep in range(6):
            eng.reset(); eng.set_kv_start([0])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.prefill(ids, mask, want_outputs=False); eng.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        o = eng.get_state()[0].double().cpu()
        if ref is None: ref = o
        print(ctx, opts, f"min {min(ts):.2f} ms", "rel vs square", float((o - ref).norm() / ref.norm()), flush=True)
