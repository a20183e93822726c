"""Prefill of short contexts by the cap on K splits of the residual GEMMs (option prefill_splitk_max): time and distance of
the last hidden state to the cap-4 result, exact and bf16 activation modes (csm-1b, bf16 weights)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
dev = "cuda:0"
cfg = CSMConfig()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
m = CSMModel(cfg); m.load_state_dict(sd); del sd
eng = m._ensure_engine(1, 520, 4, 512)
for ctx in (32, 64, 128, 256, 512):
    ids, mask = synth_context(cfg, 1, ctx // 4, ctx - ctx // 4, seed=5)
    for mode in (0, 1):
        eng.set_option("prefill_bf16", mode)
        ref = None
        line = []
        for cap in (0, 2, 4, 8):
            eng.set_option("prefill_splitk_qkv", cap)
            ts = []
            for rep in range(8):
                eng.reset(); eng.set_kv_start([0])
                torch.cuda.synchronize(); t0 = time.perf_counter()
                eng.prefill(ids, mask, want_outputs=False); eng.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            o = eng.get_state()[0].double().cpu()
            if ref is None: ref = o
            line.append(f"qkv<={cap}: {min(ts):.2f} ms (rel {float((o - ref).norm() / ref.norm()):.1e})")
        print(f"ctx {ctx:4d} {'bf16 ' if mode else 'exact'}: " + "   ".join(line), flush=True)
