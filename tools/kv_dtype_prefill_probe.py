"""K/V cache dtype against the 2 048-frame prefill (bf16 / mxfp8 modes): python tools/kv_dtype_prefill_probe.py
Measured (one box): fp32 cache 5.91 / 4.63 ms, bf16 cache 5.87 / 4.54 ms -- the context attention stages K/V to bf16 either way."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
cfg = CSMConfig(); dev = torch.device("cuda:0")
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
ids, mask = synth_context(cfg, 1, 512, 1536, seed=2)
for kvd in (torch.float32, torch.bfloat16):
    m = CSMModel(cfg); m.load_state_dict(sd); m.kv_dtype = kvd
    eng = m._ensure_engine(1, 2056, 4, 2048)
    for mode in (1, 2):
        eng.set_option("prefill_bf16", 1)
        if mode == 2 and not eng.has_mx: eng.enable_mx(m.state_dict())
        eng.set_option("prefill_mx", 1 if mode == 2 else 0)
        ts = []
        for _ in range(8):
            eng.reset(); eng.set_kv_start([0]); torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.prefill(ids, mask, want_outputs=False); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        print(kvd, ("bf16", "mxfp8")[mode - 1], f"min {min(ts):.2f} ms", flush=True)
    m._drop_engine(); del m
