import sys, time, torch
sys.path.insert(0, "/root/repo")
from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
dev = "cuda:0"
cfg = CSMConfig()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
m = CSMModel(cfg); m.load_state_dict(sd); del sd
ids, mask = synth_context(cfg, 1, 128, 384, seed=2)
for (ml, fr) in ((520, 4), (722, 209)):
    m._drop_engine()
    eng = m._ensure_engine(1, ml, fr, 512)
    for mode in (1, 0):
        eng.set_option("prefill_bf16", mode)
        for q in (0, 4):
            eng.set_option("prefill_splitk_qkv", q)
            ts = []
            for _ in range(6):
                eng.reset(); eng.set_kv_start([0]); torch.cuda.synchronize(); t0 = time.perf_counter()
                eng.prefill(ids, mask, want_outputs=False); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
            print(f"max_len {ml} frames {fr} bf16={mode} qkv_split<={q}: " + " ".join(f"{t:.2f}" for t in ts), flush=True)

# the sequence bench.py runs: engine (722, 209, 512), a 128-frame exact prefill, then 5 x bf16, then 5 x exact
m._drop_engine()
eng = m._ensure_engine(1, 722, 209, 512)
eng.reset(); eng.set_kv_start([0]); eng.prefill(ids[:, :128], mask[:, :128], want_outputs=False)
for mode in (1, 0):
    eng.set_option("prefill_bf16", mode)
    ts = []
    for _ in range(5):
        eng.reset(); eng.set_kv_start([0]); torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.prefill(ids, mask, want_outputs=False); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"max_len bench-sequence bf16={mode}: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
