"""The same engine created, measured and dropped six times in one process (B = 1, 512-frame context, 60 frame-steps after 10): does the
frame-step time depend on which incarnation it is?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
dev = "cuda:0"
cfg = CSMConfig()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
m = CSMModel(cfg); m.load_state_dict(sd); del sd
ids, mask = synth_context(cfg, 1, 128, 384, seed=2)
for i in range(8):
    m.kv_dtype = torch.float32 if (i % 2 == 0 or len(sys.argv) < 2) else torch.bfloat16
    eng = m._ensure_engine(1, 512 + 72, 71, 512)
    eng.reset(); eng.set_kv_start([0]); eng.prefill(ids, mask, want_outputs=False)
    s = eng.sampling(temperature=1.0, topk=1, seed=1)
    eng.generate(s, 10, True); eng.sync()
    ms = []
    for _ in range(3):
        eng.generate(s, 20, True); eng.sync(); ms.append(eng.last_generate_ms() / 20)
    st = eng.prefetch_stats()
    print(f"incarnation {i} kv {m.kv_dtype}: " + " ".join(f"{x:.4f}" for x in ms) + f" ms/step  rot {st['xcd_rotation']}  late {st['skipped_late_sample']}  {st['note'][9:80]}", flush=True)
    m._drop_engine(); del eng
