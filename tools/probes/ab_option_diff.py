"""A/B of an engine option in both KV modes: max |logit diff| and last_h rel-L2 between option = 1 and option = 0, teacher-forced.
python tools/probes/ab_option_diff.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
DEV = "cuda:0"
cfg = CSMConfig()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=DEV, bf16_representable=True)
m = CSMModel(cfg); m.load_state_dict(sd); del sd; m.eval()

def traced(ids, mask, n, forced, options):
    B, T = ids.shape[:2]
    eng = m._ensure_engine(B, T + n + 1, max(n, 1), B * T)
    for k, v in options.items(): eng.set_option(k, v)
    eng.reset(); eng.set_kv_start(m._kv_starts(mask, B, T))
    lt = torch.zeros(eng.max_frames, B, eng.C, eng.V, dtype=torch.float32, device=DEV)
    ht = torch.zeros(eng.max_frames, B, eng.Hb, dtype=torch.float32, device=DEV)
    fz = torch.zeros(B, eng.max_frames, eng.C, dtype=torch.int64, device=DEV); fz[:, :n] = forced.to(DEV)
    lh, _ = eng.prefill(ids, mask); ht[0] = lh
    eng.generate(eng.sampling(temperature=1.0, topk=1, seed=7, forced=fz, logits_trace=lt, last_h_trace=ht), n, True)
    eng.sync()
    return lt[:n].cpu(), ht[:n].cpu()

for name, B, T, n in (("oproj_combine", 1, 300, 4), ("attn_gqa_wide", 40, 64, 3), ("attn_oproj_gqa", 1, 200, 4), ("fuse_attn_combine", 5, 300, 3)):
    ids, mask = synth_context(cfg, B, T // 4, T - T // 4, seed=33)
    forced = torch.randint(0, cfg.audio_vocab_size, (B, n, cfg.audio_num_codebooks), generator=torch.Generator().manual_seed(6))
    for kv in (torch.float32, torch.bfloat16):
        m.reset_caches(); m.kv_dtype = kv
        a = traced(ids, mask, n, forced, {name: 1}); b = traced(ids, mask, n, forced, {name: 0})
        m._engine.set_option(name, 1)
        d = (a[0] - b[0]).abs()
        per_frame = [float(d[f].max()) for f in range(n)]
        c0 = float(d[:, :, 0].max())
        print(name, B, T, str(kv), "logits |diff| per frame", per_frame, "codebook-0", c0, "last_h rel", float((a[1]-b[1]).norm()/b[1].norm()), flush=True)
