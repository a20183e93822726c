"""Does the frame-step time depend on the engine's history in the process (engines created and destroyed before, an engine re-homed into a
larger one while the old one is alive)?  ms per frame-step of a 100-frame greedy generate at a 512-frame context, B = 1."""
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
m.kv_dtype = torch.float32
ids, mask = synth_context(cfg, 1, 128, 384, seed=2)
ids, mask = ids.to(dev), mask.to(dev)
ids_s, mask_s = ids[:, :64], mask[:, :64]

def big(tag):
    for _ in range(2):
        m.generate(ids, mask, max_new_frames=100, topk=1, stop_on_all_zeros=False)
    ms = m._engine.last_generate_ms() / 99
    st = m._engine.prefetch_stats()
    print(f"{tag}: {ms:.4f} ms/step   streamer disabled {st['health']['disabled']}  {st['note'][9:100]}", flush=True)

big("fresh engine")
m._drop_engine()
for i in range(3):
    m.generate(ids_s, mask_s, max_new_frames=4, topk=1, stop_on_all_zeros=False)   # a small engine ...
    big(f"after a small engine was outgrown ({i})")                                 # ... replaced by a larger one
# continuation growth: generate_frame stream that outgrows max_len (re-homed while the old engine is alive)
m._drop_engine()
out = m.generate(ids_s, mask_s, max_new_frames=4, topk=1, stop_on_all_zeros=False)
m2 = m
engines = []
big("after drop + small + big")
# several engines alive at once (e.g. two models in one process)
others = []
for i in range(3):
    o = CSMModel(cfg); o.load_state_dict(m.state_dict()); o.kv_dtype = torch.float32
    o.generate(ids_s, mask_s, max_new_frames=2, topk=1, stop_on_all_zeros=False)
    others.append(o)
    m._drop_engine()
    big(f"with {i + 1} other engine(s) alive, this one created last")
