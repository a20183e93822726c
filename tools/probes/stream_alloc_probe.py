"""Where does the 15 % come from when generate() runs under a torch side stream?  A: everything on the default stream; B: weights allocated
on the default stream, engine created and run under a side stream; C: weights allocated under a side stream, engine on the default
stream; D: both under the side stream.  ms per frame-step (HIP events), B = 1, 512-frame context, 60 steps."""
import contextlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
dev = torch.device("cuda:0")
cfg = CSMConfig()
side = torch.cuda.Stream()
def ctx(on): return torch.cuda.stream(side) if on else contextlib.nullcontext()
for name, w_side, e_side in (("A", 0, 0), ("B", 0, 1), ("C", 1, 0), ("D", 1, 1), ("A", 0, 0)):
    with ctx(w_side):
        sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
        m = CSMModel(cfg); m.load_state_dict(sd); del sd
        m.kv_dtype = torch.float32
        torch.cuda.synchronize()
    ids, mask = synth_context(cfg, 1, 128, 384, seed=2)
    with ctx(e_side):
        eng = m._ensure_engine(1, 512 + 80, 80, 512)
        eng.reset(); eng.set_kv_start([0]); eng.prefill(ids, mask, want_outputs=False)
        s = eng.sampling(temperature=1.0, topk=1, seed=1)
        eng.generate(s, 10, True); eng.sync()
        eng.generate(s, 60, True); eng.sync()
        ms = eng.last_generate_ms() / 60
        st = eng.prefetch_stats()
        torch.cuda.synchronize()
    print(f"{name}: weights under side stream {w_side}, engine under side stream {e_side}: {ms:.4f} ms/step  streamer finished {st['finished']} gave_up {st['gave_up']} rot {st['xcd_rotation']}", flush=True)
    m._drop_engine(); del m, eng
    torch.cuda.empty_cache()
