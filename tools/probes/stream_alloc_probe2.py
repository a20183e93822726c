"""bisecting what makes bench.py slow under a torch side stream: probe D + pieces of bench.main, one at a time (argv[1] = bit mask)"""
import contextlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
bits = int(sys.argv[1])
dev = torch.device("cuda:0")
if bits & 32:
    _z = torch.zeros(1 << 20, device=dev) + 1; torch.cuda.synchronize()      # the null stream gets its hardware queue first
if bits & 64:
    _d = [torch.cuda.Stream() for _ in range(2)]
    for d in _d:
        with torch.cuda.stream(d): _y = torch.zeros(1 << 20, device=dev) + 1
    torch.cuda.synchronize()
side = torch.cuda.Stream(priority=0)
with torch.cuda.stream(side):
    if bits & 1: torch.cuda.set_device(dev)
    if bits & 2: bench.pin_to_gpu_numa_node(0)
    cfg = CSMConfig()
    ids, mask = synth_context(cfg, 1, 128, 384, seed=2)
    sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
    m = CSMModel(cfg); m.load_state_dict(sd); del sd
    m.kv_dtype = torch.float32
    if not (bits & 4): torch.cuda.synchronize()
    if bits & 128:      # the null stream gets its hardware queue AFTER the side stream but BEFORE the engine's streams
        with torch.cuda.stream(torch.cuda.default_stream()):
            _z = torch.zeros(1 << 20, device=dev) + 1
        torch.cuda.synchronize()
    eng = m._ensure_engine(1, 512 + 112, 111, 512)
    eng.reset(); eng.set_kv_start([0])
    if bits & 8:
        eng.prefill(ids[:, :128], mask[:, :128], want_outputs=False)
        eng.set_option("prefill_bf16", 1); bench.timed_prefill(eng, ids, mask, 1); eng.set_option("prefill_bf16", 0)
        bench.timed_prefill(eng, ids, mask, 1)
    else:
        eng.prefill(ids, mask, want_outputs=False)
    s = eng.sampling(temperature=1.0, topk=1, seed=1234)
    eng.generate(s, 10, True); eng.sync()
    if bits & 16: torch.cuda.synchronize()
    eng.generate(s, 100, True); eng.sync()
    ms = eng.last_generate_ms() / 100
    print(f"bits {bits}: {ms:.4f} ms/step", flush=True)
