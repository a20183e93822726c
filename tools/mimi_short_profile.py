"""one-shot decode of [frames = 25] frames (kyutai/mimi shape), 45 calls: for rocprofv3 --kernel-trace --stats"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csm_hf_amd import MimiDecoder
from csm_hf_amd.mimi import MimiDecodeConfig, synth_mimi_state_dict
cfg = MimiDecodeConfig()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 25
dec = MimiDecoder(cfg, synth_mimi_state_dict(cfg, seed=0), "cuda:0", max_frames=max(64, T))
codes = torch.randint(0, cfg.codebook_size, (1, cfg.num_quantizers, T), generator=torch.Generator().manual_seed(1)).to("cuda:0")
for _ in range(5):
    dec.decode(codes)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40):
    dec.decode(codes)
torch.cuda.synchronize()
print(f"{T} frames one-shot: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per decode", flush=True)
