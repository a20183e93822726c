"""kyutai/mimi shape: a stream decoded frame by frame against the one-shot decode of the same codes (distance relative to the
waveform's peak), and the latency of a one-frame call.  CSM_MIMI_SKINNY=0 puts every GEMM back on the 128 x 128 tile."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csm_hf_amd import MimiDecoder
from csm_hf_amd.mimi import MimiDecodeConfig, synth_mimi_state_dict
cfg = MimiDecodeConfig()
sd = synth_mimi_state_dict(cfg, seed=0)
dec = MimiDecoder(cfg, sd, "cuda:0", max_frames=200)
codes = torch.randint(0, cfg.codebook_size, (1, cfg.num_quantizers, 160), generator=torch.Generator().manual_seed(1)).to("cuda:0")
whole = dec.decode(codes)
dec.stream_reset()
parts = [dec.stream_decode(codes[0, :, t:t + 1]).clone() for t in range(160)]
got = torch.cat(parts, dim=-1)
peak = float(whole.abs().max())
print(f"CSM_MIMI_SKINNY={os.environ.get('CSM_MIMI_SKINNY', '1')}: "
      f"160 frames streamed one by one vs one-shot decode: max |diff| / peak = {float((got - whole).abs().max()) / peak:.3e}", flush=True)
dec.stream_reset()
for t in range(10):
    dec.stream_decode(codes[0, :, t:t + 1])
ts = []
for t in range(10, 150):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dec.stream_decode(codes[0, :, t:t + 1])
    ts.append(time.perf_counter() - t0)
ts.sort()
print(f"   one frame (80 ms of audio) per call: median {ts[len(ts) // 2] * 1e3:.3f} ms, min {ts[0] * 1e3:.3f} ms", flush=True)
