"""kyutai/mimi shape: a stream decoded frame by frame against the one-shot decode of the same codes (distance relative to the
waveform's peak), and the latency of a one-frame call.  `python tools/mimi_stream_probe.py 0` puts every GEMM back on the 128 x 128 tile (options skinny_rows = 0, splitk = 0)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csm_hf_amd import MimiDecoder
from csm_hf_amd.mimi import MimiDecodeConfig, synth_mimi_state_dict
cfg = MimiDecodeConfig()
sd = synth_mimi_state_dict(cfg, seed=0)
dec = MimiDecoder(cfg, sd, "cuda:0", max_frames=200)
SKINNY = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dec.set_option("skinny_rows", SKINNY)
if SKINNY == 0:
    dec.set_option("splitk", 0)
codes = torch.randint(0, cfg.codebook_size, (1, cfg.num_quantizers, 160), generator=torch.Generator().manual_seed(1)).to("cuda:0")
whole = dec.decode(codes)
dec.stream_reset()
parts = [dec.stream_decode(codes[0, :, t:t + 1]).clone() for t in range(160)]
got = torch.cat(parts, dim=-1)
peak = float(whole.abs().max())
print(f"skinny_rows={SKINNY}: "
      f"160 frames streamed one by one vs one-shot decode: max |diff| / peak = {float((got - whole).abs().max()) / peak:.3e}", flush=True)
for T in (1, 2, 4, 8):
    dec.stream_reset()
    ts = []
    for i in range(160 // T):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dec.stream_decode(codes[0, :, i * T:(i + 1) * T])
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[len(ts) // 4:])
    print(f"   {T} frame(s) ({80 * T} ms of audio) per call: median {ts[len(ts) // 2] * 1e3:.3f} ms, min {ts[0] * 1e3:.3f} ms", flush=True)
