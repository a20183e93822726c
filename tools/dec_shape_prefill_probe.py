"""Lead for 33-64-row batches: the decoder's four linears (hidden 1024, ffn 8192) at 64 rows on the PREFILL-path kernels (exact
three-plane LDS-DMA tile with 64-row workgroups) -- a model whose BACKBONE has the decoder's layer shape, one 64-frame prefill; run
under `rocprofv3 --kernel-trace --stats` and read the GEMM durations against the decode chain's (profiles/r03_b64_rows64.txt:
QKV 9 + o_proj 8 + gate/up 15 + down_proj 15.5 us per decoder layer-pass at 64 rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
cfg = CSMConfig(backbone_config=dict(hidden_size=1024, intermediate_size=8192, num_hidden_layers=16, num_attention_heads=16,
                                     num_key_value_heads=4, head_dim=64, rms_norm_eps=1e-5, rope_theta=500000.0))
dev = torch.device("cuda:0")
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
m = CSMModel(cfg); m.load_state_dict(sd)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ids, mask = synth_context(cfg, 1, rows // 4, rows - rows // 4, seed=2)
for mode in ("exact", "bf16"):
    m.prefill_precision = mode
    eng = m._ensure_engine(1, 200, 4, 128)
    ts = []
    for _ in range(8):
        eng.reset(); eng.set_kv_start([0]); torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.prefill(ids, mask, want_outputs=False); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"decoder-shaped 16-layer stack, {rows} rows, mode={mode}: min {min(ts):.3f} ms = {min(ts) / 16 * 1e3:.1f} us per layer (all launches)", flush=True)
