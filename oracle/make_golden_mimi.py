"""Golden vectors for Mimi decode (row f-2) by RUNNING transformers' MimiModel -- the implementation of the codec's published
architecture that is in the image (the reference's own `moshi` package is not).  Build container only.

    python oracle/make_golden_mimi.py

Loads the seeded synthetic decode-path checkpoint of `csm_hf_amd.mimi.synth_mimi_state_dict` into `transformers.MimiModel`
(strict on the decode path's keys), decodes seeded codes, checks the oracle (oracle/mimi_oracle.py) against it and stores
codes + waveform under tests/golden/.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformers import MimiConfig, MimiModel  # noqa: E402

from csm_hf_amd.mimi import MimiDecodeConfig, synth_mimi_state_dict, mimi_state_dict_spec  # noqa: E402
from oracle import mimi_oracle as MO  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def hf_config(cfg: MimiDecodeConfig) -> MimiConfig:
    return MimiConfig(num_quantizers=cfg.num_quantizers, num_semantic_quantizers=cfg.num_semantic_quantizers,
                      codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim, hidden_size=cfg.hidden_size,
                      vector_quantization_hidden_dimension=cfg.codebook_dim,
                      num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                      num_key_value_heads=cfg.num_attention_heads, head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size,
                      sliding_window=cfg.sliding_window, upsampling_ratios=list(cfg.upsampling_ratios), num_filters=cfg.num_filters,
                      kernel_size=cfg.kernel_size, last_kernel_size=cfg.last_kernel_size, residual_kernel_size=cfg.residual_kernel_size,
                      compress=cfg.compress, norm_eps=cfg.norm_eps, upsample_groups=cfg.hidden_size)


def main():
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    # full_long: the kyutai/mimi shape PAST its attention window -- sliding_window = 250 transformer positions, two per
    # frame (the codes are upsampled 12.5 -> 25 Hz ahead of the transformer), i.e. 125 frames: 150 frames wrap it; two sequences
    for name, cfg, B, T in (("tiny", MimiDecodeConfig.tiny(), 2, 9), ("full", MimiDecodeConfig(), 1, 12),
                            ("full_long", MimiDecodeConfig(), 2, 150)):
        if only and name not in only:
            continue
        t0 = time.time()
        sd = synth_mimi_state_dict(cfg, seed=0)
        model = MimiModel(hf_config(cfg)).eval()
        missing, unexpected = model.load_state_dict(sd, strict=False)
        want = {k for k, _, _ in mimi_state_dict_spec(cfg)}
        assert not unexpected and not (want & set(missing)), (unexpected, want & set(missing))
        g = torch.Generator().manual_seed(11)
        codes = torch.randint(0, cfg.codebook_size, (B, cfg.num_quantizers, T), generator=g)
        with torch.no_grad():
            ref = model.decode(codes).audio_values
        mine = MO.decode(sd, cfg, codes)
        err = float((mine - ref).abs().max()) / float(ref.abs().max())
        assert ref.shape == (B, 1, T * cfg.samples_per_frame) and err < 1e-4, (ref.shape, err)
        if name == "full_long":
            # 2 x 150 x 1920 fp32 samples are 2.3 MB: keep sequence 0 whole and every 8th sample of sequence 1
            np.savez_compressed(os.path.join(GOLD, f"mimi_{name}.npz"), codes=codes.numpy(), audio0=ref[0].numpy().astype(np.float32),
                                audio1_every8=ref[1, :, ::8].numpy().astype(np.float32), peak=np.float32(ref.abs().max()),
                                oracle_max_rel_err=np.float32(err))
        else:
            np.savez_compressed(os.path.join(GOLD, f"mimi_{name}.npz"), codes=codes.numpy(), audio=ref.numpy().astype(np.float32),
                                oracle_max_rel_err=np.float32(err))
        print(f"[golden] mimi_{name}: {tuple(ref.shape)} samples, |audio| max {float(ref.abs().max()):.3f}, rms {float(ref.pow(2).mean().sqrt()):.3f}, "
              f"oracle vs transformers max rel err {err:.2e} ({time.time() - t0:.1f}s)", flush=True)


if __name__ == "__main__":
    main()
