"""Probe for the config-5 MX-fp8 pin: how far is the engine's `prefill_precision = "mxfp8"` context from the
oracle with the same OCP-MX rounding wrapped around its backbone linears, by context length -- and how far are two runs of
THAT ORACLE from each other when the only difference is fp32-level noise in front of every quantiser (the sensitivity of the
simulated network itself: an instruction-level accumulation difference moves elements across e4m3 rounding steps).
usage (GPU box): python tools/mx_pin_probe.py [frames ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csm_hf_amd import CSMConfig, CSMModel            # noqa: E402
from csm_hf_amd.synth import synth_state_dict, synth_context   # noqa: E402
from oracle import csm_oracle as O, mx_sim as MX      # noqa: E402  (checker only)

DEV = "cuda:0"


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


class noisy_mx(MX.mx_linears):
    """mx_linears with a relative perturbation of `eps` (fp32 accumulation-order class) in front of the activation quantiser"""

    def __init__(self, oracle_module, weights, eps, seed):
        super().__init__(oracle_module, weights)
        self.eps, self.g = eps, torch.Generator().manual_seed(seed)

    def __enter__(self):
        self.orig = self.O.F.linear

        def lin(x, w, b=None):
            if w.data_ptr() not in self.keys:
                return self.orig(x, w, b)
            k = w.data_ptr()
            if k not in self.cache:
                self.cache[k] = MX.mx_round(w)
            xn = x * (1.0 + self.eps * torch.randn(x.shape, generator=self.g))
            return self.orig(MX.mx_round(xn), self.cache[k], b)
        self.O.F.linear = lin
        return self


def main():
    frames = [int(a) for a in sys.argv[1:]] or [64, 512, 2048]
    cfg = CSMConfig()
    sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=DEV, bf16_representable=True)
    m = CSMModel(cfg)
    m.load_state_dict(sd)
    m.eval()
    sdc = {k: v.float().cpu() for k, v in sd.items()}
    lin = [v for k, v in sdc.items() if k.startswith("backbone.layers.") and k.endswith("_proj.weight")]
    for S in frames:
        ids, mask = synth_context(cfg, 1, S // 8, S - S // 8, seed=5)
        m.prefill_precision = "exact"
        ex = m.forward(ids.to(DEV), mask.to(DEV), return_dict=True).last_hidden_state.float().cpu()
        m.prefill_precision = "mxfp8"
        ship = m.forward(ids.to(DEV), mask.to(DEV), return_dict=True).last_hidden_state.float().cpu()
        m._engine.set_option("prefill_bf16_attn", 0)
        strict = m.forward(ids.to(DEV), mask.to(DEV), return_dict=True).last_hidden_state.float().cpu()
        m._engine.set_option("prefill_bf16_attn", 1)
        with torch.inference_mode():
            with MX.mx_linears(O, lin):
                lh = O.forward(sdc, cfg, ids, mask)[0]
            with noisy_mx(O, lin, 2e-5, 1):
                lh_n1 = O.forward(sdc, cfg, ids, mask)[0]
            with noisy_mx(O, lin, 2e-7, 2):
                lh_n2 = O.forward(sdc, cfg, ids, mask)[0]
            lh_exact = O.forward(sdc, cfg, ids, mask)[0]
        print(f"S = {S:5d}: engine exact vs oracle {rel(ex, lh_exact):.2e} | engine mx strict vs oracle+MX {rel(strict, lh):.3e}, shipped "
              f"{rel(ship, lh):.3e} | oracle+MX vs itself with 2e-5 / 2e-7 relative noise before every quantiser {rel(lh_n1, lh):.3e} / "
              f"{rel(lh_n2, lh):.3e} | MX class: oracle+MX vs oracle exact {rel(lh, lh_exact):.3e}, engine mx vs engine exact {rel(strict, ex):.3e}",
              flush=True)
    m._drop_engine()


if __name__ == "__main__":
    main()
