"""Stress of the weight streamer's end / give-up logic (VERDICT r5 item 1 a): N fresh engines (so every first call captures,
instantiates and uploads a graph and loads code objects), three generate() calls each, every call's statistics record kept when
anything gave up or the call took more than twice the median.  `python tools/streamer_stress.py [n_engines] [tiny|csm1b]`.
Output: JSON lines + a summary line; the round's record is profiles/r06_streamer_stress.txt."""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from csm_hf_amd import CSMConfig, CSMModel  # noqa: E402
from csm_hf_amd.synth import synth_state_dict, synth_context  # noqa: E402

DEV = "cuda:0"


def main():
    n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    which = sys.argv[2] if len(sys.argv) > 2 else "tiny"
    rows = []
    if which == "tiny":
        cfg = CSMConfig.tiny()
        sd = synth_state_dict(cfg, seed=0, std=0.05)
        frames, ctx = 12, (4, 6)
    else:
        cfg = CSMConfig()
        sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=DEV, bf16_representable=True)
        frames, ctx = 40, (16, 48)
    for i in range(n_eng):
        dtype = torch.bfloat16 if (which != "tiny" or i % 2 == 0) else torch.float32
        m = CSMModel(cfg)
        m.load_state_dict({k: v.to(dtype) for k, v in sd.items()} if which == "tiny" else sd)
        m = m.to(DEV).eval()
        ids, mask = synth_context(cfg, 1, ctx[0], ctx[1], seed=12)
        ids, mask = ids.to(DEV), mask.to(DEV)
        for call in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.generate(ids, mask, max_new_frames=frames, topk=1, stop_on_all_zeros=False)
            m._engine.sync()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            st = m._engine.prefetch_stats()
            rows.append({"engine": i, "call": call, "dtype": str(dtype), "call_ms": round(ms, 2), "stats": st})
        m._drop_engine()
        del m
    med = {c: statistics.median(r["call_ms"] for r in rows if r["call"] == c) for c in range(3)}
    odd = [r for r in rows if r["stats"]["gave_up"] or r["call_ms"] > 2 * med[r["call"]]]
    for r in odd:
        print(json.dumps(r))
    print(json.dumps({"model": which, "engines": n_eng, "calls": len(rows), "median_call_ms": med,
                      "calls_with_give_ups": sum(1 for r in rows if r["stats"]["gave_up"]),
                      "calls_retired_by_end_rule": sum(1 for r in rows if not r["stats"]["note"].endswith(" 0")),
                      "max_call_ms": max(r["call_ms"] for r in rows),
                      "disabled_engines": sum(1 for r in rows if r["call"] == 2 and r["stats"].get("health", {}).get("disabled", 0))}))


if __name__ == "__main__":
    main()
