"""Reproduction of the round-5 driver failure of tests/test_gpu_round2.py::test_weight_streamer_is_transparent[bf16]
(`gave_up 256, finished 0, launches_counted 2724`): the tiny model's generate() at several frame counts, the streamer's
whole statistics record after each call (un-truncated, with the stop record) and the call's wall time.  Runs against
whichever tree it is started from (`python tools/streamer_repro.py` in the repo root, or in a worktree of an older
commit), so the round-5 and round-6 libraries can be compared on ONE box.  Output: one JSON line per call."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from csm_hf_amd import CSMConfig, CSMModel  # noqa: E402
from csm_hf_amd.synth import synth_state_dict, synth_context  # noqa: E402

DEV = "cuda:0"


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "head"
    for dtype in (torch.float32, torch.bfloat16):
        cfg = CSMConfig.tiny()
        sd = synth_state_dict(cfg, seed=0, std=0.05)
        m = CSMModel(cfg)
        m.load_state_dict({k: v.to(dtype) for k, v in sd.items()})
        m = m.to(DEV).eval()
        ids, mask = synth_context(cfg, 1, 4, 6, seed=12)
        ids, mask = ids.to(DEV), mask.to(DEV)
        for n in (12, 12, 12, 40, 40, 100, 100, 12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.generate(ids, mask, max_new_frames=n, topk=1, stop_on_all_zeros=False)
            m._engine.sync()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            st = m._engine.prefetch_stats()
            print(json.dumps({"tree": tag, "dtype": str(dtype), "frames": n, "call_ms": round(ms, 2), "stats": st}), flush=True)
        m._drop_engine()


if __name__ == "__main__":
    main()
