#!/usr/bin/env python
"""Stage-by-stage run of the path for use under `rocprofv3 --pmc ...` when a counter pass crashes: every stage
prints a line to stderr after it has completed.  usage: python tools/pmc_probe.py [tiny|1b]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csm_hf_amd import CSMConfig  # noqa: E402
from csm_hf_amd.engine import Engine  # noqa: E402
from csm_hf_amd.synth import synth_context, synth_state_dict  # noqa: E402


def say(*a):
    print(*a, file=sys.stderr, flush=True)


big = len(sys.argv) > 1 and sys.argv[1] == "1b"
cfg = CSMConfig() if big else CSMConfig.tiny()
dev = torch.device("cuda:0")
sd = synth_state_dict(cfg, seed=1, dtype=torch.bfloat16, device=dev)
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
eng = Engine(cfg, sd, dev, dtype=torch.bfloat16, max_batch=1, max_len=int(kw.get("maxlen", 1024)),
             max_frames=int(kw.get("maxframes", 64)), max_prefill_rows=int(kw.get("rows", 4096)))
if "kvstart" in sys.argv:
    eng.set_kv_start([0])
del sd
say("engine ok")
for T in (8, 128, 512):
    ids, mask = synth_context(cfg, 1, T // 4, T - T // 4, seed=3)
    for opt in (0, 1):
        eng.set_option("flash_prefill", opt)
        eng.reset()
        eng.prefill(ids.to(dev), mask.to(dev), want_outputs=False)
        torch.cuda.synchronize()
        say("prefill flash=%d T=%d ok" % (opt, T))
s = eng.sampling(topk=1, temperature=1.0)
if "graphfirst" not in sys.argv:
    eng.generate(s, 2, use_graph=False)
    torch.cuda.synchronize()
    say("generate eager ok")
eng.generate(s, int(kw.get("n", 4)), use_graph="nograph" not in sys.argv)
torch.cuda.synchronize()
say("generate graph ok")
us, wb = eng.bench_gemv(4096, 1024, M=1, norm=True, epi=2)
say("bench_gemv ok")
say("done")
