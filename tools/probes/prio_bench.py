"""A/B: the frame-step graph replayed on a HIGH-priority HIP stream (torch.cuda.Stream(priority=-1) as the current stream while the
engine is created) against the default stream.  usage: python tools/probes/prio_bench.py [prio] -- <bench.py flags>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

prio = int(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[3:]
import bench  # noqa: E402

if prio == 98:      # default stream stays current, but torch's side-stream pool exists (32 streams per priority are created at the first request)
    _s = torch.cuda.Stream()
    bench.main()
elif prio == 97:    # two pools
    _s = torch.cuda.Stream(); _t = torch.cuda.Stream(priority=-1)
    bench.main()
elif prio == 99:
    bench.main()
else:
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    print("priority range", lo, hi, file=sys.stderr)
    s = torch.cuda.Stream(priority=prio)
    with torch.cuda.stream(s):
        bench.main()
