show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); w=d['weight_streamer']; print(d['hip_event_ms_per_step'], w['health']['disabled'], w['note'][9:110])"; }
echo -n "good: "; python bench.py --steps 40 --warmup 5 --lean 2>/dev/null | show
echo -n "shared (between=3): "; CSM_EXP_DUMMIES_BETWEEN=3 python bench.py --steps 40 --warmup 5 --lean 2>/dev/null | show
echo -n "null-order bad: "; CSM_EXP_NO_NULL=1 python tools/probes/prio_bench.py 0 -- --steps 40 --warmup 5 --lean 2>/dev/null | show
echo -n "null-order fixed: "; python tools/probes/prio_bench.py 0 -- --steps 40 --warmup 5 --lean 2>/dev/null | show
echo -n "between=4: "; CSM_EXP_DUMMIES_BETWEEN=4 python bench.py --steps 40 --warmup 5 --lean 2>/dev/null | show
echo -n "between=1: "; CSM_EXP_DUMMIES_BETWEEN=1 python bench.py --steps 40 --warmup 5 --lean 2>/dev/null | show
