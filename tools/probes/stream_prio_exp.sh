run() { python tools/probes/prio_bench.py $1 -- --steps 100 --warmup 10 --lean 2>/tmp/err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['hip_event_ms_per_step'])"; }
echo -n "default (own stream, greatest): "; CSM_EXP_PRINT=1 run 99; grep "csm\]" /tmp/err | head -1
echo -n "own stream prio 0: "; CSM_EXP_OWN_PRIO=0 run 99
echo -n "own stream prio least(1?): "; CSM_EXP_OWN_PRIO=1 run 99
echo -n "side torch stream prio 0: "; run 0
echo -n "side torch ctx but own stream: "; CSM_EXP_OWN_STREAM=1 run 0
echo -n "side torch stream, s2 prio 0: "; CSM_EXP_S2_PRIO=0 run 0
echo -n "side torch stream prio -1, s2 prio 0: "; CSM_EXP_S2_PRIO=0 run -1
echo -n "default, s2 prio 0: "; CSM_EXP_S2_PRIO=0 run 99
echo -n "default, s2 prio -1: "; CSM_EXP_S2_PRIO=-1 run 99
