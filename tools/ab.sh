#!/bin/bash
# usage: ab.sh label [bench args...]  -> one line
L=$1; shift
timeout 300 python bench.py --lean "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'], d['hip_event_ms_per_step'], d['roofline']['frac'], d.get('parity',{}).get('equal_all'))"
