#!/bin/bash
# B = 1: KV splits of the backbone attention x split merge inside the launch (fuse_attn_combine = 2: at every batch size)
for ns in 0 32 16 8 4; do for f in 0 2; do
  echo -n "nsplit_backbone=$ns fuse_attn_combine=$f : "
  timeout 300 python bench.py --no-cpu-baseline --config4 0 --lean --steps 300 --opt nsplit_backbone=$ns --opt fuse_attn_combine=$f 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('ms/step %.4f  frames/s %.1f  parity %s' % (d['ms_per_step'], d['value'], d.get('parity', {}).get('equal_all')))"
done; done
