#!/bin/bash
# frame-step time by batch size (csm-1b bf16, 512-frame context, greedy), exact and decode_precision = bf16
for b in ${BATCHES:-1 2 4 8 16 24 32 48 64 96 128}; do for o in 0 1; do
  [ $b = 1 ] && [ $o = 1 ] && continue
  echo -n "batch $b decode_bf16=$o : "
  timeout 300 python bench.py --no-cpu-baseline --config4 0 --lean --batch $b --steps ${STEPS:-60} --opt decode_bf16=$o 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('ms/step %.4f  frames/s %.1f  frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))"
done; done
