#!/bin/bash
# A/B of one engine option on bench.py: bash tools/ab_opt.sh <option> "<values>" "<batches>" [steps]   (B = 1 runs include the parity check)
OPT=$1; VALS=$2; BATCHES=${3:-1}; STEPS=${4:-200}
for b in $BATCHES; do for rep in 1 2; do for v in $VALS; do
  echo -n "batch $b $OPT=$v : "
  timeout 300 python bench.py --no-cpu-baseline --config4 0 --lean --batch $b --steps $STEPS --opt $OPT=$v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('ms/step %.4f  frames/s %.1f  checksum %s parity %s' % (d['ms_per_step'], d['value'], d['tokens_checksum_per_rank'], d.get('parity', {}).get('equal_all')))"
done; done; done
