#!/bin/bash
# p = 1 loss and training step at <points> with the default and the forced cooperative kernel (SHW_P1_KERNEL=coop)
for PTS in "$@"; do
for round in 1 2; do
for k in default coop; do
  if [ "$k" = default ]; then unset SHW_P1_KERNEL; else export SHW_P1_KERNEL=coop; fi
  a=$(python bench.py --p 1 --steps 100 --warmup 50 --points $PTS --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('%.4f parity %.1e' % (d['ms_per_step'], d['parity_rel_err']))")
  b=$(python bench.py --mode train --p 1 --steps 100 --warmup 50 --points $PTS 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('%.4f' % d['ms_per_step'])")
  echo "$k N=$PTS p=1 loss $a train $b"
done; done; done
