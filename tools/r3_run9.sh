#!/bin/bash
# planes pipeline on / off at the per-rank shapes of the 8-GPU configurations (cfg3: 32768 rows, cfg5: 125000 draws)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
O=gpurun_out/r3/run9.log
: > $O
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for b in 16384 24576 32768 40960; do
  for r in 1 2; do
  USFLOWS_AMD_PLANES=0 python bench.py --no-cpu-baseline --no-fast-mode --no-kernel-timing --batch $b --steps 20 2>/dev/null | p "B=$b fp32-activations" >> $O
  USFLOWS_AMD_PLANES=1 python bench.py --no-cpu-baseline --no-fast-mode --no-kernel-timing --batch $b --steps 20 2>/dev/null | p "B=$b planes" >> $O
  done
done
cat $O
