#!/bin/bash
# same-box A/B of the headline between the in-tree library and another build (USFLOWS_AMD_LIB): alternating runs
# usage (on the GPU box, from the repo root): tools/ab_headline.sh tools/libusflows_prev.so [rounds]
other=$1; rounds=${2:-3}
for i in $(seq $rounds); do
  for lib in "" "$other"; do
    if [ -n "$lib" ]; then export USFLOWS_AMD_LIB=$lib; else unset USFLOWS_AMD_LIB; fi
    python bench.py --no-also --no-cpu-baseline --no-fast-mode --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-in-tree}', o['ms_per_step'], o['roofline']['avg_launch_ms'], o['roofline']['all_kernels_ms_per_step'])"
  done
done
