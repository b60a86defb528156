#!/bin/bash
# round 3, experiment 5: SLP packing on / off in the 32-row-wave coupling kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp5.log
: > $O
for r in 1 2 3; do for v in 0 1; do echo "-- v$v (1 = -fno-slp-vectorize)" >> $O; timeout -k 10 120 tools/exp_cplanes_v$v 65536 3 2 2>&1 | grep -v "^16-row kernel vs" >> $O || echo "rc $?" >> $O; done; done
timeout -k 10 120 tools/exp_cplanes_v1 65536 2 2 >> $O 2>&1
timeout -k 10 120 tools/exp_cplanes_v1 65536 3 1 >> $O 2>&1
cat $O
