#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp6.log
: > $O
for r in 1 2 3; do for v in 0 1; do echo "-- v$v f16x2" >> $O; timeout -k 10 120 tools/exp_cplanes_v$v 65536 2 2 2>&1 | grep -v "waves\]" >> $O || echo "rc $?" >> $O; done; done
for v in 0 1; do echo "-- v$v f16x2 NH=1" >> $O; timeout -k 10 120 tools/exp_cplanes_v$v 65536 2 1 2>&1 | grep -v "waves\]" >> $O; done
for v in 0 1; do echo "-- v$v bf16x3" >> $O; timeout -k 10 120 tools/exp_cplanes_v$v 65536 3 2 2>&1 | grep -v "waves\]" >> $O; done
cat $O
