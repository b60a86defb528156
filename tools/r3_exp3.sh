#!/bin/bash
# round 3, experiment 3: coupling 32-row waves, register diet in phase 3; fragment read-ahead 1 / 2 / 3 tiles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp3.log
: > $O
for r in 1 2; do for ah in 1 2 3; do echo "-- AH=$ah" >> $O; timeout -k 10 120 tools/exp_cplanes_ah$ah 65536 3 2 >> $O 2>&1 || echo "rc $?" >> $O; done; done
for a in "32768 3 2" "65536 2 2" "65536 3 1" "1000 3 2" "65536 2 1" "100 3 2"; do
  timeout -k 10 120 tools/exp_cplanes_ah3 $a >> $O 2>&1 || echo "exp_cplanes $a rc $?" >> $O
done
timeout -k 10 120 tools/exp_cplanes_s 65536 3 2 >> $O 2>&1 || echo "rc $?" >> $O
timeout -k 10 120 tools/exp_cplanes_s 65536 2 2 >> $O 2>&1 || echo "rc $?" >> $O
cat $O
