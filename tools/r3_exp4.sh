#!/bin/bash
# round 3, experiment 4: ablations of the 32-row-wave coupling kernel (wrong results by design; timing + phase stamps)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp4.log
: > $O
for abl in 0 1 2 4 6 8 16 32 64 127 0; do echo "-- ablation mask $abl" >> $O; timeout -k 10 120 tools/exp_cplanes_abl$abl 65536 3 2 2>&1 | grep -A1 "32-row waves" >> $O || echo "rc $?" >> $O; done
cat $O
