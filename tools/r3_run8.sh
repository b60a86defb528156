#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
O=gpurun_out/r3/run8.log
: > $O
USF_CONV_WREG=0 python tools/bench_conv_w.py >> $O 2>&1
for d in 0 1 2 4 8 16 32 3 5 7 29 63 0; do USF_CONVW_DBG=$d python tools/bench_conv_w.py 2>&1 | grep WREG >> $O; done
cat $O
