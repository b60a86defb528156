#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
O=gpurun_out/r3/run6
timeout -k 10 300 python tools/fit_small_batch.py 32 60 > $O.fit32.log 2>&1; echo "fit32 rc $?"; tail -3 $O.fit32.log
timeout -k 10 300 python tools/fit_small_batch.py 256 30 > $O.fit256.log 2>&1; echo "fit256 rc $?"; tail -2 $O.fit256.log
timeout -k 10 900 python -m pytest tests/test_training_gpu.py tests/test_sophia.py tests/test_train_kernels_gpu.py -m gpu -x -q > $O.pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O.pytest.log
