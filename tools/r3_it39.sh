#!/bin/bash
# hidden activations saved by the forward plan at small batches: training tests, then the flat step time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 600 python3 -m pytest tests/test_training_gpu.py tests/test_train_kernels_gpu.py -x -q -m gpu > gpurun_out/r3/it39.pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r3/it39.pytest.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
python3 tools/fit_small_batch.py 32 60 2>&1 | tail -1
python3 tools/fit_small_batch.py 256 30 2>&1 | tail -1
