#!/bin/bash
# small-batch flat training after the queued gradient jobs / bound flat gradients / 8-way split-K: tests, then the step time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_train_kernels_gpu.py tests/test_training_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/r3/it31.pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r3/it31.pytest.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
python3 tools/fit_small_batch.py 32 60 2>&1 | tail -2
python3 tools/fit_small_batch.py 256 30 2>&1 | tail -2
