#!/bin/bash
# captured training step with gradients taken over by autograd (no zeroing, no per-parameter adds): tests, then step times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_image_training.py tests/test_training_gpu.py tests/test_sophia.py tests/test_image_flows.py -x -q -m gpu > gpurun_out/r3/it32.pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r3/it32.pytest.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
FIT_IMAGE_ONLY=device python3 tools/fit_image.py mnist_image 32 2>&1 | tail -2
FIT_IMAGE_ONLY=device python3 tools/fit_image.py mnist_image 256 2>&1 | tail -1
FIT_IMAGE_ONLY=device python3 tools/fit_image.py cifar_image 32 2>&1 | tail -1
python3 tools/fit_small_batch.py 32 60 2>&1 | tail -1
