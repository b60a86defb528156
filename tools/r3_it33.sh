#!/bin/bash
# affine blocks' parameter maps as one launch forward / one backward: tests, then the small-batch step times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_image_training.py tests/test_image_flows.py -x -q -m gpu > gpurun_out/r3/it33.pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r3/it33.pytest.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
FIT_IMAGE_ONLY=device python3 tools/fit_image.py mnist_image 32 2>&1 | tail -1
FIT_IMAGE_ONLY=device python3 tools/fit_image.py mnist_image 256 2>&1 | tail -1
FIT_IMAGE_ONLY=device python3 tools/fit_image.py cifar_image 32 2>&1 | tail -1
USFLOWS_AMD_AFFINE_PREP=0 FIT_IMAGE_ONLY=device python3 tools/fit_image.py mnist_image 32 2>&1 | tail -1
