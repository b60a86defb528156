#!/bin/bash
# kernel 3 of usf_conv_wgrad_f32 with 2 / 4 waves per sample at small batches: tests, then step times with and without
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_image_training.py -x -q -m gpu > gpurun_out/r3/it38.pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r3/it38.pytest.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
for s in 1 0; do
  USF_WGRAD_RSPLIT=$s FIT_IMAGE_ONLY=device python3 tools/fit_image.py cifar_image 32 2>&1 | tail -1
  USF_WGRAD_RSPLIT=$s FIT_IMAGE_ONLY=device python3 tools/fit_image.py mnist_image 32 2>&1 | tail -1
done
