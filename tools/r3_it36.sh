#!/bin/bash
# smaller sample groups per block for the convolution at small batches: tests, then small-batch inference / training times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_image_flows.py tests/test_image_training.py -x -q -m gpu > gpurun_out/r3/it36.pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r3/it36.pytest.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
for s in 1 0; do
  USF_CONV_SMALL_S=$s python3 tools/image_small_batch.py cifar_image 100 2>&1 | tail -1
  USF_CONV_SMALL_S=$s python3 tools/image_small_batch.py mnist_image 100 2>&1 | tail -1
  USF_CONV_SMALL_S=$s FIT_IMAGE_ONLY=device python3 tools/fit_image.py cifar_image 32 2>&1 | tail -1
  USF_CONV_SMALL_S=$s FIT_IMAGE_ONLY=device python3 tools/fit_image.py mnist_image 32 256 2>&1 | tail -2
done
