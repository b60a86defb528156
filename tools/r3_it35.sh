#!/bin/bash
# transposed pack jobs through LDS tiles: tests, then the flat step times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_prep_gpu.py tests/test_training_gpu.py tests/test_flow_gpu.py -x -q -m gpu > gpurun_out/r3/it35.pytest.log 2>&1
rc=$?; tail -4 gpurun_out/r3/it35.pytest.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
python3 tools/fit_small_batch.py 32 60 2>&1 | tail -1
python3 bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
