#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3/full.pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3/full.pytest.log
tail -6 gpurun_out/r3/full.pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3/smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r3/smoke.log
