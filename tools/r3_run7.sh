#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
O=gpurun_out/r3/run7
timeout -k 10 900 python -m pytest tests/test_training_gpu.py tests/test_image_flows.py -m gpu -x -q > $O.pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O.pytest.log
for v in 1 0; do
USF_CONV_WREG=$v timeout -k 10 600 python bench.py --config mnist_image --steps 10 --no-cpu-baseline > $O.mnist_image_wreg$v.json 2> $O.mnist_image.err; echo "mnist image rc $?"
done
USF_CONV_WREG=1 timeout -k 10 600 python bench.py --config cifar_image --steps 5 --no-cpu-baseline > $O.cifar_image_wreg1.json 2> $O.cifar_image.err; echo "cifar image rc $?"
python - <<'PY'
import json
for n in ("mnist_image_wreg1","mnist_image_wreg0","cifar_image_wreg1"):
    try:
        d=json.loads(open(f"gpurun_out/r3/run7.{n}.json").read().strip().splitlines()[-1])
        r=d.get("roofline") or {}
        print(n, d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), r.get("avg_launch_ms"))
        for k,v in list(r["all_kernels"].items())[:4]: print("    ",k,v)
    except Exception as e:
        print(n, "failed", e)
PY
