#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
O=gpurun_out/r3/run10
timeout -k 10 900 python -m pytest tests/test_image_flows.py tests/test_abi.py -m gpu -x -q > $O.pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O.pytest.log
for v in 1 0; do
USF_CONV_RES=$v timeout -k 10 600 python bench.py --config mnist_image --steps 10 --no-cpu-baseline > $O.mnist_image_res$v.json 2> $O.mnist_image.err; echo "mnist image rc $?"
done
python - <<'PY'
import json
for n in ("mnist_image_res1","mnist_image_res0"):
    try:
        d=json.loads(open(f"gpurun_out/r3/run10.{n}.json").read().strip().splitlines()[-1])
        r=d.get("roofline") or {}
        print(n, d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), r.get("avg_launch_ms"))
        for k,v in list(r["all_kernels"].items())[:8]: print("    ",k,v)
    except Exception as e:
        print(n, "failed", e)
PY
