#!/bin/bash
mkdir -p gpurun_out/r3
timeout -k 10 600 python3 -m pytest tests/test_image_training.py -x -q -m gpu > gpurun_out/r3/it6.pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r3/it6.pytest.log
for spec in "mnist_image 65536" "cifar_image 16384"; do
  set -- $spec
  timeout -k 10 300 python3 bench.py --config $1 --mode train --batch $2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3/it6.$1.json 2> gpurun_out/r3/it6.err || { echo FAILED $1; tail -5 gpurun_out/r3/it6.err; exit 1; }
  python3 - $1 <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r3/it6.{sys.argv[1]}.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["kernel_ms_per_step"])
for k, v in list(r["all_kernels"].items())[:16]:
    print("     ", k, v)
PY
done
