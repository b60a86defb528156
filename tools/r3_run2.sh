#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
O=gpurun_out/r3/run2
timeout -k 10 1000 python -m pytest tests/test_planes_gpu.py tests/test_configs_gpu.py tests/test_flow_gpu.py -m gpu -x -q > $O.pytest.log 2>&1; echo "pytest rc $?" >> $O.pytest.log
tail -5 $O.pytest.log
timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline > $O.bench.json 2> $O.bench.err; echo "bench rc $?"
USF_CP_W32=0 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-fast-mode > $O.bench_w16.json 2> $O.bench_w16.err; echo "bench w16 rc $?"
timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-fast-mode > $O.bench2.json 2>> $O.bench.err; echo "bench rc $?"
timeout -k 10 300 python bench.py --steps 10 --no-cpu-baseline --no-fast-mode --conj --householder 1 > $O.bench_conj.json 2>> $O.bench.err; echo "bench conj rc $?"
timeout -k 10 300 python bench.py --steps 10 --no-cpu-baseline --no-fast-mode --conj --householder 1 --no-merge-affine > $O.bench_conj_nomerge.json 2>> $O.bench.err; echo "bench conj nomerge rc $?"
timeout -k 10 600 python bench.py --config mnist_image --steps 10 > $O.mnist_image.json 2> $O.mnist_image.err; echo "mnist image rc $?"
timeout -k 10 600 python bench.py --config cifar_image --steps 5 > $O.cifar_image.json 2> $O.cifar_image.err; echo "cifar image rc $?"
tail -c 1500 $O.mnist_image.err; tail -c 600 $O.cifar_image.err; tail -c 600 $O.bench.err
python - <<'PY'
import json
for n in ("bench","bench_w16","bench2","bench_conj","bench_conj_nomerge","mnist_image","cifar_image"):
    try:
        d=json.loads(open(f"gpurun_out/r3/run2.{n}.json").read().strip().splitlines()[-1])
        r=d.get("roofline") or {}
        print(n, d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), r.get("avg_launch_ms"), (r.get("all_kernels_ms_per_step") or ""), d["config"].get("merge_affine"), (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("parity_max_rel_vs_cpu_fp32"))
    except Exception as e:
        print(n, "failed", e)
PY
