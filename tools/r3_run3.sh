#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
O=gpurun_out/r3/run3
timeout -k 10 120 tools/exp_cplanes_s 65536 3 2 > $O.harness.log 2>&1
timeout -k 10 120 tools/exp_cplanes_s 65536 2 2 >> $O.harness.log 2>&1
timeout -k 10 120 tools/exp_cplanes_s 65536 3 1 >> $O.harness.log 2>&1
timeout -k 10 120 tools/exp_cplanes_s 777 3 2 >> $O.harness.log 2>&1
grep -v "^   differs" $O.harness.log
for r in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-fast-mode > $O.bench_w32_$r.json 2> $O.bench.err; echo "bench rc $?"
USF_CP_W32=0 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-fast-mode > $O.bench_w16_$r.json 2> $O.bench_w16.err; echo "bench w16 rc $?"
done
timeout -k 10 300 python bench.py --steps 10 --no-cpu-baseline --no-fast-mode --conj --householder 1 > $O.bench_conj.json 2>> $O.bench.err; echo "bench conj rc $?"
python - <<'PY'
import json
for n in ("bench_w32_1","bench_w16_1","bench_w32_2","bench_w16_2","bench_conj"):
    try:
        d=json.loads(open(f"gpurun_out/r3/run3.{n}.json").read().strip().splitlines()[-1])
        r=d.get("roofline") or {}
        print(n, d["value"], d["ms_per_step"], r.get("frac"), r.get("avg_launch_ms"), (r.get("all_kernels_ms_per_step") or ""), d["config"].get("merge_affine"))
    except Exception as e:
        print(n, "failed", e)
PY
