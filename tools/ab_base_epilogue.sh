set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_planes_gpu.py tests/test_configs_gpu.py tests/test_flow_gpu.py -q -x -m gpu > gpurun_out/base_epi_tests.log 2>&1 || { tail -30 gpurun_out/base_epi_tests.log; exit 1; }
tail -2 gpurun_out/base_epi_tests.log
for i in 1 2 3; do
  USFLOWS_AMD_TUNE=base_in_epilogue=1 timeout -k 10 200 python bench.py --no-also --no-cpu-baseline --steps 20 --warmup 4 > gpurun_out/ab_base_on_$i.json 2> gpurun_out/ab_base_on_$i.err
  USFLOWS_AMD_TUNE=base_in_epilogue=0 timeout -k 10 200 python bench.py --no-also --no-cpu-baseline --steps 20 --warmup 4 > gpurun_out/ab_base_off_$i.json 2> gpurun_out/ab_base_off_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_base_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"], d.get("parity_max_rel"))
    except Exception as e: print(f, "ERR", e)
PY
