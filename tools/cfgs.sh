cd $GRAFT_REPO_ROOT
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('roofline',{}) and d['roofline'].get('achieved'), d.get('roofline',{}) and d['roofline'].get('all_kernels_ms_per_step'), d.get('param_prep_first_call_s'))"; }
python bench.py --no-cpu-baseline 2>/dev/null | p default
python bench.py --no-cpu-baseline --gemm f32 2>/dev/null | p f32
python bench.py --no-cpu-baseline --mode sample 2>/dev/null | p sample
python bench.py --no-cpu-baseline --dim 3072 --blocks 48 --hidden 1024 1024 --batch 32768 --steps 3 --warmup 1 2>/dev/null | p cfg4
python bench.py --no-cpu-baseline --batch 1024 --steps 20 2>/dev/null | p b1024
