cd $GRAFT_REPO_ROOT
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('roofline',{}) and d['roofline'].get('all_kernels_ms_per_step'))"; }
for b in 256 1024 4096 16384; do
USFLOWS_AMD_LIB=$GRAFT_REPO_ROOT/usflows_amd/csrc/libusflows_base.so python bench.py --no-cpu-baseline --batch $b --steps 20 2>/dev/null | p base$b
python bench.py --no-cpu-baseline --batch $b --steps 20 2>/dev/null | p new$b
done
