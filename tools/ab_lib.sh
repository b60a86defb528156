# same-box A/B of two builds of the library (run on the GPU box from the repo root): tools/ab_lib.sh other.so [bench flags]
cd $GRAFT_REPO_ROOT
OTHER=$1; shift
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  python bench.py --no-cpu-baseline --no-fast-mode --no-kernel-timing --steps 20 "$@" 2>/dev/null | p "this build "
  USFLOWS_AMD_LIB=$OTHER python bench.py --no-cpu-baseline --no-fast-mode --no-kernel-timing --steps 20 "$@" 2>/dev/null | p "other build"
done
