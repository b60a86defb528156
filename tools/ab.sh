# A/B of two library builds inside the cfg2 flow (run on the GPU box from the repo root):
#   usflows_amd/csrc/libusflows_base.so (copy of the build to compare against) vs the current libusflows_hip.so
cd $GRAFT_REPO_ROOT
run() { "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$LABEL', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"; }
for i in 1 2 3; do
LABEL=BASE run env USFLOWS_AMD_LIB=$GRAFT_REPO_ROOT/usflows_amd/csrc/libusflows_base.so
LABEL=NEW run env
done
