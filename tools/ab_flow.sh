# A/B of two builds of libusflows_hip.so in the cfg2 flow on ONE box: alternating runs of bench.py (headline only), lib A =
# the in-tree build, lib B = $1 (a path inside the repo, e.g. tools/ab/lib_x.so).  Prints samples/s, ms/step and the per-kernel
# ms per step of every run.
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
B=${1:?path of the second library}
for i in 1 2 3; do
  for lib in "" "$B"; do
    env ${lib:+USFLOWS_AMD_LIB=$GRAFT_REPO_ROOT/$lib} python bench.py --steps 30 --warmup 5 --no-also --no-cpu-baseline --no-fast-mode ${AB_ARGS:-} | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('${lib:-in-tree}', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_kernels_ms_per_step'])"
  done
done
