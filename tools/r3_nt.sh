#!/bin/bash
# planes GEMM with non-temporal operand loads (1) / + plane stores (3): FETCH_SIZE / WRITE_SIZE per launch and time in the flow
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for v in ${NTV:-0 1 3}; do
  lib=usflows_amd/csrc/libusflows_hip.so
  [ $v != 0 ] && lib=tools/libusflows_hip_nt$v.so
  export USFLOWS_AMD_LIB=$GRAFT_REPO_ROOT/$lib
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fast-mode 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('NT=$v', d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms'], r['all_kernels_ms_per_step'])"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/r3/nt_$v_$ctr
    rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/r3/nt_${v}_$ctr -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fast-mode > /dev/null 2>&1
    python3 - gpurun_out/r3/nt_${v}_$ctr $ctr $v <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"]
        if "gemm_planes_kernel<3, 5, false>" in k or "coupling_planes_kernel<3, 2>" in k:
            agg[k[k.index("usf::") + 5:k.index("(")]].append(float(row["Counter_Value"]))
for k, v in agg.items():
    print("   NT=%s %s %s mean KB %.0f over %d" % (sys.argv[3], sys.argv[2], k, sum(v) / len(v), len(v)))
PY
  done
done
