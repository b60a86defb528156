#!/bin/bash
# L2 / fabric counters of the planes weight-gradient kernel in the harness (tools/exp_wgradp, built beforehand):
#   tools/wgradp_pmc.sh [M N K]  ->  gpurun_out/wgradp_traffic.json + per-counter summaries on stdout
# one rocprofv3 --pmc pass per counter group, nothing but --pmc on the command line
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-65536}; N=${2:-784}; K=${3:-784}
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  d=$R/gpurun_out/wgp_pmc_$(echo $c | tr ' ' '_')
  rm -rf $d
  timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d $d -- $R/tools/exp_wgradp $M $N $K > $R/gpurun_out/wgp_pmc.log 2>&1 || exit 1
done
python3 - $R $M $N $K <<'PY'
import csv, glob, json, sys, collections
R, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
agg = collections.defaultdict(list)
for f in glob.glob(R + "/gpurun_out/wgp_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_planes_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
ld = -(-N // 32) * 32, -(-K // 32) * 32
operand_bytes = 3 * 2 * (-(-M // 32) * 32) * (ld[0] + ld[1])
out = {"kernel": "wgrad_planes_kernel", "M": M, "N": N, "K": K, "launches_averaged": len(next(iter(agg.values()))),
       "counters_per_launch": m,
       "fetch_bytes": 2 * 1024 * m.get("FETCH_SIZE", 0), "fetch_correction": "FETCH_SIZE is in KB and counts 128-B requests as 64 B on gfx950 (MI355X_MICROARCH.md, HBM): x 2",
       "write_bytes": 1024 * m.get("WRITE_SIZE", 0),
       "operand_plane_bytes": operand_bytes,
       "l2_hit_rate": m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)),
       "source": "tools/wgradp_pmc.sh: rocprofv3 --pmc <counter group> -- tools/exp_wgradp M N K, one pass per group"}
json.dump(out, open(R + "/gpurun_out/wgradp_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
