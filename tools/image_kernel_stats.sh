cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp IMAGE_PROFILE_PLAIN=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/img_ktrace -- python3 tools/image_profile.py ${IMG_ROWS:-65536} ${IMG_CFG:-mnist} > gpurun_out/img_rocprof.log 2>&1
f=$(find gpurun_out/img_ktrace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open("gpurun_out/r02_image_kernel_stats.md", "w") as o:
    o.write("# r02: rocprofv3 --kernel-trace --stats of `python3 tools/image_profile.py 65536` (IMAGE_PROFILE_PLAIN=1)\n\n"
            "log_prob of the reference's MNIST image configuration (in_dims [16, 7, 7], ConvNet2D(c_hidden 32, 1 layer, gated,\n"
            "layer-normalised), 2 coupling blocks, Householder 1, conjugation) at 65 536 rows: 3 warm-up + 10 timed calls.\n\n"
            "| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|\n")
    for r in rows[:12]:
        n = r["Name"]; n = n if len(n) < 110 else n[:107] + "..."
        o.write(f"| `{n}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['Percentage']} |\n")
print(open("gpurun_out/r02_image_kernel_stats.md").read())
PY
grep "image flow" gpurun_out/img_rocprof.log
