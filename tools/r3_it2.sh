#!/bin/bash
# image-flow training step: device backward (usf_conv_wgrad_f32 ...) against torch autograd + MIOpen, by batch size
mkdir -p gpurun_out/r3
out=gpurun_out/r3/it2.log
: > $out
for cfg in mnist_image cifar_image; do
  for B in 32 256 4096 16384 65536; do
    if [ $cfg = cifar_image ] && [ $B -gt 16384 ]; then continue; fi
    for dev in 1 0; do
      USFLOWS_AMD_IMAGE_TRAIN=$dev timeout -k 10 300 python3 bench.py --config $cfg --mode train --batch $B --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3/it2.$cfg.$B.$dev.json 2> gpurun_out/r3/it2.err || { echo "FAILED $cfg $B $dev" >> $out; tail -5 gpurun_out/r3/it2.err >> $out; exit 1; }
      python3 - "$cfg" "$B" "$dev" >> $out <<'PY'
import json, sys
cfg, B, dev = sys.argv[1:4]
d = json.loads(open(f"gpurun_out/r3/it2.{cfg}.{B}.{dev}.json").read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(cfg, "B", B, "device" if dev == "1" else "torch ", d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), r.get("kernel_ms_per_step"))
if dev == "1" and B in ("65536", "16384"):
    for k, v in list((r.get("all_kernels") or {}).items())[:14]:
        print("     ", k, v)
PY
    done
  done
done
cat $out
