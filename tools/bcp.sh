#!/bin/bash
# build the coupling-on-planes harness variants: tools/bcp.sh name "flags" [name "flags" ...]
cd /root/repo/tools || exit 1
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off"
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc $F $2 exp_cplanes.hip -o exp_cplanes_$1 > /tmp/bcp_$1.log 2>&1 || { echo "build $1 failed"; grep error /tmp/bcp_$1.log | head -5; }
  shift 2
done
ls exp_cplanes_* | tr '\n' ' '
