#!/bin/bash
# register / branch summary of the fused coupling on planes (tuning aid): tools/cpregs.sh [extra -D flags]
cd /root/repo/usflows_amd/csrc || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off "$@" -S --cuda-device-only usf_coupling_planes.hip -o /tmp/cp.s 2>/tmp/cp.err
grep -E "error" -A3 /tmp/cp.err | head -20
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|sgpr_count|name):" /tmp/cp.s | paste - - - - | sed 's/\s\+/ /g' | cut -c1-160
awk '/^_ZN3usf22coupling_planes_kernelILi3ELi2EEEvNS_8CplPArgsE:/,/s_endpgm/' /tmp/cp.s > /tmp/cp32.s
echo "<3,2>: branches $(grep -c s_cbranch /tmp/cp32.s), mfma $(grep -c v_mfma /tmp/cp32.s), lines $(wc -l < /tmp/cp32.s)"
