#!/usr/bin/env bash
# The byte-table kernel's step loop sits at the 128-VGPR limit: any change can make the allocator spill into it (a scratch
# reload there also waits for the prefetched code rows).  Disassemble and count scratch / flat operations in the loop.
#   scripts/check_q8_isa.sh [extra -D flags]
set -eu
cd "$(dirname "$0")/../annlite_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" -S --cuda-device-only scan_q8.hip -o /tmp/q8_check.s 2>/dev/null
for sk in Lb1E Lb0E; do
  a=$(grep -n "^_ZN7annlite18adc_scan_q8_kernelILi16ELi16E${sk}Li2ELi1ELb1EEEvNS_8ScanArgsE:" /tmp/q8_check.s | cut -d: -f1)
  b=$(grep -n "amdhsa_kernel _ZN7annlite18adc_scan_q8_kernelILi16ELi16E${sk}Li2ELi1ELb1E" /tmp/q8_check.s | cut -d: -f1)
  sed -n "${a},${b}p" /tmp/q8_check.s > /tmp/q8_check_k.s
  # (the step loop's look-ups are the FIRST 16 of these in the kernel; the consumer's row queue has its own, out of line)
  lo=$(grep -n "ds_read_b128 .* offset:256" /tmp/q8_check_k.s | sed -n 1p | cut -d: -f1)
  hi=$(grep -n "ds_read_b128 .* offset:256" /tmp/q8_check_k.s | sed -n 16p | cut -d: -f1)
  lo=$((lo - 160)); hi=$((hi + 90))
  n_s=$(sed -n "${lo},${hi}p" /tmp/q8_check_k.s | grep -c "scratch_" || true)
  n_f=$(grep -c "flat_" /tmp/q8_check_k.s || true)
  n_v=$(sed -n "${lo},${hi}p" /tmp/q8_check_k.s | grep -cE "^\s+v_" || true)
  echo "kernel $sk: lines $(wc -l < /tmp/q8_check_k.s), scratch ops in the step loop region: $n_s, VALU there: $n_v, flat ops in the kernel: $n_f, total scratch: $(grep -c scratch_ /tmp/q8_check_k.s || true)"
done
