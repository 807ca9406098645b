#!/usr/bin/env bash
# every instantiation of the byte-table kernel: scratch operations near its step loop (the densest cluster of LDS look-ups)
set -eu
cd "$(dirname "$0")/../annlite_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" -S --cuda-device-only scan_q8.hip -o /tmp/q8_all.s 2>/dev/null
grep -n "^_ZN7annlite18adc_scan_q8_kernel[A-Za-z0-9_]*:" /tmp/q8_all.s | while IFS=: read a sym rest; do
  b=$(grep -n "amdhsa_kernel $sym" /tmp/q8_all.s | cut -d: -f1)
  sed -n "${a},${b}p" /tmp/q8_all.s > /tmp/q8_one.s
  # densest 100-line window of ds_read_b128 / ds_read_b64
  c=$(grep -n "ds_read_b128\|ds_read_b64" /tmp/q8_one.s | awk -F: '{print int($1/100)}' | sort -n | uniq -c | sort -rn | head -1 | awk '{print $2}')
  lo=$((c*100-150)); hi=$((c*100+250)); [ $lo -lt 1 ] && lo=1
  echo "$(echo $sym | sed 's/_ZN7annlite18adc_scan_q8_kernelI//; s/EEvNS_8ScanArgsE//') : lines $(wc -l < /tmp/q8_one.s), scratch near the step loop [$lo..$hi]: $(sed -n ${lo},${hi}p /tmp/q8_one.s | grep -c scratch_ || true), total scratch $(grep -c scratch_ /tmp/q8_one.s || true), flat $(grep -c flat_ /tmp/q8_one.s || true)"
done
