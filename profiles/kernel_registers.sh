#!/bin/bash
# kernel_registers.sh — register / spill / LDS table of the POA kernels from the compiler's resource-usage remarks (no GPU needed).
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -DHYPO_BUILD_ID=\"x\" -c $R/hypo_amd/csrc/poa_kernel.hip \
    -Rpass-analysis=kernel-resource-usage -o /tmp/poa_dev.o 2>&1 | python3 -c '
import re, sys
print("kernel,sgprs,sgpr_spills,vgprs,vgpr_spills,lds_bytes,waves_per_simd_by_vgprs")
cur = {}
def flush():
    if cur.get("Name"):
        n = cur["Name"]
        m = re.search(r"PoaCfgILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)", n)
        short = f"poa_class_kernel<GW={m.group(1)} CPL={m.group(2)} LMAX={m.group(3)} NMAX={m.group(4)}>" if m else re.sub(r"^_ZN4hypo\d+", "", n)[:40]
        v = int(cur["VGPRs"])
        print(",".join([short, cur["TotalSGPRs"], cur["SGPRs Spill"], str(v), cur["VGPRs Spill"], cur["LDS Size [bytes/block]"], str(512 // ((v + 7) // 8 * 8))]))
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+(?: \[[^\]]+\])?): (\S+)", line)
    if not m: continue
    k, val = m.group(1).strip(), m.group(2)
    if k == "Function Name": flush(); cur.clear(); cur["Name"] = val
    else: cur[k] = val
flush()'
