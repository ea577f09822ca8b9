#!/bin/bash
# Development tool: per-kernel VGPR / SGPR / scratch / LDS / occupancy of a HIP source, from the compiler remarks.
src=${1:-bevy_hanabi_amd/csrc/hanabi_amd.hip}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value -Iinclude "$@" -c "$src" -o /tmp/ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur=m.group(1); d={}; continue
    for k in ("VGPRs","AGPRs","SGPRs","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]","LDS Size [bytes/block]"):
        m=re.search(re.escape(k)+r": (\d+)",l)
        if m and cur: d[k]=m.group(1)
    if cur and "LDS Size" in l:
        import subprocess
        name=subprocess.run(["c++filt",cur],capture_output=True,text=True).stdout.strip()
        name=re.sub(r"\(.*","",name)[:90]
        print("%-92s vgpr %3s agpr %2s sgpr %3s scratch %4s occ %s lds %s"%(name,d.get("VGPRs"),d.get("AGPRs"),d.get("SGPRs"),d.get("ScratchSize [bytes/lane]"),d.get("Occupancy [waves/SIMD]"),d.get("LDS Size [bytes/block]")))
        cur=None
'
