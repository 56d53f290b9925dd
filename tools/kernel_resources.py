"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks: one line per kernel (VGPRs, spills, scratch, LDS, occupancy).
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2> remarks.txt; python tools/kernel_resources.py remarks.txt [name filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r"remark: (.*?)(?: \[-Rpass)", line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = body.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in body:
        k, v = body.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = re.sub(r"\(.*", "", dem).replace("void doa::", "")
    if flt and flt not in dem:
        continue
    print(f"{dem:90s} VGPR {r.get('VGPRs','?'):>4s} spill {r.get('VGPRs Spill','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} LDS {r.get('LDS Size [bytes/block]','?'):>6s} occ {r.get('Occupancy [waves/SIMD]','?')}")
