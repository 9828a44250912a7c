#!/usr/bin/env python3
"""Per-kernel register/LDS/spill table from hipcc -Rpass-analysis=kernel-resource-usage."""
import re, subprocess, sys
src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Iinclude", "-Impnn_amd/csrc",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_ru.o"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        cur = {"name": name}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
print(f"{'kernel':48s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>6s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
for r in rows:
    print(f"{r['name'][:48]:48s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>6s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>7s}")
