#!/usr/bin/env python3
"""Instruction mix of a kernel's MFMA region (first v_mfma up to the last branch of the function = the loop's back edge).
usage: tools/imix.py mpnn_amd/csrc/mpnn_edge.hip 'k_messageILi3ELb0ELb0' [extra hipcc flags]"""
import collections
import re
import subprocess
import sys
import tempfile

src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Iinclude", "-Impnn_amd/csrc",
                    "-S", "--cuda-device-only", "-o", f.name, src] + extra, check=True, stderr=subprocess.DEVNULL)
    lines = open(f.name).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]
mf = [i for i, l in enumerate(body) if "v_mfma" in l]
tail = max(i for i in range(mf[-1], len(body)) if "s_cbranch" in body[i] or "s_branch" in body[i] or i == mf[-1])
groups, ops = collections.Counter(), collections.Counter()
for l in body[mf[0]:tail]:
    l = l.strip()
    if not l or l[0] in ";.":
        continue
    op = l.split()[0]
    ops[op] += 1
    if op.startswith("v_mfma"): g = "mfma"
    elif op.startswith("ds_"): g = "lds"
    elif op.startswith(("global_load", "buffer_load")): g = "vmem_load"
    elif op.startswith(("global_store", "buffer_store", "global_atomic")): g = "vmem_store"
    elif op.startswith("scratch_"): g = "scratch"
    elif op.startswith("s_waitcnt"): g = "waitcnt"
    elif op.startswith("s_nop"): g = "nop"
    elif op.startswith("s_"): g = "salu"
    elif op.startswith(("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt")): g = "trans"
    elif op.startswith("v_pk_"): g = "valu_packed"
    elif "dpp" in op: g = "valu_dpp"
    else: g = "valu"
    groups[g] += 1
print(dict(groups))
print(ops.most_common(30))
