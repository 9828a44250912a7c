#!/usr/bin/env python3
"""Register / spill / LDS table of every kernel in one HIP source (from the code object metadata hipcc emits).
usage: tools/kres.py mpnn_amd/csrc/mpnn_edge.hip [extra hipcc flags]"""
import re
import subprocess
import sys
import tempfile

src, extra = sys.argv[1], sys.argv[2:]
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Iinclude", "-Impnn_amd/csrc",
                    "-S", "--cuda-device-only", "-o", f.name, src] + extra, check=True, stderr=subprocess.DEVNULL)
    text = open(f.name).read()
blocks = re.split(r"\n  - \.agpr_count:", text)[1:]
print(f"{'kernel':58s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'spill':>6s} {'scratch':>8s} {'lds':>7s}")
for b in blocks:
    get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", b) or [None, "?"])[1]
    name = get("name")
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"\(anonymous namespace\)::|\(.*$", "", short)
    short = short.replace("void ", "")
    agpr = b.split("\n")[0].strip()
    print(f"{short[:58]:58s} {get('vgpr_count'):>5s} {agpr:>5s} {get('sgpr_count'):>5s} {get('vgpr_spill_count'):>6s} "
          f"{get('private_segment_fixed_size'):>8s} {get('group_segment_fixed_size'):>7s}")
