#!/usr/bin/env python3
"""Scratch bytes, VGPR count and spill count of every kernel in a HIP source (device-only compile to assembly, metadata note parsed).

usage: python tools/kernel_resources.py torus-fhe_amd/csrc/thfhe_mk.hip [name-filter]
"""
import re
import subprocess
import sys
import tempfile


def resources(src):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-S", "--cuda-device-only",
                        "-o", f.name, src] + sys.argv[3:], check=True, stderr=subprocess.DEVNULL)
        text = open(f.name).read()
    pat = re.compile(r"\.name:\s+(_Z\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)\n(?:\s+\.(?!name)\S+:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)")
    return [(m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))) for m in pat.finditer(text)]


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = resources(sys.argv[1])
    demangle = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True)
    for (name, scratch, vgpr, spill), nice in zip(rows, demangle.stdout.splitlines()):
        nice = re.sub(r"\(anonymous namespace\)::|\(.*$", "", nice).replace("void ", "")
        if flt in nice:
            print(f"{nice:48s} scratch {scratch:5d} B  vgpr {vgpr:3d}  spilled {spill}")
