#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy table of every kernel in a .hip file, from hipcc's -Rpass-analysis=kernel-resource-usage
remarks, compiled with the flags __graft_entry__.SOURCES gives that file.  No GPU needed.
    python tools/kernel_resources.py gat_layer_fused.hip [substring ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

def main():
    name = sys.argv[1]
    flt = sys.argv[2:]
    src = os.path.join(ROOT, "bathymetric-gnn_amd", "csrc", name)
    extra = list(ge.SOURCES.get(name, []))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null", "-Wno-unused-result",
           "-Rpass-analysis=kernel-resource-usage"] + extra
    t = subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
    filt = subprocess.run(["c++filt"], input="\n".join(b.split()[0] for b in blocks), capture_output=True, text=True).stdout.split("\n")
    for b, dem in zip(blocks, filt):
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
        dem = dem.replace("bgnn::", "").replace("void ", "")
        dem = re.sub(r"\(.*", "", dem)
        if flt and not any(f in dem for f in flt):
            continue
        print(f"{dem[:100]:100s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} scratch {g('ScratchSize .bytes/lane.'):>4} occ {g('Occupancy .waves/SIMD.'):>2} LDS {g('LDS Size .bytes/block.'):>6}")

main()
