#!/usr/bin/env python3
"""Instruction mix per kernel instance from hipcc's ISA (--save-temps), compiled with the flags __graft_entry__.SOURCES gives the file.
    python tools/isa_mix.py gat_layer_fused.hip [substring ...]
Straight-line counts (the fused kernels are fully unrolled; loops are counted once)."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
name, flt = sys.argv[1], sys.argv[2:]
src = os.path.join(ROOT, "bathymetric-gnn_amd", "csrc", name)
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "x.o", "-Wno-unused-result",
                    "--save-temps"] + list(ge.SOURCES.get(name, [])), cwd=td, check=True, capture_output=True)
    s = open(os.path.join(td, name.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
parts = re.split(r"\n(_Z\w+): +; @\w+\n", s)
names, bodies = parts[1::2], parts[2::2]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for d, b in zip(dem, bodies):
    d = re.sub(r"\(.*", "", d).replace("void bgnn::", "")
    if "kernel" not in d or (flt and not any(f in d for f in flt)):
        continue
    body = b.split(".Lfunc_end")[0]
    c = collections.Counter()
    for l in body.split("\n"):
        if not l.startswith("\t") or l.strip().startswith((".", ";")):
            continue
        i = l.strip().split()[0]
        k = ("mfma" if i.startswith("v_mfma") else "trans" if re.match(r"v_(exp|rcp|log|sqrt|rsq|sin|cos)", i) else
             "valu_pk" if i.startswith("v_pk_") else "valu_f64" if re.search(r"_f64", i) and i.startswith("v_") else "valu" if i.startswith("v_") else
             "lds" if i.startswith("ds_") else "vmem" if i.startswith(("global_", "buffer_", "scratch_")) else "waitcnt" if i.startswith("s_waitcnt") else
             "barrier" if i.startswith("s_barrier") else "salu" if i.startswith("s_") else "other")
        c[k] += 1
    print(f"{d[:60]:60s}", " ".join(f"{k} {v}" for k, v in sorted(c.items())))
