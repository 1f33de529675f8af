#!/usr/bin/env python3
"""Diagnostic builds of ONE kernel source with extra -D flags, linked against the regular objects into a variant
library that `TECM_LIB=<path>` selects (tecmollm/_lib.py):

    python tools/build_variant.py spatial_fwd.hip stamps -DSPF_STAMPS
    TECM_LIB=tec-mollm_amd/tecmollm/variants/libtecmollm_hip_stamps.so python tools/spatial_bench.py
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g


def main():
    src, name, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    g.build()
    out_dir = os.path.join(g.PKG, "tecmollm", "variants")
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(g.OBJ_DIR, f"{os.path.splitext(src)[0]}__{name}.o")
    subprocess.check_call([g.HIPCC, *g.FLAGS, *flags, "-x", "hip", "-c", os.path.join(g.CSRC, src), "-o", obj])
    objs = [os.path.join(g.OBJ_DIR, os.path.splitext(s)[0] + ".o") for s in g.SOURCES if s != src] + [obj]
    lib = os.path.join(out_dir, f"libtecmollm_hip_{name}.so")
    subprocess.check_call([g.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    print(lib)


if __name__ == "__main__":
    main()
