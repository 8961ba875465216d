#!/usr/bin/env python3
"""Build a variant of libtmf_hip.so in which ONE source is replaced (and / or compiled with extra flags): kernel A/B runs.
    python tools/build_variant.py NAME [--replace conv3d_wino.hip=tools/_alt/x.hip] [--flags "-DFOO=1"]
-> transmf_ad_amd/libtmf_NAME.so, to be loaded with TMF_LIB=...  The other objects are the regular build's."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transmf_ad_amd import build as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--replace", action="append", default=[])
    ap.add_argument("--flags", default="")
    a = ap.parse_args()
    B.build(verbose=False)
    rep = dict(r.split("=") for r in a.replace)
    objs = []
    os.makedirs(os.path.join(ROOT, "tools", "_alt"), exist_ok=True)
    for src in B.SOURCES:
        if src in rep:
            o = os.path.join(ROOT, "tools", "_alt", f"{a.name}_{src.replace('.hip', '.o')}")
            cmd = [B._hipcc(), "-x", "hip", "-c", os.path.join(ROOT, rep[src]), "-o", o, "-I", B.CSRC] + B.FLAGS + B.FILE_FLAGS.get(src, []) + a.flags.split()
            subprocess.check_call(cmd)
            objs.append(o)
        else:
            objs.append(os.path.join(B.CSRC, src.replace(".hip", ".o")))
    lib = os.path.join(ROOT, "transmf_ad_amd", f"libtmf_{a.name}.so")
    subprocess.check_call([B._hipcc(), "-shared", "-o", lib] + objs + ["--offload-arch=gfx950"])
    print(lib)


if __name__ == "__main__":
    main()
