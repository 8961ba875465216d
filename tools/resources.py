#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in the built objects (transmf_ad_amd/csrc/*.o), read from the code-object
metadata — no recompilation.   python tools/resources.py [substring]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "transmf_ad_amd", "csrc")


def kernels_of(obj):
    """[{name, vgpr, sgpr, scratch, lds}] for one host object with an embedded gfx950 code object."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        rc = subprocess.call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(td, "x.o")],
                             stderr=subprocess.DEVNULL)
        if rc != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []                   # host-only object (snet_path.o / fusion_path.o: pure launch orchestration)
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.DEVNULL)
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    out, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"\s*(?:- )?\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "group_segment_fixed_size" and cur.get("name"):
            out.append(cur)
            cur = {}
        if k in ("name", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size"):
            cur[{"name": "name", "vgpr_count": "vgpr", "sgpr_count": "sgpr", "private_segment_fixed_size": "scratch",
                 "group_segment_fixed_size": "lds"}[k]] = v if k == "name" else int(v)
    if cur.get("name"):
        out.append(cur)
    return [k for k in out if "vgpr" in k]


def disassembly_of(obj, out_path):
    """llvm-objdump -d of the gfx950 code object embedded in a host object, rewritten into the shape of a `hipcc -S` listing
    (kernel labels `_Z...:`, one instruction per line, no encodings) for tools/asm_checks.py.  False for host-only objects."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        rc = subprocess.call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(td, "x.o")],
                             stderr=subprocess.DEVNULL)
        if rc != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return False
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.DEVNULL)
        text = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True)
    with open(out_path, "w") as f:
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
            if m:
                f.write(m.group(1) + ":\n")
                continue
            t = line.split("//")[0].rstrip()
            if t.startswith("\t") or t.startswith("  "):
                f.write("\t" + t.strip() + "\n")
    return True


def all_kernels():
    res = {}
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".o"):
            res[f] = kernels_of(os.path.join(CSRC, f))
    return res


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for f, ks in all_kernels().items():
        for k in ks:
            if pat in k["name"]:
                print(f"{f:18s} vgpr {k['vgpr']:4d} scratch {k.get('scratch', 0):5d} lds {k.get('lds', 0):7d}  {k['name'][:110]}")
