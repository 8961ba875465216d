#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing.

On gfx950 the fp32 MFMA and the vector ALU are ONE issue resource per SIMD: every v_* instruction
in a conv kernel's stage loop costs 4 cycles of matrix time, so VALU-per-MFMA is the number to drive down.

    hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only csrc/conv3d_mfma.hip -o /tmp/conv.s
    python tools/isa_mix.py /tmp/conv.s <mangled-name-substring> [min_mfma_per_block]
"""
import re, sys


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l.split(":")[0] and not l.startswith(".L") and not l.startswith("\t"))
    blocks, cur, name = [], {}, lines[start]
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end"):
            break
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append((name, cur)); cur, name = {}, l.split(":")[0]
            continue
        if not l.startswith("\t") or t.startswith((".", ";")) or not t:
            continue
        c = classify(t.split()[0])
        cur[c] = cur.get(c, 0) + 1
        if t.startswith(("s_cbranch", "s_branch")):
            cur.setdefault("br", []).append(t.split()[-1])
    blocks.append((name, cur))
    tot = {}
    for n, b in blocks:
        for k, v in b.items():
            if k != "br": tot[k] = tot.get(k, 0) + v
    print("static totals:", tot)
    thr = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    for n, b in blocks:
        if b.get("mfma", 0) >= thr or (len(sys.argv) > 4):
            m = b.get("mfma", 0)
            print(f"{n[:28]:28s} mfma {m:4d} valu {b.get('valu', 0):4d} lds {b.get('lds', 0):3d} vmem {b.get('vmem', 0):3d} salu {b.get('salu', 0):4d} "
                  f"wait {b.get('wait', 0):3d} bar {b.get('barrier', 0)}  valu/mfma {b.get('valu', 0) / max(m, 1):.2f}  -> {' '.join(b.get('br', []))}")


if __name__ == "__main__":
    main()
