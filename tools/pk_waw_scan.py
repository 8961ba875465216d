#!/usr/bin/env python3
"""Scan a hipcc -S listing for the sequence that produced wrong results on gfx950: a packed fp32
instruction (v_pk_mul/add/fma_f32) writing a register pair, followed within `window` instructions by a single-pass VALU
instruction that overwrites one half of that pair (write-after-write).
    python tools/pk_waw_scan.py file.s [window]"""
import re, sys

def main():
    path = sys.argv[1]; window = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    fn, hits, recent = None, {}, []
    for line in open(path):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            fn, recent = m.group(1), []
            continue
        t = line.strip()
        if not line.startswith("\t") or not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        if not op.startswith("v_"):
            if op.startswith(("s_barrier", "s_cbranch", "s_branch")):
                recent = []
            continue
        dst = t.split()[1].rstrip(",") if len(t.split()) > 1 else ""
        written = set()
        mm = re.match(r"v\[(\d+):(\d+)\]", dst)
        if mm:
            written = set(range(int(mm.group(1)), int(mm.group(2)) + 1))
        elif re.match(r"v(\d+)$", dst):
            written = {int(dst[1:])}
        if not op.startswith("v_pk_") and not op.startswith("v_mfma"):
            for age, (pop, pw) in enumerate(reversed(recent[-window:])):
                if written & pw:
                    hits.setdefault(fn, []).append(f"{pop} -> {op} {dst} (distance {age + 1})")
        if op.startswith(("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32")):
            recent.append((t[:40], written))
        else:
            recent.append((op, set()))
    total = sum(len(v) for v in hits.values())
    print(f"{path}: {total} packed-fp32 write-after-write sequences in {len(hits)} kernels")
    for k, v in sorted(hits.items(), key=lambda kv: -len(kv[1]))[:12]:
        print(f"  {len(v):4d}  {k[:110]}   e.g. {v[0]}")

if __name__ == "__main__":
    main()
