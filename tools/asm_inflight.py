#!/usr/bin/env python3
"""Counted-wait check of a built object (gfx950): conv3d_winox.hip keeps weight loads to REGISTERS in flight across several
positions of its main loop and waits for them with s_waitcnt vmcnt(N), N = the number of vector-memory operations issued behind
the load it needs (loads, LDS-DMA copies and stores retire in order).  The loads are inline assembly, so the compiler believes
their destination registers are ready from the moment of issue: a copy, a spill or an early use would read stale data, and a
wrong N would let an MFMA start on weights that have not arrived.  This walks the disassembly of the kernel with a model of the
vmcnt queue — straight through the listing, every loop body (backward branch) a second time with the state at its back edge —
and reports every instruction that touches a register whose load is still outstanding.

    python tools/asm_inflight.py transmf_ad_amd/csrc/conv3d_winox.o [kernel-name-substring]      exit code 1 when something is found
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def _regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def functions_of(obj):
    """{kernel: [(address, opcode, operands, branch target address or None)]} from llvm-objdump of the embedded gfx950 code object."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        rc = subprocess.call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(td, "x.o")],
                             stderr=subprocess.DEVNULL)
        if rc != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return {}
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.DEVNULL)
        text = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True)
    funcs, cur, base, name = {}, None, 0, None
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\w+)>:", line)
        if m:
            base, name = int(m.group(1), 16), m.group(2)
            cur = funcs.setdefault(name, [])
            continue
        if cur is None or "//" not in line:
            continue
        code, comment = line.split("//", 1)
        m = re.match(r"\s*(\S+)\s*(.*)", code.rstrip())
        ma = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        if not m or not ma:
            continue
        target = None
        mt = re.search(r"<" + re.escape(name) + r"\+0x([0-9a-f]+)>", comment)
        if mt and m.group(1).startswith(("s_cbranch", "s_branch")):
            target = base + int(mt.group(1), 16)
        cur.append((int(ma.group(1), 16), m.group(1), m.group(2).strip(), target))
    return funcs


_VMEM = re.compile(r"^(buffer|global|scratch|flat)_(load|store|atomic)")


def check_function(ins):
    """-> (violations, register loads seen, deepest queue).  ins: the list functions_of() returns for one kernel.
    The listing is cut into basic blocks; every control-flow EDGE is walked once, with the queue as it is at the end of (one
    walk of) its source block — so a loop body is seen with the state of its entry and with the state of its back edge."""
    index = {a: i for i, (a, _o, _g, _t) in enumerate(ins)}
    leaders = {0}
    for i, (_a, op, _g, target) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            if i + 1 < len(ins):
                leaders.add(i + 1)
            if target is not None and target in index:
                leaders.add(index[target])
    starts = sorted(leaders)
    end_of = {b: (starts[k + 1] - 1 if k + 1 < len(starts) else len(ins) - 1) for k, b in enumerate(starts)}
    bad, seen_bad, stats = [], set(), {"loads": 0, "depth": 0}

    def walk(b, fifo):
        for i in range(b, end_of[b] + 1):
            addr, op, args, _t = ins[i]
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", args)
                if m:
                    del fifo[: max(0, len(fifo) - int(m.group(1)))]
                continue
            vmem = bool(_VMEM.match(op))
            if vmem or not op.startswith("s_"):
                used = set(_regs(args))
                for dst, at in fifo:
                    hit = dst & used
                    if hit and (addr, at) not in seen_bad:
                        seen_bad.add((addr, at))
                        bad.append(f"{addr:#x}: '{op} {args}' touches v{min(hit)}, in flight since {at:#x}")
            if vmem:
                is_load = "_load" in op and not re.search(r"\blds\b", args)
                fifo.append((set(_regs(args.split(",")[0])) if is_load else set(), addr))
                stats["depth"] = max(stats["depth"], len(fifo))
        return fifo

    loads = sum(1 for _a, op, args, _t in ins if op.startswith("buffer_load") and not re.search(r"\blds\b", args))
    edges, work = set(), [(0, [])]
    while work:
        b, fifo = work.pop()
        out = walk(b, list(fifo))
        last = ins[end_of[b]]
        succ = []
        if last[1].startswith(("s_cbranch", "s_branch")) and last[3] is not None and last[3] in index:
            succ.append(index[last[3]])
        if not last[1].startswith(("s_branch", "s_endpgm", "s_setpc")) and end_of[b] + 1 < len(ins):
            succ.append(end_of[b] + 1)
        for t in succ:
            if (b, t) not in edges:
                edges.add((b, t))
                work.append((t, out))
    return bad, loads, stats["depth"]


def check_object(obj, only=""):
    out = []
    for name, ins in functions_of(obj).items():
        if only in name:
            bad, loads, depth = check_function(ins)
            out.append((name, bad, loads, depth))
    return out


if __name__ == "__main__":
    rc = 0
    for name, bad, loads, depth in check_object(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""):
        for b in bad[:30]:
            print(name[:50], b)
        print(f"{name[:70]}: {loads} register loads, deepest vmcnt queue {depth}, {len(bad)} accesses to registers in flight")
        rc |= 1 if bad or depth > 63 else 0
    sys.exit(rc)
