"""Two checks of a gfx950 assembly listing (hipcc -S --cuda-device-only, or the disassembly of a built object:
tools/resources.disassembly_of) that the compiler does not make for us.

1. in flight: the persistent Winograd kernels (csrc/conv3d_wino.hip: load_b / bpin) keep weight loads to REGISTERS
   (buffer_load_dwordx4 from inline asm) in flight across a phase on purpose.  The compiler believes the asm's outputs are ready,
   so a copy or a spill of such a register before the next s_waitcnt vmcnt(0) would read stale data.
2. store data: a buffer/global store of more than 64 bits whose data registers a vector instruction overwrites in the very
   next slot.  LLVM's hazard recognizer exempts stores with an SGPR soffset; on the MI355X such a pair (buffer_store_dwordx4
   v[70:73] .. s4 offen; v_pk_add_f32 v[70:71], ..) stored a wrong v71 in lanes 12-15 of every row of 16 (round 5: found through
   the Winograd forward's statistics, fixed with store_guard()).

    python tools/asm_checks.py listing.s [kernel-name-substring]         exit code 1 when something is found
tests/test_host_cpu.py runs the store check on the disassembly of EVERY built object and the in-flight check on conv3d_wino.o.
"""
import re
import sys


def _regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def _instructions(path, only):
    """(kernel, line number, opcode, operand text); kernel changes reset the caller's state through a None opcode."""
    kern = None
    for i, line in enumerate(open(path).read().split("\n"), 1):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            kern = m.group(1)
            yield kern, i, None, ""
            continue
        if not t or t[0] in ";." or kern is None or only not in kern:
            continue
        t = t.split(";")[0].strip()
        m = re.match(r"(\S+)\s*(.*)", t)
        if m and not m.group(1).endswith(":"):
            yield kern, i, m.group(1), m.group(2)


def inflight_uses(path, only=""):
    inflight, bad, loads = {}, [], 0
    for kern, i, op, args in _instructions(path, only):
        if op is None or (op == "s_waitcnt" and "vmcnt(0)" in args):
            inflight = {}
            continue
        if op.startswith("buffer_load_dwordx4") and " lds" not in args:
            for r in _regs(args.split(",")[0]):
                inflight[r] = i
            loads += 1
            continue
        if not inflight or op.startswith("s_"):
            continue
        for r in _regs(args):
            if r in inflight:
                bad.append(f"{kern[:60]} line {i}: '{op} {args}' touches v{r}, in flight since line {inflight[r]}")
    return loads, bad


def store_data_overwrites(path, only="", wait_states=1):
    pending, bad, stores = [], [], 0                           # (data registers, wait states since the store, line)
    for kern, i, op, args in _instructions(path, only):
        if op is None:
            pending = []
            continue
        if op.startswith("v_"):
            dst = set(_regs(args.split(",")[0]))
            for rs, age, at in pending:
                if age < wait_states and dst & rs:
                    bad.append(f"{kern[:60]} line {i}: '{op} {args}' overwrites the data of the store at line {at} after {age} wait states")
        step = int(args.strip() or 0) + 1 if op == "s_nop" else 1
        pending = [(rs, age + step, at) for rs, age, at in pending if age + step < wait_states]
        if re.match(r"(buffer|global|flat)_store_dwordx[34]", op):
            data = args.split(",")[1 if op[0] in "gf" else 0]
            pending.append((set(_regs(data)), 0, i))
            stores += 1
    return stores, bad


if __name__ == "__main__":
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    n1, b1 = inflight_uses(sys.argv[1], only)
    n2, b2 = store_data_overwrites(sys.argv[1], only)
    for line in (b1 + b2)[:40]:
        print(line)
    print(f"{n1} register loads from asm, {len(b1)} uses before the wait; {n2} wide stores, {len(b2)} overwritten in the next slot")
    sys.exit(1 if b1 or b2 else 0)
