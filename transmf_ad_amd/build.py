"""Build libtmf_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library has a
plain C ABI (include/tmf_hip.h) and links only the HIP runtime."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtmf_hip.so")
SOURCES = ["conv3d_mfma.hip", "conv3d_bf16.hip", "conv1_fused.hip", "bn_act_pool.hip", "attention.hip", "token_ops.hip", "token_gemm.hip"]
# -fno-slp-vectorize: the SLP vectoriser pairs scalar fp32 work into v_pk_*_f32 and then patches one half of the pair
# with a single-pass instruction (v_pk_mul_f32 v[6:7] ...; v_mov_b32 v6, v5; v_pk_add_f32 ..., v[6:7]).  On gfx950 that
# sequence intermittently delivered the stale half in lanes 16-31 when the instructions issued back to back (the
# first-block reduce pass at the tail of its grid: DESIGN.md 3.6) — found as a loss of run-to-run bit reproducibility.
# Without the pass none of these sequences are left in the conv / BatchNorm / attention kernels (tools/pk_waw_scan.py).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libtmf_hip.so cannot be built")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link libtmf_hip.so next to this file."""
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, "tmf_common.h"), os.path.join(HERE, "..", "include", "tmf_hip.h")]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc, "-x", "hip", "-c", s, "-o", o] + FLAGS
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-o", LIB] + objs + ["--offload-arch=gfx950"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
