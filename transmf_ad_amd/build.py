"""Build libtmf_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library has a
plain C ABI (include/tmf_hip.h) and links only the HIP runtime."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtmf_hip.so")
SOURCES = ["conv3d_mfma.hip", "conv3d_wino.hip", "conv3d_winox.hip", "conv3d_bf16.hip", "conv1_fused.hip", "conv1_gram.hip", "bn_act_pool.hip", "attention.hip", "token_ops.hip", "token_gemm.hip", "xformer_fused.hip", "snet_path.hip", "fusion_path.hip", "input_pipeline.hip", "heads.hip", "adam.hip"]
# -fno-slp-vectorize: clang's SLP pass pairs adjacent scalar fp32 adds / muls into v_pk_*_f32.  Beside MFMAs that is
# slower on gfx950 (MI355X_MICROARCH.md: packed fp32 fillers cost +22..26 cycles per MFMA gap against scalar v_fma_f32) and
# it makes the hot loops depend on the vectoriser's cost model of the day.  (Round 1 also blamed a run-to-run
# nondeterminism on a hardware hazard of the packed sequence; round 2 refuted that — profiles/r02_pk_waw_investigation.txt:
# the sequence is correct in isolation, wait states do not cure the old failing object, and today's sources are
# bit-reproducible with and without the flag.)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
# per-source additions.  input_pipeline.hip: its interpolation kernels are BIT-identical to a numpy restatement with one
# rounding per written operation; -ffp-contract=fast fuses multiplies and adds across statements and ignores
# `#pragma clang fp contract(off)`, so that file is compiled without contraction.
FILE_FLAGS = {"input_pipeline.hip": ["-ffp-contract=off"]}
# extra compile flags for one-off instrumented builds (e.g. TMF_EXTRA_FLAGS=-DTMF_ABLATE for tools/bf16_ablate.py); part
# of the stamp, so such a library is only loaded by processes started with the same setting
FLAGS += os.environ.get("TMF_EXTRA_FLAGS", "").split()


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libtmf_hip.so cannot be built")


def _digest(paths, extra=""):
    """Content hash of the sources + headers + compile flags an artefact was built from."""
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _headers():
    return [os.path.join(CSRC, "tmf_common.h"), os.path.join(HERE, "..", "include", "tmf_hip.h")]


def source_digest():
    """Digest of everything libtmf_hip.so is made from (all sources, headers, FLAGS).  build() writes it to
    libtmf_hip.so.stamp; _lib.load() refuses a library whose stamp does not match the sources next to it."""
    return _digest([os.path.join(CSRC, s) for s in SOURCES] + _headers(), " ".join(FLAGS) + repr(sorted(FILE_FLAGS.items())))


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link libtmf_hip.so next to this file.  An object is rebuilt when the
    digest of (its source, the headers, FLAGS) differs from the stamp written next to it — so a change of compile flags
    (e.g. -fno-slp-vectorize, a correctness matter: DESIGN.md 3.6) rebuilds everything, unlike an mtime comparison."""
    hipcc = _hipcc()
    hdrs = _headers()
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        flags = FLAGS + FILE_FLAGS.get(src, [])
        dig = _digest([s] + hdrs, " ".join(flags))
        if force or not os.path.exists(o) or _read(o + ".stamp") != dig:
            cmd = [hipcc, "-x", "hip", "-c", s, "-o", o] + flags
            if verbose:
                print(" ".join(cmd), flush=True)
            if os.path.exists(o + ".stamp"):
                os.remove(o + ".stamp")
            procs.append((src, o, dig, subprocess.Popen(cmd)))
    failed = [src for src, _o, _d, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed on {failed}")
    for _src, o, dig, _p in procs:
        with open(o + ".stamp", "w") as f:
            f.write(dig)
    want = source_digest()
    if force or procs or not os.path.exists(LIB) or _read(LIB + ".stamp") != want:
        cmd = [hipcc, "-shared", "-o", LIB] + objs + ["--offload-arch=gfx950"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(LIB + ".stamp", "w") as f:
            f.write(want)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
