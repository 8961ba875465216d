import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    import torch
    # the host-side oracle (MKL-DNN conv3d) collapses when oversubscribed (256 SMT threads on the GPU box)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    # libtmf_hip.so must be the build of the sources next to it (_lib.load() refuses a stale one): bring it up to date
    # here — a no-op when the stamps match, an incremental hipcc build otherwise (hipcc cross-compiles without a GPU)
    try:
        from transmf_ad_amd import build as _b
        lib_stamp = _b._read(_b.LIB + ".stamp")
        if not os.path.exists(_b.LIB) or lib_stamp != _b.source_digest():
            _b.build(verbose=False)
    except Exception as e:           # no hipcc: the tests that need the library will say so themselves
        print(f"[conftest] could not (re)build libtmf_hip.so: {e}", file=sys.stderr)
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than ~20 s")


def pytest_collection_modifyitems(config, items):
    """GPU tests fail loudly (not skip) when selected without a GPU: a silent skip would
    look like a pass.  They are deselected by `-m "not gpu"` on the CPU box."""
    return
