"""pytest configuration: registers the ``gpu`` marker and puts the product package
(``sink-flash-attention-kernel_amd/``) and the repo root (for ``oracle``) on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sink-flash-attention-kernel_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest


@pytest.fixture(params=["rule", "asm"])
def dkdv(request):
    """Backward parity tests run twice: under the library's own dK/dV kernel rule ("rule": what ships) and with the
    hand-placed kernel forced wherever its body serves the shape ("asm": SFA_FLAG_BWD_DKDV_ASM, so that small test shapes
    reach it too).  Yields the mode; tests/util.py::dkdv_kernel_name() says which kernel name sfa_last_path() must show."""
    from sink_attention import set_backward_options
    prev = set_backward_options(dkdv=request.param)
    try:
        yield request.param
    finally:
        set_backward_options(overlap=prev[0], dkdv=prev[1] or "rule")
