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
