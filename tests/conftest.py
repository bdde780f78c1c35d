import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C-ABI library is a build product (git-ignored): build it when the tree is fresh
    lib = os.path.join(ROOT, "katana.jl_amd", "libkatana_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "katana.jl_amd", "csrc")], check=True)


@pytest.fixture(scope="session")
def ktn():
    import katana_jl_amd
    return katana_jl_amd
