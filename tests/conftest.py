import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure: (re)build it before any test uses it."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    yield


@pytest.fixture(scope="session", autouse=True)
def _build_library():
    """libtgx.so is a build product (git-ignored): bring it up to date where hipcc exists (a no-op after
    __graft_entry__.build(); hipcc cross-compiles for gfx950 without a GPU).  Without hipcc the prebuilt file that
    travelled with the tree is used as is."""
    import shutil

    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "term_amd", "csrc")], check=True)
    yield


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)
