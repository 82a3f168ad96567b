import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the suite forces rare paths (tiny buffers, injected failures, kernels side by side ...) through the library's DEVELOPMENT switches, which
# it reads only under E2S_DEV=1 (ebwt2snp_amd/csrc/e2s_env.h); child processes inherit it
os.environ["E2S_DEV"] = "1"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
