import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pranet-v2_amd")          # plays the role of the reference's binary_seg/ (cwd of its scripts)
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("PN2_TUNE_COLD", "0")        # tests do not need cold-operand tile tuning (a 512 MB cache-evicting fill per timed candidate)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
