import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "universal-volumetric_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_OUT = "/root/reference/example/public/liam/output"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def hipemu_lib():
    """Test-only host emulation build of the product sources (tests/hipemu); never shipped."""
    import fcntl
    import subprocess
    with open(os.path.join(ROOT, "tests", "hipemu", ".build.lock"), "w") as lk:      # pytest-xdist workers build it once, in turn
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "universal-volumetric_amd"), "hipemu"])
    return os.path.join(ROOT, "tests", "hipemu", "libuvolcodec_hipemu.so")


@pytest.fixture(scope="session")
def gpu_codec():
    import uvol
    c = uvol.Codec(device=0)          # raises if the HIP library / GPU is missing: no fallback
    yield c
    c.close()
