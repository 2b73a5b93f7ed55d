import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the shared libraries exist (build is incremental; seconds when up to date)."""
    import __graft_entry__ as g
    need = [os.path.join(ROOT, "orb-slam3-mac_amd", "lib", "liborbhip.so"),
            os.path.join(ROOT, "orb-slam3-mac_amd", "lib", "libsynth.so"),
            os.path.join(ROOT, "oracle", "liborb_oracle.so")]
    import glob

    def stale(lib, patterns):
        if not os.path.exists(lib):
            return True
        t = os.path.getmtime(lib)
        return any(os.path.getmtime(f) > t for pat in patterns for f in glob.glob(os.path.join(ROOT, pat)))
    if (stale(need[0], ["orb-slam3-mac_amd/csrc/*", "include/*.h"]) or stale(need[1], ["orb-slam3-mac_amd/synth/*"]) or
            stale(need[2], ["oracle/*.c", "oracle/*.h", "oracle/*.inc"])):
        g.build()                      # incremental (make): only what is out of date
    yield


@pytest.fixture(scope="session")
def gpu_ctx():
    import orbhip
    ctx = orbhip.Context(0)
    yield ctx
    ctx.close()
