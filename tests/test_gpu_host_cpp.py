"""The signature-preserving C++ host classes (orb-slam3-mac_amd/host) run end to end on the GPU."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_host_cpp_smoke():
    exe = os.path.join(ROOT, "orb-slam3-mac_amd", "lib", "host_smoke")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "HOST_CPP_OK" in r.stdout, r.stdout


def test_host_cpp_built():
    assert os.path.exists(os.path.join(ROOT, "orb-slam3-mac_amd", "lib", "host_smoke"))
