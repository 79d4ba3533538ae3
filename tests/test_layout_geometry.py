"""The LDS layout arithmetic the kernels and the host's table builders share -- the compact B operand's copy stride (msdr_chain_mfw.hiph
mw_compact_stride: room for the prefetch, and the 16 lanes of an LDS pass in 16 different 16-byte slots, for EVERY halo), the block-cadence
tile geometry and window swizzle (msdr_chain_mfb.hiph), the Q15 block kernel's LDS sizes -- checked on the host by tests/cpp/geometry_check.hip,
which includes the kernel headers as they are and launches nothing."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc missing")
def test_layout_arithmetic_of_the_kernel_headers():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "geometry_check"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "geometry_check")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "geometry_check: 0 failures" in r.stdout, r.stdout[-3000:]
