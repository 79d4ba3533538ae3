"""The C++ operator-graph runtime (minimal-sdr_amd/host: AudioStream / AudioConnection / nodes) --
built with g++ against libmsdr.so, exercised by tests/cpp/test_graph.cpp which checks every block
bit-exact against the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    subprocess.check_call(["make", "-C", CPP, "-s"])
    return os.path.join(CPP, "test_graph")


def test_graph_runtime_builds_and_refuses_to_run_without_gpu():
    exe = _build()
    out = subprocess.run([exe, "--no-gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_reference_graph_bit_exact_on_gpu():
    exe = _build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK (graphs A, B, C, D bit-exact vs oracle)" in out.stdout
