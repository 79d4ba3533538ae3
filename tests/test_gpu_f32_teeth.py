"""The fp32 accuracy gate has teeth: deliberately degraded builds of the library FAIL it.

`make -C minimal-sdr_amd mutants` (part of __graft_entry__.build(); never the product library) compiles msdr_api.hip three more times with
-DMSDR_MUTATE=k:
    1  the lo pieces of the chain's tap fragments zeroed: the xh x Bl product is gone, the taps keep 11 bits  (error ~ 2e-4)
    2  the taps rounded to 16 significant bits before they are split                                          (error ~ 4e-6: INSIDE the
       north-star's 1e-5 -- only the contract's float64 clause, applied to every case, sees it)
    3  time segments start their cascade from zero state: no re-convergence over a warm-up                    (error at segment boundaries)
Each test below runs gate tests in a child process against one mutant (MSDR_LIB) and asserts that the child PASSES -- marked
xfail(strict=True): the expected outcome is a failure of the gate, and a mutant that slips through turns this test red.  The
a-priori error model behind each clause of the contract is DESIGN.md 5."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MUT = {1: ["tests/test_gpu_f32_contract.py::test_fp32_contract_on_the_fuzzers_cases"],
       2: ["tests/test_gpu_f32_contract.py::test_fp32_contract_on_the_fuzzers_cases"],
       3: ["tests/test_gpu_chain.py::test_chain_f32_time_segments_vs_sequential"]}


def _run(lib, tests):
    env = dict(os.environ, MSDR_LIB=lib)
    return subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)


def test_the_product_library_passes_the_same_gate_in_a_child_process():
    """The control: the very same child invocation with the product library is green (so a red mutant run is the mutant's doing)."""
    r = _run(os.path.join(ROOT, "minimal-sdr_amd", "lib", "libmsdr.so"), sorted({t for ts in MUT.values() for t in ts}))
    assert r.returncode == 0, r.stdout[-3000:]


@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.xfail(strict=True, reason="a degraded build must NOT pass the accuracy gate")
def test_a_degraded_build_passes_the_gate(k):
    lib = os.path.join(ROOT, "minimal-sdr_amd", "lib_mut%d" % k, "libmsdr.so")
    if not os.path.exists(lib):
        pytest.skip("mutant %d not built (make -C minimal-sdr_amd mutants)" % k)
    r = _run(lib, MUT[k])
    print(r.stdout[-1500:])
    assert "passed" in r.stdout or "failed" in r.stdout, r.stdout[-2000:]       # the child really ran the tests (an import error is not a verdict)
    assert r.returncode == 0
