"""GPU: the okvis::Estimator drop-in (okvis_amd/csrc/host/okvis_estimator_adapter.hpp) compiled against the reference's REAL
okvis headers (oracle/ref/adapter_runtime.cpp -> oracle/_ref/adapter_runtime; third-party headers from oracle/shim) and driven
through VioBackendInterface the way ThreadedKFVio does, over a sliding window in which frames and landmarks are marginalised
(the path on which the adapter used to throw)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "adapter_runtime")


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/adapter_runtime not built (needs the reference tree at build time)")
def test_adapter_runs_a_sliding_window_with_marginalisation():
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    assert "ADAPTER RUNTIME OK" in r.stdout
    assert "removed" in r.stdout


REF_TEST = os.path.join(ROOT, "oracle", "_ref", "reference_test_estimator")


@pytest.mark.skipif(not os.path.exists(REF_TEST), reason="oracle/_ref/reference_test_estimator not built (needs the reference tree at build time)")
def test_reference_test_estimator_source_passes_on_the_backend():
    """The reference's OWN okvis_ceres/test/TestEstimator.cpp, compiled unmodified with <okvis/Estimator.hpp> resolved to the
    drop-in (oracle/ref/Makefile): four extrinsics cases, 7 frames of 1763 landmarks added up front (most of them unobserved
    at first), optimize(10) per frame, applyMarginalizationStrategy(2, 3), and its three accuracy assertions
    (TestEstimator.cpp:229-236: speed/bias 0.04, rotation 1e-2, translation 0.1)."""
    r = subprocess.run([REF_TEST], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, (r.returncode, r.stdout[-2500:], r.stderr[-1500:])
    assert "1 tests, 0 failed" in r.stdout and r.stdout.count("== LAST OPTIMIZATION ==") == 4
