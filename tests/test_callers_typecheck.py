"""The reference's CALLERS of okvis::Estimator type-checked against the drop-in: ThreadedKFVio.hpp:420 holds an
okvis::Estimator by value, Frontend.hpp:106,327 take okvis::Estimator&, VioKeyframeWindowMatchingAlgorithm.hpp:81,172 store a
pointer, the two OpenGV adapters read poses / landmarks / extrinsics through it.  Their translation units, unmodified and where they
lie under the reference tree, go through `g++ -fsyntax-only` (full semantic analysis, templates instantiated) with
<okvis/Estimator.hpp> resolved to oracle/ref/product_shadow/ — the product's adapter — in front of the reference's own header, the
reference's real OKVIS headers for everything else, and declaration-only stand-ins for the third-party libraries that are not
installed here (Eigen / ceres / glog / OpenCV: oracle/shim; Boost, BRISK, OpenGV, OpenCV GUI: oracle/shim/callers).

Needs the reference tree (this container; not the GPU box): skipped where it is absent."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("OKVIS_REFERENCE", "/root/reference")
MODULES = ["okvis_util", "okvis_kinematics", "okvis_time", "okvis_cv", "okvis_common", "okvis_ceres", "okvis_timing", "okvis_matcher",
           "okvis_frontend", "okvis_multisensor_processing"]
CALLERS = ["okvis_multisensor_processing/src/ThreadedKFVio.cpp",
           "okvis_frontend/src/Frontend.cpp",
           "okvis_frontend/src/VioKeyframeWindowMatchingAlgorithm.cpp",
           "okvis_frontend/src/FrameNoncentralAbsoluteAdapter.cpp",
           "okvis_frontend/src/FrameRelativeAdapter.cpp"]

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "okvis_ceres")) or shutil.which("g++") is None,
                                reason="needs the reference tree and g++")


def _flags():
    inc = [os.path.join(ROOT, "oracle", "ref", "product_shadow"), os.path.join(ROOT, "oracle", "shim", "callers"),
           os.path.join(ROOT, "oracle", "shim")] + [os.path.join(REF, m, "include") for m in MODULES] + \
          [os.path.join(ROOT, "okvis_amd", "csrc", "host"), os.path.join(ROOT, "include")]
    return ["g++", "-std=gnu++14", "-fsyntax-only", "-w"] + ["-I" + i for i in inc]


@pytest.mark.parametrize("unit", CALLERS)
def test_caller_compiles_against_the_drop_in(unit):
    src = os.path.join(REF, unit)
    r = subprocess.run(_flags() + [src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # ... and it was the drop-in it saw, not the reference's own Estimator.hpp
    d = subprocess.run(_flags()[:2] + ["-M", "-w"] + _flags()[4:] + [src], capture_output=True, text=True, timeout=600)
    deps = d.stdout.replace("\\\n", " ").split()
    est = [p for p in deps if p.endswith("okvis/Estimator.hpp")]
    assert est and all("product_shadow" in p for p in est), est
    assert any(p.endswith("okvis_estimator_adapter.hpp") for p in deps)
    assert not any(p.endswith("okvis_ceres/include/okvis/Estimator.hpp") for p in deps)


def test_the_check_is_a_real_check(tmp_path):
    """Negative control: a caller that uses something okvis::Estimator does not have is refused by the same command."""
    bad = tmp_path / "bad_caller.cpp"
    bad.write_text("#include <okvis/Estimator.hpp>\n"
                   "int f(okvis::Estimator& e) { return e.thisMemberDoesNotExist(3); }\n")
    r = subprocess.run(_flags() + [str(bad)], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "thisMemberDoesNotExist" in r.stderr
    good = tmp_path / "good_caller.cpp"
    good.write_text("#include <okvis/Estimator.hpp>\n"
                    "size_t f(okvis::Estimator& e) { e.optimize(3, 1, false); return e.numFrames() + e.numLandmarks(); }\n")
    r = subprocess.run(_flags() + [str(good)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
