"""CPU: the boundary of the frontend pieces (include/okvis_amd_frontend.h) — header is plain C, the library exports every
entry, arguments are checked before the device is touched, no GPU -> no result — and sanity of the checker itself: the
reference's ProbabilisticStereoTriangulator (oracle/_ref) recovers noise-free points."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_lib as R  # noqa: E402
from okvis_amd import _lib, frontend as F, synthetic  # noqa: E402
from okvis_amd.window import DIST_EQUIDISTANT, DIST_RADTAN  # noqa: E402


def test_header_declares_what_the_library_exports(tmp_path):
    hdr = open(os.path.join(ROOT, "include", "okvis_amd_frontend.h")).read()
    declared = sorted(set(re.findall(r"\b(okvis_fe_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(F.SYMBOLS)
    L = _lib.lib()
    for s in declared:
        getattr(L, s)
    src = tmp_path / "t.c"
    src.write_text('#include "okvis_amd_frontend.h"\n#include <stdio.h>\n'
                   "int main(void) { okvis_fe_context* c = 0; int rc = okvis_fe_create(&c, 0);\n"
                   '  printf("%d %d\\n", rc, (int)sizeof(okvis_fe_camera)); if (rc == 0) okvis_fe_destroy(c); return 0; }\n')
    libdir = os.path.join(ROOT, "okvis_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                           str(tmp_path / "t"), "-L", libdir, "-lokvis_amd_ba", "-Wl,-rpath," + libdir])
    rc, size = (int(x) for x in subprocess.check_output([str(tmp_path / "t")]).split())
    assert size == C.sizeof(F.CameraC) == 112
    import torch
    if not torch.cuda.is_available():
        assert rc == -4                                   # OKVIS_BA_ERR_NO_DEVICE: there is no CPU path


def test_null_context_is_an_argument_error():
    L = _lib.lib()
    F.declare(L)
    cam = F.camera(synthetic.TEST_INTR_RADTAN, DIST_RADTAN)
    assert L.okvis_fe_stereo_triangulate(None, C.byref(cam), C.byref(cam), None, None, 0, None, 0, None, 0, None, None, 0, None, None,
                                         None) == -1
    assert L.okvis_fe_project_landmarks(None, C.byref(cam), None, None, 0, None, None, None, None) == -1
    assert L.okvis_fe_gate_3d2d(None, 0, None, None, 0, None, 0, None, None, None) == -1


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not available")
@pytest.mark.parametrize("model,intr", [(DIST_RADTAN, synthetic.TEST_INTR_RADTAN), (DIST_EQUIDISTANT, synthetic.TEST_INTR_EQUI)])
def test_reference_triangulator_recovers_noise_free_points(model, intr):
    rng = np.random.default_rng(3)
    ref = F.Frontend(api=(R.lib(), "ref_fe_"))
    cam = F.camera(intr, model)
    T_AB = np.array([0.2, 0.01, -0.02, 0, 0, 0, 1.0])
    n = 50
    depth = rng.uniform(1.0, 6.0, n)
    p_A = np.c_[rng.uniform(-0.4, 0.4, n) * depth, rng.uniform(-0.3, 0.3, n) * depth, depth]
    uvA, okA = synthetic.project_points(intr, model, p_A)
    uvB, okB = synthetic.project_points(intr, model, p_A - T_AB[:3])
    assert okA.all() and okB.all()
    kpA, kpB = np.c_[uvA, np.full(n, 8.0)], np.c_[uvB, np.full(n, 8.0)]
    pairs = np.c_[np.arange(n), np.arange(n)]
    hp, cov, flags = ref.stereo_triangulate(cam, cam, T_AB, np.diag([1e-2] * 3 + [1e-8] * 3), kpA, kpB, pairs)
    assert (flags & F.TRI_VALID != 0).all() and (flags & F.TRI_NOT_PARALLEL != 0).all()
    assert np.abs(hp[:, :3] / hp[:, 3:] - p_A).max() < 2e-3          # keypoints are float32: millimetres at metres
    assert (np.linalg.eigvalsh(cov) > 0).all()
    # depth is the uncertain direction: the largest eigenvector of the covariance looks along the ray
    w, v = np.linalg.eigh(cov)
    ray = p_A / np.linalg.norm(p_A, axis=1)[:, None]
    assert (np.abs(np.einsum("ij,ij->i", v[:, :, 2], ray)) > 0.9).all()
