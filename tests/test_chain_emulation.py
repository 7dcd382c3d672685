"""CPU: the chain solve of ba_chain.hpp as a numpy statement (tests/chain_emulation.py) against numpy.linalg.solve — the algorithm,
on the structures the index build sends to it; and the host-side decision which windows are sent (okvis_ba_check_window_lists)."""
import numpy as np
import pytest

from tests.chain_emulation import chain_solve, chain_structured_system, is_chain_structured


@pytest.mark.parametrize("n_pose,n_sb", [(10, 10), (8, 3), (4, 4), (1, 1), (2, 2), (5, 1), (3, 5), (12, 10), (20, 6), (6, 11)])
def test_chain_solve_is_a_solve(n_pose, n_sb):
    rng = np.random.default_rng(100 * n_pose + n_sb)
    for prior in (0.0, 1e10):
        H, g, Dp = chain_structured_system(rng, n_pose, n_sb, pose_prior=prior)
        assert is_chain_structured(H, Dp)
        x = chain_solve(H, g, Dp)
        xr = np.linalg.solve(H, g)
        # (both are backward stable: compare residuals, and the solutions where the conditioning allows)
        res = lambda v: np.abs(H @ v - g).max() / (np.abs(H).max() * np.abs(v).max() + np.abs(g).max())
        assert res(x) < 1e-13 and res(x) < 50 * res(xr) + 1e-15
        if prior == 0.0:
            assert np.abs(x - xr).max() <= 1e-11 * np.abs(xr).max()


def test_which_windows_go_to_the_chain_solver():
    from okvis_amd import solver, synthetic
    from okvis_amd.window import SOLVE_CHAIN, SOLVE_DENSE, default_options, set_options
    w = synthetic.make_window(10, 40, 1.0, 5)
    assert solver.index_lists(w)["chain"] == 10
    assert solver.index_lists(w, set_options(default_options(), tuning_solve_mode=SOLVE_DENSE))["chain"] == 0
    # an IMU term between frames that are not neighbours in the reduced order: no chain
    w2 = synthetic.make_window(6, 40, 1.0, 6)
    w2.imu_sb1 = np.asarray(w2.imu_sb1).copy()
    w2.imu_pose1 = np.asarray(w2.imu_pose1).copy()
    w2.imu_sb1[0], w2.imu_pose1[0] = 3, 3
    assert solver.index_lists(w2, set_options(default_options(), tuning_solve_mode=SOLVE_CHAIN))["chain"] == 0
    # fixed speed/bias blocks drop out of the chain: the free ones still have to be neighbours
    w3 = synthetic.make_window(6, 40, 1.0, 7)
    w3.sb_fixed = np.array([1, 0, 0, 0, 0, 0], np.uint8)
    assert solver.index_lists(w3, set_options(default_options(), tuning_solve_mode=SOLVE_CHAIN))["chain"] == 5
    assert solver.index_lists(w3)["chain"] == 0      # OKVIS_BA_SOLVE_AUTO: the chain solver from eight blocks on (where it is the faster one)
