"""A numpy statement of the chain solve of okvis_amd/csrc/ba_chain.hpp — the same data flow (two sweeps towards the middle block,
L^-1 / D^-1 / (D^-1 R_C)^T published per block, one column of the pose couplings per lane, the pose system's Schur complement in
the fixed block order, back-substitution from the middle outwards) in plain loops, so that the ALGORITHM is checked on the CPU
against numpy.linalg.solve (tests/test_chain_emulation.py) and the kernel against both on the GPU (tests/test_gpu_chain_solve.py).
Test infrastructure."""
import numpy as np


def chain_structured_system(rng, n_pose, n_sb, pose_prior=1e16, coupled_poses=2, scale=1.0):
    """A random SPD system with the structure of a window's reduced camera system: n_pose dense 6-blocks, then n_sb 9-blocks in a
    chain (block k couples to k - 1, k + 1 and to poses); one pose carries a 1e16 yaw-like prior."""
    Dp, D = 6 * n_pose, 6 * n_pose + 9 * n_sb
    J = rng.standard_normal((3 * D, D))
    # the pose part: dense
    H = np.zeros((D, D))
    Jp = rng.standard_normal((4 * Dp + 8, Dp))
    H[:Dp, :Dp] = Jp.T @ Jp
    for k in range(n_sb):
        rows = list(range(Dp + 9 * k, Dp + 9 * k + 9))
        cols = list(rows)
        if k + 1 < n_sb:
            cols += list(range(Dp + 9 * (k + 1), Dp + 9 * (k + 1) + 9))
        # the poses an IMU term of frame k touches: frame k and k + 1 (when there are that many poses)
        for pidx in range(coupled_poses):
            p = min(k + pidx, n_pose - 1)
            cols += list(range(6 * p, 6 * p + 6))
        cols = sorted(set(cols))
        Jk = rng.standard_normal((15, len(cols))) * scale
        H[np.ix_(cols, cols)] += Jk.T @ Jk
    H[np.arange(Dp, D), np.arange(Dp, D)] += 1e-3
    if pose_prior:
        n = rng.standard_normal(3)
        n /= np.linalg.norm(n)
        H[3:6, 3:6] += pose_prior * np.outer(n, n)
        H[0:3, 0:3] += 1e4 * np.eye(3)
    H = 0.5 * (H + H.T)
    g = rng.standard_normal(D)
    return H, g, Dp


def is_chain_structured(H, Dp):
    D = H.shape[0]
    Ks = (D - Dp) // 9
    for a in range(Ks):
        for b in range(a):
            if a - b > 1 and np.any(H[Dp + 9 * a:Dp + 9 * a + 9, Dp + 9 * b:Dp + 9 * b + 9] != 0):
                return False
    return True


def chain_solve(H, g, Dp):
    """x with H x = g by the steps of ba_chain.hpp::chain_solve"""
    D = H.shape[0]
    Ks = (D - Dp) // 9
    nL, nR, mid = Ks // 2, (Ks - 1) // 2, Ks // 2
    sb = lambda k: slice(Dp + 9 * k, Dp + 9 * k + 9)
    A = [H[sb(k), sb(k)].copy() for k in range(Ks)]
    N = [np.c_[H[sb(k), :Dp], g[sb(k)]].copy() for k in range(Ks)]      # pose couplings | rhs
    P, Dinv, RC, Tt = [None] * Ks, [None] * Ks, [None] * Ks, [None] * Ks

    def ldl(Ak):
        n = Ak.shape[0]
        L, d = np.eye(n), np.zeros(n)
        W = Ak.copy()
        for K in range(n):
            d[K] = W[K, K]
            L[K + 1:, K] = W[K + 1:, K] / d[K]
            W[K + 1:, K + 1:] -= np.outer(L[K + 1:, K], W[K, K + 1:])
        return L, d

    contrib_A = np.zeros((9, 9))
    contrib_N = np.zeros((9, Dp + 1))
    for side in (0, 1):
        for t in range(nL if side == 0 else nR):
            k = t if side == 0 else Ks - 1 - t
            n = k + 1 if side == 0 else k - 1
            M = H[sb(k), sb(n)]                         # rows: block k, columns: block n
            L, d = ldl(A[k])
            P[k], Dinv[k] = np.linalg.inv(L), 1.0 / d
            RC[k] = P[k] @ M
            Tt[k] = (Dinv[k][:, None] * RC[k]).T        # (D^-1 R_C)^T
            Y = P[k] @ N[k]
            upd_A, upd_N = Tt[k] @ RC[k], Tt[k] @ Y
            N[k] = Y
            if side == 1 and n == mid:
                contrib_A -= upd_A
                contrib_N -= upd_N
            else:
                A[n] = A[n] - upd_A
                N[n] = N[n] - upd_N
    A[mid] = A[mid] + contrib_A
    N[mid] = N[mid] + contrib_N
    L, d = ldl(A[mid])
    P[mid], Dinv[mid] = np.linalg.inv(L), 1.0 / d
    N[mid] = P[mid] @ N[mid]
    # the pose system
    Spp = H[:Dp, :Dp].copy()
    rp = g[:Dp].copy()
    for k in range(Ks):
        Y = N[k]
        Spp -= Y[:, :Dp].T @ (Dinv[k][:, None] * Y[:, :Dp])
        rp -= Y[:, :Dp].T @ (Dinv[k] * Y[:, Dp])
    xp = np.linalg.solve(Spp, rp)
    x = np.zeros(D)
    x[:Dp] = xp
    u = [N[k][:, Dp] - N[k][:, :Dp] @ xp for k in range(Ks)]
    x[sb(mid)] = P[mid].T @ (Dinv[mid] * u[mid])
    for side in (0, 1):
        for t in reversed(range(nL if side == 0 else nR)):
            k = t if side == 0 else Ks - 1 - t
            n = k + 1 if side == 0 else k - 1
            x[sb(k)] = P[k].T @ (Dinv[k] * (u[k] - RC[k] @ x[sb(n)]))
    return x
