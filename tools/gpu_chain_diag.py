"""Diagnostics (not a test): where the chain solver's solution differs from numpy's, per block, over a set of shapes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.chain_emulation import chain_structured_system
from tests.test_gpu_chain_solve import reduced_solve
from okvis_amd.window import SOLVE_CHAIN, SOLVE_DENSE
for n_pose, n_sb in [(1, 1), (2, 1), (2, 2), (3, 3), (4, 4), (5, 1), (10, 2), (10, 3), (10, 10), (3, 5)]:
    rng = np.random.default_rng(1000 * n_pose + n_sb)
    H, g, Dp = chain_structured_system(rng, n_pose, n_sb, pose_prior=0.0)
    xr = np.linalg.solve(H, g)
    xc, _, fc = reduced_solve(H, g, Dp, SOLVE_CHAIN)
    sc = np.abs(xr).max()
    ep = np.abs(xc[:Dp] - xr[:Dp]).max() / sc
    es = [np.abs(xc[Dp + 9 * k:Dp + 9 * k + 9] - xr[Dp + 9 * k:Dp + 9 * k + 9]).max() / sc for k in range(n_sb)]
    print(f"poses {n_pose} sb {n_sb} fail {fc}: pose part {ep:.1e}   sb blocks " + " ".join(f"{e:.0e}" for e in es), flush=True)

# which term of the back-substitution is off: the two-block case against the pieces of the numpy statement
import numpy as np
from tests import chain_emulation as E
rng = np.random.default_rng(2002)
H, g, Dp = chain_structured_system(rng, 2, 2, pose_prior=0.0)
xr = np.linalg.solve(H, g)
xc, _, _ = reduced_solve(H, g, Dp, SOLVE_CHAIN)
A0 = H[Dp:Dp + 9, Dp:Dp + 9]; M = H[Dp:Dp + 9, Dp + 9:Dp + 18]
L = np.linalg.cholesky(A0); d = np.diag(L) ** 2; L = L / np.diag(L)[None, :]
P = np.linalg.inv(L); Dinv = 1.0 / d
N0 = np.c_[H[Dp:Dp + 9, :Dp], g[Dp:Dp + 9]]
Y = P @ N0
u = Y[:, Dp] - Y[:, :Dp] @ xr[:Dp]
RC = P @ M
x1 = xr[Dp + 9:Dp + 18]
full = P.T @ (Dinv * (u - RC @ x1))
noG = P.T @ (Dinv * u)
plusG = P.T @ (Dinv * (u + RC @ x1))
GT = P.T @ (Dinv[:, None] * RC)
print("block 0: GPU - full %.1e   GPU - (no G term) %.1e   GPU - (+G) %.1e   GPU - (G transposed) %.1e" % (
    np.abs(xc[Dp:Dp + 9] - full).max(), np.abs(xc[Dp:Dp + 9] - noG).max(), np.abs(xc[Dp:Dp + 9] - plusG).max(),
    np.abs(xc[Dp:Dp + 9] - (noG - GT.T @ x1)).max()))

# the solver's LDS image against the numpy pieces (layout: LChain::make, ba_chain.hpp)
def ldl16_nb(D): return (D + 1 + 15) // 16
def area(D):
    nb = ldl16_nb(D); return max(nb * (nb + 1) // 2 * 256, 2 * nb * 256 + 3 * nb * 288 + 4 * 256 + 2 * nb * 16 + 288)
D = H.shape[0]; Ks = (D - Dp) // 9; NP = (Dp + 2) & ~1; CB = 90; nb = ldl16_nb(Dp)
oA = nb * (nb + 1) // 2 * 256; oC = oA + Ks * CB; oT = oC + max(Ks - 1, 1) * 2 * CB; oX = oT + Ks * CB; dead = oX + CB + 9 * NP
oN = (max(dead, area(Dp)) + 1) & ~1; oP = oN + Ks * 9 * NP; oG = oP + Ks * CB; oDi = oG + Ks * CB
img = np.zeros(20000)
xc, _, _ = reduced_solve(H, g, Dp, SOLVE_CHAIN, dump=img)
Pg = img[oP:oP + CB].reshape(9, 10)[:, :9].T            # Pcol[j][i] = P[i][j]
print("P (strictly lower) err %.1e" % np.abs(np.tril(Pg, -1) - np.tril(P, -1)).max())
print("dinv err %.1e" % np.abs(img[oDi:oDi + 9] - Dinv).max())
Gg = img[oG:oG + CB].reshape(9, 10)[:, :9]
Gw = P.T @ (Dinv[:, None] * RC)
print("G err %.1e (|G| %.1e)" % (np.abs(Gg - Gw).max(), np.abs(Gw).max()))
Yg = img[oN:oN + 9 * NP].reshape(9, NP)[:, :Dp + 1]
print("Y err %.1e" % np.abs(Yg - Y).max())
print("T region now (dead after the pose solve):", np.abs(img[oT:oT + CB]).max())
