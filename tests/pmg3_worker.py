"""Worker of tests/test_gpu_pmg3.py: the third multigrid level is switched by an environment variable that the library
reads once, so the checks run in their own process (TLFEA_PMG_LEVELS=3 set by the caller).  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import MATERIALS, fixed_x0, load_mesh, make_gpu, tl  # noqa: E402


def dof_csr(off, cols, vals, n_nodes):
    """DOF-level layout [row][d][k][e] -> scipy CSR"""
    ro = np.zeros(3 * n_nodes + 1, dtype=np.int64)
    ci, va = [], []
    for I in range(n_nodes):
        deg = off[I + 1] - off[I]
        blk = vals[9 * off[I]:9 * off[I + 1]].reshape(3, deg, 3)
        cc = (3 * cols[off[I]:off[I + 1]][:, None] + np.arange(3)[None, :]).reshape(-1)
        for dd in range(3):
            ci.append(cc)
            va.append(blk[dd].reshape(-1))
            ro[3 * I + dd + 1] = ro[3 * I + dd] + 3 * deg
    return sp.csr_matrix((np.concatenate(va), np.concatenate(ci), ro), shape=(3 * n_nodes, 3 * n_nodes))


def main(mesh):
    X, conn = load_mesh(mesh)
    fixed = fixed_x0(X)
    d = make_gpu(X, conn, MATERIALS["svk"], fixed)
    rng = np.random.default_rng(7)
    x = X + rng.normal(0.0, 1e-4, X.shape)
    x[fixed] = X[fixed]
    d.UpdatePositions(x[:, 0].copy(), x[:, 1].copy(), x[:, 2].copy())
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.SetParameters(tl.SyncedNewtonParams(1e-4, 1e-4, 1e-4, 1e14, 5, 10, 1e-3))
    s.AssembleHessian()
    out = dict(mesh=mesh, precond=s.GetPreconditioner())
    par0, par1, c_off, c_cols, Hc = s.RetrievePmgLevel()
    na, nnz3, deg3 = s.GetPmgLevel3Info()
    out.update(n_vertex=int(len(c_off) - 1), n_aggregates=na, nnz3=nnz3, degree3=deg3)
    agg, rvec, active, off3, cols3, H3 = s.RetrievePmgLevel3()
    nc = len(c_off) - 1
    A1 = dof_csr(c_off, c_cols, Hc, nc)
    # P2: u_i = t_A + w_A x r_i ;  W_i0 = I, W_i1 = -[r_i]x
    rows, cols, vals = [], [], []
    for i in range(nc):
        A = agg[i]
        r = rvec[i]
        S = -np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
        for a in range(3):
            rows.append(3 * i + a); cols.append(6 * A + a); vals.append(1.0)
            for b in range(3):
                if S[a, b] != 0.0:
                    rows.append(3 * i + a); cols.append(6 * A + 3 + b); vals.append(S[a, b])
    P2 = sp.csr_matrix((vals, (rows, cols)), shape=(3 * nc, 6 * na))
    ref = (P2.T @ A1 @ P2).toarray()
    for A in np.where(active <= 0)[0]:          # unusable rotations (0) and empty grid cells (-1): identity rows
        ref[6 * A + 3:6 * A + 6, 6 * A + 3:6 * A + 6] = np.eye(3)
    for A in np.where(active < 0)[0]:
        ref[6 * A:6 * A + 3, 6 * A:6 * A + 3] = np.eye(3)
    dev = dof_csr(off3, cols3, H3, 2 * na).toarray()
    out["galerkin_relerr"] = float(np.abs(dev - ref).max() / np.abs(ref).max())
    out["pattern_covers"] = bool(np.all((np.abs(ref) > 0) <= (dof_csr(off3, cols3, np.ones_like(H3), 2 * na).toarray() > 0)))
    out["sizes"] = [int(np.bincount(agg).min()), int(np.bincount(agg).max())]
    out["inactive"] = int((active == 0).sum())
    out["empty_cells"] = int((active < 0).sum())
    sym = np.abs(dev - dev.T).max() / np.abs(dev).max()
    out["symmetry"] = float(sym)
    w = np.linalg.eigvalsh(0.5 * (dev + dev.T))
    out["min_eig_over_max"] = float(w.min() / w.max())
    # solve with the three-level cycle vs the Chebyshev preconditioner
    b = rng.normal(size=3 * X.shape[0])
    res = {}
    for pre in (1, 2):
        s.SetLinSolveOpts(tl.LinSolveOpts(1e-12, 20000, 5, 0, 0.0, 0, pre))
        xs, it, rel = s.LinearSolve(b)
        res[pre] = (xs, it, rel)
    out["its_cheb"], out["its_pmg3"] = int(res[1][1]), int(res[2][1])
    out["rel_pmg3"] = float(res[2][2])
    out["solution_relerr"] = float(np.abs(res[1][0] - res[2][0]).max() / np.abs(res[1][0]).max())
    print(json.dumps(out), flush=True)
    del s
    d.Destroy()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "res4")
