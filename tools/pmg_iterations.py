import numpy as np, sys
sys.path.insert(0, '.')
from tests.helpers import *
from oracle import orc
for tag in ("res2", "res4", "bunny"):
    try:
        X, conn = load_mesh(tag)
    except Exception as e:
        print(tag, "skip", e); continue
    fixed = fixed_x0(X)
    d = make_gpu(X, conn, MATERIALS["svk"], fixed)
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.SetParameters(tl.SyncedNewtonParams(1e-4, 1e-4, 1e-4, 1e14, 5, 10, 1e-3))
    s.AssembleHessian()
    b = np.random.default_rng(3).normal(size=3 * X.shape[0])
    out = {}
    for pre in (1, 2):
        s.SetLinSolveOpts(tl.LinSolveOpts(1e-12, 20000, 5, 0, 0.0, 0, pre))
        x, it, rel = s.LinearSolve(b)
        out[pre] = (it, rel)
    print(tag, conn.shape[0], "elements: chebyshev", out[1], "pmg", out[2], "pmg info", s.GetPmgInfo())
    del s; d.Destroy()
