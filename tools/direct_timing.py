#!/usr/bin/env python3
"""Times of the sparse direct solve (the engine's multifrontal Cholesky; TLFEA_DIRECT_BACKEND=rocsolver for rocSOLVER csrrf)
against the iterative solve on the same H: set-up (ordering, plan, allocation), re-factorisation + solve per call.
usage: python3 tools/direct_timing.py [big]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tl = importlib.import_module("total-lagrangian-fea_amd")
from tests.helpers import MATERIALS, fixed_x0, load_mesh, make_gpu  # noqa: E402

cases = [("beam_3x2x1", None), ("res2", None), ("bunny", None), ("box 6x6x12", (6, 6, 12)), ("config B", (12, 12, 12))]
if "big" in sys.argv or "only-big" in sys.argv:
    cases += [("box 20^3", (20, 20, 20)), ("box 30^3", (30, 30, 30)), ("bar 60x20x20", (60, 20, 20))]
if "only-big" in sys.argv:
    cases = cases[-3:-1]
if "only-b" in sys.argv:
    cases = cases[4:5]
os.environ.setdefault("TLFEA_DIRECT_TRACE", "1")
for name, cells in cases:
    if cells:
        X, conn = tl.mesh_utils.structured_t10_box(*cells)
        fixed = np.where(X[:, 2] < 1e-12)[0].astype(np.int32)
    else:
        X, conn = load_mesh(name)
        fixed = fixed_x0(X)
    d = make_gpu(X, conn, MATERIALS["svk"], fixed)
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.SetParameters(tl.SyncedNewtonParams(1e-4, 1e-6, 1e-4, 1e14, 5, 10, 1e-3))
    s.AssembleHessian()
    b = np.random.default_rng(3).normal(size=3 * X.shape[0])
    s.SetLinSolveOpts(tl.LinSolveOpts(1e-12, 20000, 10))
    s.LinearSolve(b)
    t0 = time.perf_counter(); _, it, rel = s.LinearSolve(b); t_it = time.perf_counter() - t0
    s.SetLinSolveOpts(tl.LinSolveOpts(method=1))
    t0 = time.perf_counter(); s.LinearSolve(b); t_first = time.perf_counter() - t0
    t0 = time.perf_counter(); _, _, rel_d = s.LinearSolve(b); t_re = time.perf_counter() - t0
    t0 = time.perf_counter(); _, _, rel_d = s.LinearSolve(b); t_re = min(t_re, time.perf_counter() - t0)
    print(f"{name}: {3 * X.shape[0]} DOF | iterative {t_it * 1e3:.1f} ms ({it} CG iterations, rel {rel:.1e}) | direct: first call "
          f"{t_first:.2f} s (ordering + plan + allocation + factor), re-factor + solve {t_re * 1e3:.1f} ms (rel {rel_d:.1e})", flush=True)
    del s
    d.Destroy()
