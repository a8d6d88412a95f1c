#!/usr/bin/env python3
"""Newton iterations on the reference's TetGen meshes (SURVEY 8(d) real-mesh sanity set): ms per Newton iteration incl. the
linear solve, CG iterations, cycle in use, and the sparse direct solve on the same H.  usage: python3 tools/real_mesh_timing.py"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tl = importlib.import_module("total-lagrangian-fea_amd")
from tests.helpers import MATERIALS, fixed_x0, load_mesh, make_gpu  # noqa: E402

for tag, mat in (("res4", "svk"), ("res8", "svk"), ("res16", "svk"), ("teapot", "svk"), ("teapot", "neo"), ("bunny", "neo")):
    X, conn = load_mesh(tag)
    if tag in ("teapot", "bunny"):
        fixed = np.where(X[:, 2] < X[:, 2].min() + 0.05 * (X[:, 2].max() - X[:, 2].min()))[0].astype(np.int32)
    else:
        fixed = fixed_x0(X)
    f_ext = np.zeros(3 * X.shape[0])
    top = np.where(X[:, 0] > X[:, 0].max() - 0.02 * (X[:, 0].max() - X[:, 0].min()))[0]
    f_ext[3 * top + 2] = -500.0 / len(top)
    d = make_gpu(X, conn, MATERIALS[mat], fixed, f_ext)
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.Setup()
    s.SetParameters(tl.SyncedNewtonParams(1e-4, 1e-4, 1e-4, 1e14, 5, 10, 1e-3))
    its, ms = [], []
    for k in range(18):
        if k % 3 == 0:
            s.BeginStep()
        t0 = time.perf_counter()
        _, it = s.NewtonIteration()
        ms.append((time.perf_counter() - t0) * 1e3)
        its.append(it)
    cyc = s.GetPmgCycleInfo()
    b = np.random.default_rng(3).normal(size=3 * X.shape[0])
    s.SetLinSolveOpts(tl.LinSolveOpts(method=1))
    s.LinearSolve(b)
    t0 = time.perf_counter(); _, _, rel = s.LinearSolve(b); t_dir = (time.perf_counter() - t0) * 1e3
    print(f"{tag} ({conn.shape[0]} T10, {3 * X.shape[0]} DOF, {mat}): {np.median(ms[3:]):.2f} ms per Newton iteration (median of 15; max {np.max(ms[3:]):.1f}), "
          f"{np.mean(its[3:]):.1f} CG iterations ({cyc['levels']}-level cycle, vertex-level polynomial degree {cyc['vertex_degree']}, "
          f"{cyc['level3_nodes']} level-3 nodes); direct re-factor + solve {t_dir:.1f} ms (rel {rel:.1e})",
          flush=True)
    del s
    d.Destroy()
