#!/usr/bin/env python3
"""BASELINE config 3 literally: the ~1 M-element T10 bar (config C: 972 000 elements, 4.0 M DOF) with the sparse DIRECT
solve on one MI355X -- the engine's multifrontal Cholesky with the top of the dissection tree taken front by front (stack of
update matrices) so that the 118 GB factor and its workspaces fit the 288 GB of HBM.  usage: python3 tools/direct_config_c.py"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TLFEA_DIRECT_TRACE", "1")
tl = importlib.import_module("total-lagrangian-fea_amd")
wl = importlib.import_module("total-lagrangian-fea_amd.workloads")
import torch  # noqa: E402  (device memory query only)

cfg = sys.argv[1] if len(sys.argv) > 1 else "C"
t0 = time.time()
w = wl.build(cfg)
d, s = wl.make_engine(tl, w)
d.UpdatePositions(w["x0"][:, 0], w["x0"][:, 1], w["x0"][:, 2])
print(f"config {cfg}: {w['conn'].shape[0]} elements, {3 * w['X'].shape[0]} DOF; mesh + engine set-up {time.time() - t0:.1f} s", flush=True)
s.BeginStep()
s.AssembleHessian()
n = 3 * w["X"].shape[0]
b = np.random.default_rng(3).normal(size=n)
s.SetLinSolveOpts(tl.LinSolveOpts(1e-12, 20000, 25))
s.LinearSolve(b)
t0 = time.perf_counter(); xi, it, rel_i = s.LinearSolve(b); t_it = time.perf_counter() - t0
print(f"p-multigrid CG: {t_it * 1e3:.1f} ms, {it} iterations, rel {rel_i:.1e}", flush=True)
s.SetLinSolveOpts(tl.LinSolveOpts(method=1))
t0 = time.perf_counter(); xd, _, rel_d = s.LinearSolve(b); t_first = time.perf_counter() - t0
print(f"direct, first call (ordering + plan + allocation + factor + solve): {t_first:.1f} s, rel {rel_d:.1e}", flush=True)
free, total = torch.cuda.mem_get_info()
print(f"device memory in use: {(total - free) / 1e9:.1f} GB of {total / 1e9:.1f} GB", flush=True)
t0 = time.perf_counter(); xd, _, rel_d = s.LinearSolve(b); t_re = time.perf_counter() - t0
print(f"direct, re-factor + solve: {t_re:.2f} s, rel {rel_d:.1e}; |x_direct - x_cg| / |x_cg| = {np.linalg.norm(xd - xi) / np.linalg.norm(xi):.1e}",
      flush=True)
# one Newton iteration with the direct solve (what BASELINE config 3 names)
t0 = time.perf_counter(); ng, _ = s.NewtonIteration(); t_n = time.perf_counter() - t0
print(f"Newton iteration with the direct solve: {t_n:.2f} s (|g| {ng:.3e}) = {w['conn'].shape[0] / t_n:.3e} element-updates/s", flush=True)
