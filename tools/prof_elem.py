#!/usr/bin/env python3
"""Profiling driver: one engine of a BASELINE config, one Newton iteration, then every hot kernel launched `reps` times
back to back (tlfea_newton_time_kernels).  Run under rocprofv3 (--kernel-trace --stats, or --pmc ... in separate passes):
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o run --output-format csv -- python3 tools/prof_elem.py C 3
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tl = importlib.import_module("total-lagrangian-fea_amd")
wl = importlib.import_module("total-lagrangian-fea_amd.workloads")

cfg = sys.argv[1] if len(sys.argv) > 1 else "C"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = wl.build(cfg)
d, s = wl.make_engine(tl, w)
s.SetLinSolveOpts(tl.LinSolveOpts(1e-12, 300, 25, on_unconverged=1))
d.UpdatePositions(w["x0"][:, 0], w["x0"][:, 1], w["x0"][:, 2])
s.BeginStep()
print("newton iteration:", s.NewtonIteration())
print({k: round(v * 1e3, 1) for k, v in s.TimeKernels(reps=reps).items()}, "us per launch")
del s
d.Destroy()
