#!/usr/bin/env python3
"""Host-side cost of one exchange of the partitioned path as bench.py runs it (torch.distributed 'nccl' all_reduce called
from the engine's callback): config B on ONE rank with a declared interface plane of multiplicity 1, so every collective
is an identity all-reduce.  Compares the Newton iteration with and without the interface (both on the Chebyshev path the
multi-GPU runs use) and divides the difference by the number of collectives.
  python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/exchange_overhead.py"""
import importlib
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tl = importlib.import_module("total-lagrangian-fea_amd")
wl = importlib.import_module("total-lagrangian-fea_amd.workloads")
par = importlib.import_module("total-lagrangian-fea_amd.partition")


def run(with_iface, comm=None):
    w = wl.build("B")
    d, s = wl.make_engine(tl, w)
    s.SetLinSolveOpts(tl.LinSolveOpts(1e-12, 5000, 25, 0, 0.0, 0, 1))  # Chebyshev, as with an interface
    if with_iface:
        X = w["X"]
        plane = np.where(np.abs(X[:, 0] - 0.5) < 1e-9)[0].astype(np.int32)
        part = par.Partition(0, 1, X, None, np.arange(X.shape[0]), plane, np.arange(len(plane)), len(plane),
                             np.ones(X.shape[0]))
        par.attach(s, part, torch, dist, native_rccl=comm)
    d.UpdatePositions(w["x0"][:, 0], w["x0"][:, 1], w["x0"][:, 2])
    for i in range(6):
        if i % 3 == 0:
            s.BeginStep()
        s.NewtonIteration()
    torch.cuda.synchronize()
    c0 = s.n_collectives if (with_iface and comm is None) else 0
    t0 = time.perf_counter()
    its = []
    for i in range(12):
        if i % 3 == 0:
            s.BeginStep()
        its.append(s.NewtonIteration()[1])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 12
    ncoll = (s.n_collectives - c0) / 12 if (with_iface and comm is None) else 0
    del s
    d.Destroy()
    return ms, float(np.mean(its)), ncoll


if __name__ == "__main__":
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    a = run(False)
    b = run(True)
    print("no interface   : %.3f ms per Newton iteration, %.1f CG iterations" % a[:2])
    print("with interface : %.3f ms per Newton iteration, %.1f CG iterations, %.0f collectives" % b)
    print("per collective : %.1f us (1-rank identity all-reduce + callback)" % ((b[0] - a[0]) * 1e3 / max(1.0, b[2])))
    comm = par.rccl_communicator(dist, 0, 1)
    c = run(True, comm)
    print("built-in RCCL  : %.3f ms per Newton iteration, %.1f CG iterations -> %.1f us per collective"
          % (c[0], c[1], (c[0] - a[0]) * 1e3 / max(1.0, b[2])))
    par.rccl_destroy(comm)
    dist.destroy_process_group()
