import os, sys, json
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
from oracle import orc
from tests import helpers
from tests.dist_worker import problem
import importlib
par = importlib.import_module("total-lagrangian-fea_amd.partition")
tl = helpers.tl
import torch, torch.distributed as dist
import scipy.sparse as sp
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
X, conn, fixed, f_ext = problem(os.environ.get("DBG_MESH", "box"))
m = helpers.MATERIALS["svk"]
owner = par.slab_owner(X, conn, world)
part = par.partition_from_global(X, conn, owner, rank, world)
fixed_loc = part.localize_nodes(fixed)
f_share = part.share_of_nodal_vector(f_ext)
torch.cuda.set_device(0)
d = helpers.make_gpu(part.X, part.conn, m, fixed_loc, f_share)
s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
s.SetParameters(tl.SyncedNewtonParams(1e-6, 0.0, 1e-6, 1e14, 5, 12, 1e-3))
s.SetLinSolveOpts(tl.LinSolveOpts(1e-13, 20000, 10))
par.attach(s, part, torch, dist)
ng = s.EvalGradient()
g = s.RetrieveGradientToCPU()
s.AssembleHessian()
ro, ci, val = s.RetrieveHessianCSRToCPU()
N = X.shape[0]
Hl = sp.csr_matrix((val, ci, ro), shape=(3*part.X.shape[0],)*2).tocoo()
gd = (3*part.l2g[:,None] + np.arange(3)[None,:]).reshape(-1)
Hg = sp.coo_matrix((Hl.data, (gd[Hl.row], gd[Hl.col])), shape=(3*N,3*N)).tocsr()
b = np.random.default_rng(5).normal(size=3*N)
pv = np.random.default_rng(9).normal(size=3*N)
qv = s.ApplyHessian(pv[gd])
s.SetLinSolveOpts(tl.LinSolveOpts(1e-13, 1, 1))
x1, _, _ = s.LinearSolve(b[gd])
s.SetLinSolveOpts(tl.LinSolveOpts(1e-13, 2, 1))
x2, _, _ = s.LinearSolve(b[gd])
s.SetLinSolveOpts(tl.LinSolveOpts(1e-13, 20000, 10))
xs, iters, rel = s.LinearSolve(b[gd])
objs = [None]*world
dist.all_gather_object(objs, (gd, g, Hg, xs, iters, rel, ng))
o1 = [None]*world
dist.all_gather_object(o1, (x1, x2, part.node_weight, qv, part.iface_nodes))
if rank == 0:
    o = helpers.make_oracle(X, conn, m, fixed, f_ext)
    f_int = o.internal_force(o.v)
    g_ref = o.grad_L(f_int, 1e-3, 1e14)
    ro, ci, val = o.assemble_hessian(1e-3, 1e14)
    H_ref = sp.csr_matrix((val, ci, ro), shape=(3*N,3*N))
    Hsum = sum(o_[2] for o_ in objs)
    print("H err", abs(Hsum - H_ref).max() / abs(H_ref).max())
    for r,(gd_, g_, _, xs_, it_, rel_, ng_) in enumerate(objs):
        print(r, "g err", np.max(np.abs(g_ - g_ref[gd_]))/np.max(np.abs(g_ref)), "ng", ng_, np.linalg.norm(g_ref), "iters", it_, rel_)
    x_ref = orc.solve_spd_upper(ro, ci, val, b)
    Hd = H_ref.toarray()
    n3 = 3*N
    Dinv = np.zeros((n3, n3))
    for i in range(N):
        Dinv[3*i:3*i+3, 3*i:3*i+3] = np.linalg.inv(Hd[3*i:3*i+3, 3*i:3*i+3])
    z = Dinv @ b; q = Hd @ z
    a0 = (b @ z) / (z @ q)
    xe1 = a0 * z
    r1 = b - a0 * q; z1 = Dinv @ r1
    beta = (r1 @ z1) / (b @ z)
    p1 = z1 + beta * z; q1 = Hd @ p1
    a1 = (r1 @ z1) / (p1 @ q1)
    xe2 = xe1 + a1 * p1
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez("gpurun_out/dist_dbg.npz", b=b, **{f"gd{r}": objs[r][0] for r in range(world)},
             **{f"x2_{r}": o1[r][1] for r in range(world)}, **{f"nw{r}": o1[r][2] for r in range(world)})
    for r,(gd_, *_rest) in enumerate(objs):
        x1_, x2_, nw_, qv_, ifn_ = o1[r]
        qe = np.abs(qv_ - (H_ref @ pv)[gd_]) / np.abs(H_ref @ pv).max()
        bad = np.where(qe > 1e-10)[0]
        print(r, "apply err", qe.max(), "bad dofs", bad[:12], "bad node weights", nw_[bad[:12]//3], "n_iface_loc", len(ifn_))
        e1 = np.abs(x1_ - xe1[gd_]); e2 = np.abs(x2_ - xe2[gd_])
        print(r, "iter1 err", e1.max()/np.abs(xe1).max(), "iter2 err", e2.max()/np.abs(xe2).max(),
              "worst dof node weight", nw_[np.argmax(e2)//3], "ratio x1/xe1 median", np.median(x1_/xe1[gd_]))
    for r,(gd_, g_, _, xs_, it_, rel_, ng_) in enumerate(objs):
        print(r, "x err", np.max(np.abs(xs_ - x_ref[gd_]))/np.max(np.abs(x_ref)))
dist.barrier()
