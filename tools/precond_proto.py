"""CPU prototype (SciPy) to choose the preconditioner of the device PCG before writing HIP code.
Builds H of a BASELINE config with the oracle and counts CG iterations to rel 1e-12 for candidates."""
import sys, time, importlib
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, '.')
from oracle import orc
wl = importlib.import_module("total-lagrangian-fea_amd.workloads")
tl_mesh = importlib.import_module("total-lagrangian-fea_amd.mesh_utils")

def build(config, cells=None):
    w = wl.build(config, cells=cells)
    m = w["material"]
    mat = orc.svk(m["E"], m["nu"], rho0=m["rho0"]) if m["kind"] == "svk" else orc.mooney_rivlin(m["mu10"], m["mu01"], m["kappa"], rho0=m["rho0"])
    o = orc.T10Oracle(w["X"], w["conn"], mat, fixed=w["fixed"], f_ext=w["f_ext"])
    o.calc_dndu_pre(); o.calc_mass()
    o.x, o.y, o.z = (np.ascontiguousarray(w["x0"][:, i]) for i in range(3))
    h, rho = w["params"][6], w["params"][3]
    ro, ci, val = o.assemble_hessian(h, rho, nthreads=8)
    n = 3 * o.N
    H = sp.csr_matrix((val, ci, ro), shape=(n, n))
    f_int = o.internal_force(o.v)
    g = o.grad_L(f_int, h, rho)
    return w, H, -g

def pcg(H, b, Minv, tol=1e-12, maxit=5000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; bb = np.sqrt(b @ b)
    for it in range(1, maxit + 1):
        q = H @ p; a = rz / (p @ q); x += a * p; r -= a * q
        if np.sqrt(r @ r) <= tol * bb: return x, it
        z = Minv(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return x, maxit

def block_jacobi(H):
    n = H.shape[0]; N = n // 3
    Hb = H.tobsr(blocksize=(3, 3))
    D = np.zeros((N, 3, 3))
    for i in range(N):
        s, e = Hb.indptr[i], Hb.indptr[i + 1]
        k = s + np.searchsorted(Hb.indices[s:e], i)
        D[i] = Hb.data[k]
    Dinv = np.linalg.inv(D)
    return lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(N, 3)).reshape(-1), Dinv

def cheb_smoother(H, Dinv_apply, lam_max, degree, lam_min_frac=0.1):
    # Chebyshev polynomial smoother on D^-1 H targeting [lam_min_frac*lam_max, lam_max]
    lmax, lmin = lam_max, lam_min_frac * lam_max
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    def smooth(b, x0=None):
        x = np.zeros_like(b) if x0 is None else x0.copy()
        r = b - H @ x if x0 is not None else b.copy()
        sigma = theta / delta; rho = 1.0 / sigma
        d = Dinv_apply(r) / theta
        for k in range(degree):
            x += d
            r -= H @ d
            rho_new = 1.0 / (2 * sigma - rho)
            d = rho_new * rho * d + (2 * rho_new / delta) * Dinv_apply(r)
            rho = rho_new
        return x
    return smooth

def est_lmax(H, Dinv_apply, iters=20):
    v = np.random.default_rng(0).normal(size=H.shape[0])
    for _ in range(iters):
        v = Dinv_apply(H @ v); lam = np.linalg.norm(v); v /= lam
    return lam

def aggregate_by_bins(X, cell):
    """aggregates = nodes binned in a regular grid of size `cell` (per axis)"""
    ijk = np.floor((X - X.min(axis=0)) / cell + 1e-9).astype(np.int64)
    dims = ijk.max(axis=0) + 1
    key = (ijk[:, 2] * dims[1] + ijk[:, 1]) * dims[0] + ijk[:, 0]
    uniq, agg = np.unique(key, return_inverse=True)
    Xc = np.zeros((len(uniq), 3)); cnt = np.bincount(agg)
    for d in range(3): Xc[:, d] = np.bincount(agg, X[:, d]) / cnt
    return agg, Xc

def prolong_translations(agg, nc):
    N = len(agg)
    rows = np.arange(3 * N); cols = 3 * np.repeat(agg, 3) + np.tile(np.arange(3), N)
    return sp.csr_matrix((np.ones(3 * N), (rows, cols)), shape=(3 * N, 3 * nc))

def amg_hierarchy(H, X, cells, smooth_P=False, omega=0.6):
    levels = []
    A, Xl = H, X
    for cell in cells:
        agg, Xc = aggregate_by_bins(Xl, cell)
        P = prolong_translations(agg, len(Xc))
        if smooth_P:
            Dinv_apply, Dinv = block_jacobi(A)
            n = A.shape[0]; N = n // 3
            Dm = sp.bsr_matrix((Dinv, np.arange(N), np.arange(N + 1)), shape=(n, n)).tocsr()
            lam = est_lmax(A, Dinv_apply)
            P = P - (omega * 4.0 / (3.0 * lam)) * (Dm @ (A @ P))
        Ac = (P.T @ A @ P).tocsr()
        levels.append((A, P))
        A, Xl = Ac, Xc
    return levels, A

def make_vcycle(levels, Acoarse, degree=2, coarse="direct"):
    sm = []
    for A, P in levels:
        Dinv_apply, _ = block_jacobi(A)
        lam = est_lmax(A, Dinv_apply) * 1.1
        sm.append(cheb_smoother(A, Dinv_apply, lam, degree, 0.25))
    lu = spla.splu(Acoarse.tocsc())
    def vc(l, b):
        if l == len(levels): return lu.solve(b)
        A, P = levels[l]
        x = sm[l](b)
        rc = P.T @ (b - A @ x)
        x = x + P @ vc(l + 1, rc)
        x = sm[l](b, x)
        return x
    return lambda r: vc(0, r)

if __name__ == "__main__":
    config = sys.argv[1] if len(sys.argv) > 1 else "B"
    cells = tuple(int(c) for c in sys.argv[2].split(",")) if len(sys.argv) > 2 else None
    t0 = time.time(); w, H, b = build(config, cells); print("built", H.shape, H.nnz, f"{time.time()-t0:.1f}s")
    X = w["X"]
    Minv, _ = block_jacobi(H)
    t0 = time.time(); _, it = pcg(H, b, Minv); print("block-Jacobi PCG iterations:", it, f"{time.time()-t0:.1f}s")
    hx = np.diff(np.unique(np.round(X[:, 0], 9)))[0]  # lattice spacing
    print("lattice spacing", hx)
    for name, cs, smP, deg in [("agg 2h,4h,8h cheb2", [2, 4, 8], False, 2), ("agg 2h,4h,8h cheb3", [2, 4, 8], False, 3),
                               ("agg 3h,9h cheb2", [3, 9], False, 2), ("SA 2h,4h,8h cheb2", [2, 4, 8], True, 2),
                               ("SA 3h,9h cheb2", [3, 9], True, 2), ("SA 3h,9h cheb3", [3, 9], True, 3)]:
        t0 = time.time()
        levels, Ac = amg_hierarchy(H, X, [c * hx * 1.0001 for c in cs], smooth_P=smP)
        M = make_vcycle(levels, Ac, degree=deg)
        _, it = pcg(H, b, M)
        sizes = [l[0].shape[0] // 3 for l in levels] + [Ac.shape[0] // 3]
        nnzs = [l[0].nnz for l in levels] + [Ac.nnz]
        # work per V-cycle in fine-SpMV units: (2*deg + 1) SpMV per level
        work = sum((2 * deg + 1) * nz for nz in nnzs[:-1]) / nnzs[0]
        print(f"{name:22s} iters {it:4d}  levels {sizes}  nnz {nnzs}  SpMV-equiv/iter {work+1:.1f}  total {it*(work+1):.0f}  ({time.time()-t0:.1f}s)")


# ---- rigid-body-mode (6 dof / aggregate) hierarchy ---------------------------------------------------
def skew(r):
    return np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])

def prolong_rbm(agg, X, Xc, fine_dofs):
    """fine_dofs = 3 (displacements) or 6 (t, w); coarse always 6.  u_i = t_I + w_I x (x_i - c_I)."""
    N, nc = len(agg), len(Xc)
    rows, cols, vals = [], [], []
    for i in range(N):
        I = agg[i]; r = X[i] - Xc[I]
        S = -skew(r)  # w x r = -[r]x w
        blk = np.zeros((fine_dofs, 6))
        blk[:3, :3] = np.eye(3); blk[:3, 3:] = S
        if fine_dofs == 6: blk[3:, 3:] = np.eye(3)
        for a in range(fine_dofs):
            for b_ in range(6):
                if blk[a, b_] != 0.0:
                    rows.append(fine_dofs * i + a); cols.append(6 * I + b_); vals.append(blk[a, b_])
    return sp.csr_matrix((vals, (rows, cols)), shape=(fine_dofs * N, 6 * nc))

def block_jacobi_bs(A, bs):
    n = A.shape[0]; N = n // bs
    Ab = A.tobsr(blocksize=(bs, bs))
    D = np.zeros((N, bs, bs))
    for i in range(N):
        s, e = Ab.indptr[i], Ab.indptr[i + 1]
        k = s + np.searchsorted(Ab.indices[s:e], i)
        D[i] = Ab.data[k]
    Dinv = np.linalg.inv(D)
    return (lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(N, bs)).reshape(-1)), Dinv

def rbm_hierarchy(H, X, cells, smooth_P=False, omega=0.6):
    levels, A, Xl, fd = [], H, X, 3
    for cell in cells:
        agg, Xc = aggregate_by_bins(Xl, cell)
        P = prolong_rbm(agg, Xl, Xc, fd)
        if smooth_P:
            Dapply, Dinv = block_jacobi_bs(A, fd)
            n = A.shape[0]; N = n // fd
            Dm = sp.bsr_matrix((Dinv, np.arange(N), np.arange(N + 1)), shape=(n, n)).tocsr()
            lam = est_lmax(A, Dapply)
            P = (P - (omega * 4.0 / (3.0 * lam)) * (Dm @ (A @ P))).tocsr()
        Ac = (P.T @ A @ P).tocsr()
        levels.append((A, P, fd))
        A, Xl, fd = Ac, Xc, 6
    return levels, A

def make_vcycle_bs(levels, Acoarse, degree=2, lmin_frac=0.25):
    sm = []
    for A, P, bs in levels:
        Dapply, _ = block_jacobi_bs(A, bs)
        lam = est_lmax(A, Dapply) * 1.1
        sm.append(cheb_smoother(A, Dapply, lam, degree, lmin_frac))
    lu = spla.splu((Acoarse + 1e-12 * sp.identity(Acoarse.shape[0]) * abs(Acoarse).max()).tocsc())
    def vc(l, b):
        if l == len(levels): return lu.solve(b)
        A, P, _ = levels[l]
        x = sm[l](b)
        x = x + P @ vc(l + 1, P.T @ (b - A @ x))
        return sm[l](b, x)
    return lambda r: vc(0, r)

def run_rbm(config="B", cells=None):
    w, H, b = build(config, cells)
    X = w["X"]; hx = np.diff(np.unique(np.round(X[:, 0], 9)))[0]
    Minv, _ = block_jacobi(H)
    _, it0 = pcg(H, b, Minv); print("block-Jacobi:", it0)
    for name, cs, smP, deg, lf in [("RBM 2h,4h,8h cheb2", [2, 4, 8], False, 2, 0.25), ("RBM 3h,9h cheb2", [3, 9], False, 2, 0.25),
                                   ("RBM 2h,4h,8h cheb3 .1", [2, 4, 8], False, 3, 0.1), ("RBM-SA 2h,4h,8h cheb2", [2, 4, 8], True, 2, 0.25),
                                   ("RBM-SA 3h,9h cheb2", [3, 9], True, 2, 0.25), ("RBM 2h,4h,8h cheb1", [2, 4, 8], False, 1, 0.3)]:
        t0 = time.time()
        levels, Ac = rbm_hierarchy(H, X, [c * hx * 1.0001 for c in cs], smooth_P=smP)
        M = make_vcycle_bs(levels, Ac, degree=deg, lmin_frac=lf)
        _, it = pcg(H, b, M)
        nnzs = [l[0].nnz for l in levels] + [Ac.nnz]
        work = sum((2 * deg + 1) * nz for nz in nnzs[:-1]) / nnzs[0]
        print(f"{name:24s} iters {it:4d} nnz {nnzs} SpMV-equiv/iter {work+1:.1f} total {it*(work+1):.0f} ({time.time()-t0:.1f}s)")


def p1_prolongation(conn, N):
    """P1 (vertex) -> P2 (all nodes): vertex nodes inject, mid-edge nodes average their two end vertices."""
    verts = np.unique(conn[:, :4]); vid = -np.ones(N, dtype=np.int64); vid[verts] = np.arange(len(verts))
    rows, cols, vals = list(verts), list(vid[verts]), [1.0] * len(verts)
    done = np.zeros(N, bool)
    for k, (a, b_) in enumerate(tl_mesh.EDGES):
        mid = conn[:, 4 + k]; va = conn[:, a]; vb = conn[:, b_]
        _, first = np.unique(mid, return_index=True)
        for idx in first:
            m_ = mid[idx]
            if done[m_]: continue
            done[m_] = True
            rows += [m_, m_]; cols += [vid[va[idx]], vid[vb[idx]]]; vals += [0.5, 0.5]
    Pn = sp.csr_matrix((vals, (rows, cols)), shape=(N, len(verts)))
    return sp.kron(Pn, sp.identity(3)).tocsr(), verts

def run_pmg(config="B", cells=None):
    w, H, b = build(config, cells)
    X, conn = w["X"], w["conn"]
    Minv, _ = block_jacobi(H)
    _, it0 = pcg(H, b, Minv); print("block-Jacobi:", it0)
    P, verts = p1_prolongation(conn, X.shape[0])
    Ac = (P.T @ H @ P).tocsr()
    print("coarse P1:", Ac.shape, Ac.nnz)
    lu = spla.splu(Ac.tocsc())
    lam = est_lmax(H, Minv) * 1.1
    for deg, lf in [(1, 0.3), (2, 0.25), (3, 0.15), (4, 0.1)]:
        sm = cheb_smoother(H, Minv, lam, deg, lf)
        def M(r):
            x = sm(r)
            x = x + P @ lu.solve(P.T @ (r - H @ x))
            return sm(r, x)
        _, it = pcg(H, b, M)
        print(f"p-MG exact coarse, cheb{deg}: iters {it}  fine SpMV/iter {2*deg+2}  total fine SpMV {it*(2*deg+2)}")
    # additive: Jacobi + coarse
    def Madd(r): return Minv(r) + P @ lu.solve(P.T @ r)
    _, it = pcg(H, b, Madd); print("additive Jacobi + exact P1 coarse: iters", it)
    # what does the coarse P1 problem itself need? Jacobi-PCG iterations on Ac
    Mc, _ = block_jacobi(Ac)
    bc = P.T @ b
    _, itc = pcg(Ac, bc, Mc); print("P1 coarse problem alone, block-Jacobi PCG iterations:", itc)


def run_patch(config="B", cells=None):
    """additive Schwarz over vertex patches (vertex + the mid-edge nodes of its incident edges)"""
    w, H, b = build(config, cells)
    X, conn = w["X"], w["conn"]; N = X.shape[0]
    Minv, _ = block_jacobi(H)
    _, it0 = pcg(H, b, Minv); print("block-Jacobi:", it0)
    verts = np.unique(conn[:, :4])
    patch = {v: {v} for v in verts}
    for k, (a, b_) in enumerate(tl_mesh.EDGES):
        for e in range(conn.shape[0]):
            m_ = conn[e, 4 + k]; patch[conn[e, a]].add(m_); patch[conn[e, b_]].add(m_)
    Hc = H.tocsc()
    invs = []
    cover = np.zeros(N)
    for v in verts:
        nodes = np.array(sorted(patch[v])); cover[nodes] += 1
        dofs = (3 * nodes[:, None] + np.arange(3)[None, :]).reshape(-1)
        A = H[dofs][:, dofs].toarray()
        invs.append((dofs, np.linalg.inv(A)))
    print("patches", len(invs), "avg size", np.mean([len(d) for d, _ in invs]), "cover min/max", cover.min(), cover.max())
    wgt = np.repeat(1.0 / np.sqrt(cover), 3)
    def M_as(r):
        z = np.zeros_like(r)
        for dofs, Ai in invs: z[dofs] += Ai @ r[dofs]
        return z
    def M_ras(r):  # symmetric weighted (partition of unity split as sqrt)
        z = np.zeros_like(r); rw = wgt * r
        for dofs, Ai in invs: z[dofs] += Ai @ rw[dofs]
        return wgt * z
    _, it = pcg(H, b, M_as); print("additive Schwarz vertex patches: iters", it)
    _, it = pcg(H, b, M_ras); print("weighted additive Schwarz: iters", it)
    # two-level: patches + exact P1 coarse
    P, _ = p1_prolongation(conn, N); Ac = (P.T @ H @ P).tocsc(); lu = spla.splu(Ac)
    def M2(r): return M_as(r) + P @ lu.solve(P.T @ r)
    _, it = pcg(H, b, M2); print("additive Schwarz + exact P1 coarse: iters", it)


def run_chebpoly(config="B", cells=None):
    """PCG with a degree-d Chebyshev polynomial of D^-1 H as preconditioner (no inner dot products)."""
    w, H, b = build(config, cells)
    Minv, _ = block_jacobi(H)
    _, it0 = pcg(H, b, Minv); print("block-Jacobi PCG:", it0)
    lmax = est_lmax(H, Minv, 30) * 1.05
    # lambda_min estimate from Lanczos on the first CG iterations is emulated by trying several kappa guesses
    for kappa in (100, 400, 1000, 2500):
        for deg in (4, 8, 16):
            a, bb = lmax / kappa, lmax
            theta, delta = 0.5 * (bb + a), 0.5 * (bb - a)
            sigma = theta / delta
            def M(r, deg=deg, theta=theta, delta=delta, sigma=sigma):
                rho = 1.0 / sigma
                d = Minv(r) / theta
                z = d.copy()
                res = r.copy()
                for _ in range(deg - 1):
                    res -= H @ d
                    rho_new = 1.0 / (2 * sigma - rho)
                    d = rho_new * rho * d + (2 * rho_new / delta) * Minv(res)
                    z += d
                    rho = rho_new
                return z
            _, it = pcg(H, b, M)
            print(f"kappa {kappa:5d} deg {deg:2d}: outer iters {it:4d}  SpMVs {it*deg:5d}  kernels ~{it*(deg-1+2)}")
