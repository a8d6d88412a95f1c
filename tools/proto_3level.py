"""CPU prototype (SciPy): does a THIRD level with an exact solve on rigid-body modes of LARGE aggregates let the vertex-level
polynomial of the p-multigrid cycle be short?   python tools/proto_3level.py nx,ny,nz
Counts PCG iterations to 1e-12 for the device's two-level cycle (Chebyshev(kc) on the vertex level) and for the three-level
variant (k-term Chebyshev before and after an exact rigid-body-mode correction)."""
import importlib
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, '.')
from oracle import orc  # noqa: E402

wl = importlib.import_module("total-lagrangian-fea_amd.workloads")
EDGES = [(0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3)]


def build(cells):
    w = wl.build("C", cells=cells)
    m = w["material"]
    o = orc.T10Oracle(w["X"], w["conn"], orc.svk(m["E"], m["nu"], rho0=m["rho0"]), fixed=w["fixed"], f_ext=w["f_ext"])
    o.calc_dndu_pre()
    o.calc_mass()
    o.x, o.y, o.z = (np.ascontiguousarray(w["x0"][:, i]) for i in range(3))
    h, rho = w["params"][6], w["params"][3]
    ro, ci, val = o.assemble_hessian(h, rho, nthreads=8)
    n = 3 * o.N
    H = sp.csr_matrix((val, ci, ro), shape=(n, n))
    g = o.grad_L(o.internal_force(o.v), h, rho)
    return w, H, -g


def p_prolongation(N, conn):
    is_v = np.zeros(N, bool)
    is_v[conn[:, :4]] = True
    cid = -np.ones(N, int)
    cid[is_v] = np.arange(is_v.sum())
    rows, cols, vals = [], [], []
    v = np.where(is_v)[0]
    rows += list(v); cols += list(cid[v]); vals += [1.0] * len(v)
    par = {}
    for m, (a, b) in enumerate(EDGES):
        for e in range(conn.shape[0]):
            par[conn[e, 4 + m]] = (cid[conn[e, a]], cid[conn[e, b]])
    for n, (a, b) in par.items():
        rows += [n, n]; cols += [a, b]; vals += [0.5, 0.5]
    P = sp.csr_matrix((vals, (rows, cols)), shape=(N, is_v.sum()))
    return sp.kron(P, sp.identity(3), format="csr"), v


def block_dinv(A):
    N = A.shape[0] // 3
    Ab = A.tobsr(blocksize=(3, 3))
    D = np.zeros((N, 3, 3))
    for i in range(N):
        s, e = Ab.indptr[i], Ab.indptr[i + 1]
        D[i] = Ab.data[s + np.searchsorted(Ab.indices[s:e], i)]
    Dinv = np.linalg.inv(D)
    return lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(N, 3)).reshape(-1)


def lam_max(A, Dinv, iters=30):
    v = np.random.default_rng(0).normal(size=A.shape[0])
    lam = 1.0
    for _ in range(iters):
        v = Dinv(A @ v)
        lam = np.linalg.norm(v)
        v /= lam
    return 1.15 * lam


def cheb(A, Dinv, lmax, kappa, deg):
    """x = p_deg(D^-1 A) D^-1 b on [lmax/kappa, lmax]: deg terms (deg - 1 products)"""
    lmin = lmax / kappa
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta / delta

    def apply(b):
        r = b.copy()
        d = Dinv(r) / theta
        x = d.copy()
        rho = 1.0 / sigma
        for _ in range(deg - 1):
            r -= A @ d
            rho_new = 1.0 / (2 * sigma - rho)
            d = rho_new * rho * d + (2 * rho_new / delta) * Dinv(r)
            x += d
            rho = rho_new
        return x
    return apply


def rbm_prolongation(Xv, cell):
    ijk = np.floor((Xv - Xv.min(axis=0)) / cell + 1e-9).astype(int)
    dims = ijk.max(axis=0) + 1
    key = (ijk[:, 2] * dims[1] + ijk[:, 1]) * dims[0] + ijk[:, 0]
    uniq, agg = np.unique(key, return_inverse=True)
    na = len(uniq)
    cen = np.stack([np.bincount(agg, Xv[:, d]) / np.bincount(agg) for d in range(3)], axis=1)
    rows, cols, vals = [], [], []
    for i in range(len(Xv)):
        r = Xv[i] - cen[agg[i]]
        a = 6 * agg[i]
        for d in range(3):
            rows.append(3 * i + d); cols.append(a + d); vals.append(1.0)
        # u = w x r
        S = np.array([[0, r[2], -r[1]], [-r[2], 0, r[0]], [r[1], -r[0], 0]])
        for d in range(3):
            for e in range(3):
                if S[d, e] != 0.0:
                    rows.append(3 * i + d); cols.append(a + 3 + e); vals.append(S[d, e])
    return sp.csr_matrix((vals, (rows, cols)), shape=(3 * len(Xv), 6 * na)), na


def pcg(H, b, M, tol=1e-12, maxit=400):
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z; bb = np.sqrt(b @ b)
    for it in range(1, maxit + 1):
        q = H @ p; a = rz / (p @ q); x += a * p; r -= a * q
        if np.sqrt(r @ r) <= tol * bb:
            return it
        z = M(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return maxit


if __name__ == "__main__":
    cells = tuple(int(c) for c in sys.argv[1].split(",")) if len(sys.argv) > 1 else (30, 20, 10)
    t0 = time.time()
    w, H, b = build(cells)
    P, vnodes = p_prolongation(w["X"].shape[0], w["conn"])
    Hc = (P.T @ H @ P).tocsr()
    Xv = w["X"][vnodes]
    print(f"cells {cells}: fine {H.shape[0]} DOF, vertex level {Hc.shape[0]} DOF, {time.time() - t0:.1f} s", flush=True)
    Df, Dc = block_dinv(H), block_dinv(Hc)
    lf, lc = lam_max(H, Df), lam_max(Hc, Dc)
    sm = cheb(H, Df, lf, 8.0, 2)

    def two_grid(coarse):
        def M(r):
            x = sm(r)
            x = x + P @ coarse(P.T @ (r - H @ x))
            return x + sm(r - H @ x)
        return M

    for kc in ((24, 32, 48) if len(sys.argv) > 2 else (8, 12, 16, 24, 32, 48)):
        it = pcg(H, b, two_grid(cheb(Hc, Dc, lc, 1.5 * kc * kc, kc)))
        print(f"two-level, vertex polynomial degree {kc}: {it} CG iterations, {kc - 1} vertex-level products per cycle", flush=True)
    hcell = 3.0 / 90.0
    for ncell in ((5, 8, 10) if len(sys.argv) > 2 else (3, 5, 8)):
        P2, na = rbm_prolongation(Xv, ncell * hcell)
        H3 = (P2.T @ Hc @ P2).toarray()
        # degenerate aggregates (all members collinear / coplanar: a rotation mode that moves nothing) get a unit diagonal
        dg = np.abs(np.diag(H3))
        dead = dg < 1e-12 * dg.max()
        H3[dead, :] = 0.0
        H3[:, dead] = 0.0
        H3[dead, dead] = 1.0
        H3i = np.linalg.inv(H3)
        H3i[dead, :] = 0.0
        for k, kap in ((2, 8), (3, 16), (4, 30), (6, 60), (8, 100)):
            smc = cheb(Hc, Dc, lc, kap, k)

            def coarse(rc):
                y = smc(rc)
                y = y + P2 @ (H3i @ (P2.T @ (rc - Hc @ y)))
                return y + smc(rc - Hc @ y)
            it = pcg(H, b, two_grid(coarse))
            print(f"three-level, {na} aggregates of {ncell}^3 cells ({6 * na} exact unknowns), {k}-term smoother kappa {kap}: "
                  f"{it} CG iterations, {2 * k} vertex-level products per cycle", flush=True)
