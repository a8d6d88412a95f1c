"""Host plan of the engine's own sparse direct solve (csrc/mf_host.h): nested-dissection fronts, row structures, extend-add
maps, H -> front entry lists and level order, checked end to end by EXECUTING the plan with plain loops on the CPU
(tests/native/mf_shim.cc) against scipy's sparse solve.  Integer work + a dense reference executor -- runs without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from tests.helpers import load_mesh
from tests.test_rowgroups import adjacency, dp, ip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mu = __import__("importlib").import_module("total-lagrangian-fea_amd.mesh_utils")


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = tmp_path_factory.mktemp("mf") / "libmf_shim.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(so),
                           os.path.join(ROOT, "tests", "native", "mf_shim.cc")])
    lib = C.CDLL(str(so))
    lib.mf_build.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                             C.POINTER(C.c_double), C.c_int, C.c_longlong, C.POINTER(C.c_longlong)]
    return lib


def spd_in_engine_layout(N, off, cols, seed):
    """random SPD matrix on the node pattern, in the engine's value layout (node row i: 9 off[i] + d * 3 deg + 3 k + e)"""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(N), np.diff(off))
    blk = rng.normal(size=(cols.size, 3, 3))
    key = {(int(r), int(c)): t for t, (r, c) in enumerate(zip(rows, cols))}
    for t, (r, c) in enumerate(zip(rows, cols)):
        if c < r:
            blk[t] = blk[key[(int(c), int(r))]].T
        elif c == r:
            blk[t] = 0.5 * (blk[t] + blk[t].T) + 9.0 * (off[r + 1] - off[r]) * np.eye(3)
    vals = np.zeros(9 * cols.size)
    ri, ci, vv = [], [], []
    for t, (r, c) in enumerate(zip(rows, cols)):
        deg = off[r + 1] - off[r]
        k = t - off[r]
        for d in range(3):
            base = 9 * off[r] + d * 3 * deg + 3 * k
            vals[base:base + 3] = blk[t][d]
            for e in range(3):
                ri.append(3 * r + d)
                ci.append(3 * c + e)
                vv.append(blk[t][d, e])
    return vals, sp.csc_matrix((vv, (ri, ci)), shape=(3 * N, 3 * N))


@pytest.mark.parametrize("mesh,leaf", [("box", 8), ("box", 32), ("res2", 16), ("bunny", 32), ("beam_3x2x1", 4)])
def test_plan_factors_and_solves(shim, mesh, leaf):
    if mesh == "box":
        X, conn = mu.structured_t10_box(4, 3, 3, 2.0, 1.5, 1.5)
    else:
        X, conn = load_mesh(mesh)
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    N = X.shape[0]
    off, cols, _, _ = adjacency(conn, N)
    x, y, z = (np.ascontiguousarray(X[:, k]) for k in range(3))
    info = (C.c_longlong * 8)()
    assert shim.mf_build(N, ip(off), ip(cols), dp(x), dp(y), dp(z), leaf, 1 << 40, info) == 0
    nf, nl = int(info[0]), int(info[1])
    order = np.zeros(N, dtype=np.int32)
    c0, c1, par, dep, nr = (np.zeros(nf, dtype=np.int32) for _ in range(5))
    shim.mf_fetch(ip(order), ip(c0), ip(c1), ip(par), ip(dep), ip(nr))
    assert np.array_equal(np.sort(order), np.arange(N))                       # a permutation
    own = np.zeros(N, dtype=np.int32)
    for f in range(nf):
        own[c0[f]:c1[f]] += 1
        assert par[f] == -1 or (par[f] > f and dep[par[f]] == dep[f] - 1 and c0[par[f]] >= c1[f])
        assert nr[f] >= c1[f] - c0[f]
    assert np.all(own == 1) and (par == -1).sum() == 1 and dep.max() + 1 == nl   # every node owned once, one root
    assert int(info[7]) == (cols.size + N) // 2                                 # one entry per lower node block
    vals, A = spd_in_engine_layout(N, off, cols, 5)
    b = np.random.default_rng(7).normal(size=3 * N)
    xo = np.zeros(3 * N)
    assert shim.mf_cpu_factor_solve(dp(vals), dp(b), dp(xo)) == 0
    ref = spla.spsolve(A, b)
    assert np.linalg.norm(xo - ref) <= 1e-12 * np.linalg.norm(ref)
    assert np.linalg.norm(A @ xo - b) <= 1e-13 * np.linalg.norm(b)
    # a memory bound that the whole-level workspaces exceed: the tree is cut into subtrees that reuse their workspaces
    # (what lets config C's 118 GB factor fit next to its fronts); same ordering, same factor, same solution
    L, ws = int(info[2]), int(info[3]) + int(info[4])
    info2 = (C.c_longlong * 8)()
    for frac in (0.75, 0.5, 0.35):
        if shim.mf_build(N, ip(off), ip(cols), dp(x), dp(y), dp(z), leaf, L + int(frac * ws), info2) != 0:
            continue
        assert int(info2[2]) == L and int(info2[3]) + int(info2[4]) <= frac * ws and int(info2[4]) > 0
        x2 = np.zeros(3 * N)
        assert shim.mf_cpu_factor_solve(dp(vals), dp(b), dp(x2)) == 0
        assert np.linalg.norm(x2 - xo) <= 1e-13 * np.linalg.norm(xo)


def test_plan_refuses_what_does_not_fit(shim):
    X, conn = mu.structured_t10_box(4, 3, 3, 2.0, 1.5, 1.5)
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    N = X.shape[0]
    off, cols, _, _ = adjacency(conn, N)
    x, y, z = (np.ascontiguousarray(X[:, k]) for k in range(3))
    info = (C.c_longlong * 8)()
    assert shim.mf_build(N, ip(off), ip(cols), dp(x), dp(y), dp(z), 16, 1000, info) == 1


def _solve_on_graph(shim, N, pairs, leaf, seed=1):
    """SPD matrix on an arbitrary node graph (pairs of adjacent nodes), planned, factored and solved on the CPU"""
    adj = [set([i]) for i in range(N)]
    for a, b in pairs:
        adj[a].add(b)
        adj[b].add(a)
    off = np.zeros(N + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(a) for a in adj])
    cols = np.concatenate([np.array(sorted(a), dtype=np.int32) for a in adj])
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(N, 3))
    x, y, z = (np.ascontiguousarray(X[:, k]) for k in range(3))
    info = (C.c_longlong * 8)()
    assert shim.mf_build(N, ip(off), ip(cols), dp(x), dp(y), dp(z), leaf, 1 << 40, info) == 0
    vals, A = spd_in_engine_layout(N, off, cols, seed)
    b = rng.normal(size=3 * N)
    xo = np.zeros(3 * N)
    assert shim.mf_cpu_factor_solve(dp(vals), dp(b), dp(xo)) == 0
    assert np.linalg.norm(A @ xo - b) <= 1e-12 * np.linalg.norm(b)
    return int(info[0]), int(info[1])


def test_plan_edge_cases(shim):
    """One node; a mesh smaller than a leaf (one front); a chain (every separator is one node); two disconnected components
    (an empty separator at the root: a front without own columns); a complete graph (one dense front after all)."""
    assert _solve_on_graph(shim, 1, [], 4) == (1, 1)
    assert _solve_on_graph(shim, 7, [(i, i + 1) for i in range(6)], 32) == (1, 1)
    nf, nl = _solve_on_graph(shim, 200, [(i, i + 1) for i in range(199)], 4)
    assert nf > 30 and nl > 4
    comp = [(i, i + 1) for i in range(39)] + [(40 + i, 41 + i) for i in range(39)]       # nodes 0..39 and 40..79
    assert _solve_on_graph(shim, 80, comp, 8)[0] > 5
    assert _solve_on_graph(shim, 24, [(i, j) for i in range(24) for j in range(i)], 4)[0] >= 1
