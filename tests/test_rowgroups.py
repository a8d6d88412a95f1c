"""Host set-up of the fused tangent + assembly kernel (csrc/rowgroup_host.h): every row owned once, every (row, element)
instance listed once in ascending element order, packed accumulator offsets consistent with the sorted column lists,
accumulators of one group disjoint and within the 16-bit packing; every block's mass carried by exactly one item; the
pass table walks every instance once, group by group.  Integer work only -- runs
without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.helpers import load_mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mu = __import__("importlib").import_module("total-lagrangian-fea_amd.mesh_utils")


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = tmp_path_factory.mktemp("rg") / "librowgroup_shim.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-o", str(so),
                           os.path.join(ROOT, "tests", "native", "rowgroup_shim.cc")])
    return C.CDLL(str(so))


def adjacency(conn, N):
    E, S = conn.shape
    order = np.argsort(conn.reshape(-1), kind="stable")          # ascending element per node (codes are e*S + a)
    n2e = order.astype(np.int32)
    n2e_off = np.zeros(N + 1, dtype=np.int32)
    np.cumsum(np.bincount(conn.reshape(-1), minlength=N), out=n2e_off[1:])
    rows = np.repeat(conn, S, axis=1).reshape(-1)
    colsv = np.tile(conn, (1, S)).reshape(-1)
    pairs = np.unique(rows.astype(np.int64) * N + colsv)
    r, c = pairs // N, pairs % N
    off = np.zeros(N + 1, dtype=np.int32)
    np.cumsum(np.bincount(r, minlength=N), out=off[1:])
    return off, c.astype(np.int32), n2e_off, n2e


def ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("mesh", ["res2", "bunny", "box"])
def test_row_groups_invariants(shim, mesh):
    if mesh == "box":
        X, conn = mu.structured_t10_box(5, 4, 3, 2.5, 2.0, 1.5)
    else:
        X, conn = load_mesh(mesh)
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    N, (E, S) = X.shape[0], conn.shape
    off, cols, n2e_off, n2e = adjacency(conn, N)
    conn_cm = np.ascontiguousarray(conn.T)                       # column-major E x S, the engine's layout
    x, y, z = (np.ascontiguousarray(X[:, k]) for k in range(3))
    sizes = np.zeros(5, dtype=np.int32)
    assert shim.rg_build(N, E, S, ip(conn_cm), ip(off), ip(cols), ip(n2e_off), ip(n2e), dp(x), dp(y), dp(z), ip(sizes)) == 0
    G, n_inst, acc_max, P, G2 = map(int, sizes)
    assert G2 == G
    assert n_inst == E * S
    g_inst_off, g_row_off = np.zeros(G + 1, np.int32), np.zeros(G + 1, np.int32)
    gr_row, gr_acc = np.zeros(N, np.int32), np.zeros(N, np.int32)
    gi_code, gi_pack = np.zeros(n_inst, np.int32), np.zeros(n_inst * S, np.int32)
    shim.rg_fetch(ip(g_inst_off), ip(g_row_off), ip(gr_row), ip(gr_acc), ip(gi_code), ip(gi_pack))
    assert g_row_off[0] == 0 and g_row_off[-1] == N and g_inst_off[0] == 0 and g_inst_off[-1] == n_inst
    assert np.array_equal(np.sort(gr_row), np.arange(N))         # every row has exactly one owner
    assert np.array_equal(np.sort(gi_code), np.arange(E * S))    # every (element, local node) listed once
    deg = np.diff(off)
    gi_pack = gi_pack.reshape(n_inst, S)
    pt, g_pass_off = np.zeros(4 * P, np.int32), np.zeros(G + 1, np.int32)
    gr_info, gi_mb = np.zeros(4 * N, np.int32), np.zeros(n_inst, np.int32)
    shim.rg_fetch_passes(ip(pt), ip(g_pass_off), ip(gr_info), ip(gi_mb))
    pt, gr_info = pt.reshape(P, 4), gr_info.reshape(N, 4)
    mass_flag = gi_pack < 0                                      # bit 31: the item carries its block's M/h
    gi_pack = gi_pack & 0x7FFFFFFF
    carried = np.zeros(len(cols), dtype=np.int32)                # how many items carry the mass of each block
    pi = 0                                                       # pass cursor
    for g in range(G):
        rows = gr_row[g_row_off[g]:g_row_off[g + 1]]
        acc = gr_acc[g_row_off[g]:g_row_off[g + 1]]
        assert acc[0] == 0 and np.array_equal(np.diff(acc), 9 * deg[rows[:-1]])   # packed back to back
        assert acc[-1] + 9 * deg[rows[-1]] <= acc_max < 65536
        w = g_inst_off[g]
        for i, a0 in zip(rows, acc):
            codes = n2e[n2e_off[i]:n2e_off[i + 1]]
            n = len(codes)
            assert np.array_equal(gi_code[w:w + n], codes)      # ascending element: the fixed summation order
            c = cols[off[i]:off[i + 1]]
            for k in range(n):
                e = codes[k] // S
                assert conn[e, codes[k] % S] == i
                pos = np.searchsorted(c, conn[e])
                assert np.array_equal(c[pos], conn[e])
                assert np.array_equal(gi_pack[w + k] & 0xFFFF, a0 + 3 * pos)
                assert np.all(gi_pack[w + k] >> 16 == 3 * deg[i])
                # mass value of the block: mval[gi_mb + acc offset / 3] == mval[off[i] + pos]
                assert np.array_equal(gi_mb[w + k] + (gi_pack[w + k] & 0xFFFF) // 3, off[i] + pos)
                np.add.at(carried, off[i] + pos[mass_flag[w + k]], 1)
                if k == 0:
                    assert np.all(mass_flag[w + k])             # the lowest element is the first contribution
            w += n
        assert w == g_inst_off[g + 1]
        # row records and passes of the group
        r0, nr = g_row_off[g], g_row_off[g + 1] - g_row_off[g]
        for t, (i, a0) in enumerate(zip(rows, acc)):
            c = cols[off[i]:off[i + 1]]
            assert gr_info[r0 + t].tolist() == [a0 | (int(np.searchsorted(c, i)) << 16), off[i], deg[i], i]
        i0, i1 = g_inst_off[g], g_inst_off[g + 1]
        assert g_pass_off[g] == pi
        p0 = i0
        while True:
            inst0, meta, row0, acc_n = pt[pi]
            cnt = min(6, i1 - p0)
            assert inst0 == p0 and (meta & 7) == cnt and (meta >> 8) == nr and row0 == r0
            assert bool(meta & 8) == (p0 == i0) and bool(meta & 16) == (p0 + 6 >= i1)
            assert acc_n == acc[-1] + 9 * deg[rows[-1]]
            pi += 1
            p0 += 6
            if p0 >= i1:
                break
    assert pi == P
    assert np.all(carried == 1)                                  # every block's M/h enters H exactly once
    assert g_pass_off[0] == 0 and g_pass_off[-1] == P and np.all(np.diff(g_pass_off) > 0)
    # locality: consecutive groups are spatial neighbours (Morton order) -- the median distance between the first rows of
    # consecutive groups is a few element sizes, far below the body's extent
    first = X[gr_row[g_row_off[:-1]]]
    hop = np.linalg.norm(np.diff(first, axis=0), axis=1)
    assert np.median(hop) < 0.35 * np.linalg.norm(X.max(0) - X.min(0))


@pytest.mark.parametrize("mesh", ["res2", "bunny", "box"])
def test_row_groups_affine_form(shim, mesh):
    """Work lists of the affine-element kernel (build_row_groups4): rows and instances owned once, the header and the
    sixteen 16-bit block offsets of every instance -- entry (n, p) is the column of vertex n (p == n) or of the mid-edge
    node of (n, p) --, passes of at most 16 instances that walk every group once."""
    if mesh == "box":
        X, conn = mu.structured_t10_box(5, 4, 3, 2.5, 2.0, 1.5)
    else:
        X, conn = load_mesh(mesh)
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    N, (E, S) = X.shape[0], conn.shape
    off, cols, n2e_off, n2e = adjacency(conn, N)
    conn_cm = np.ascontiguousarray(conn.T)
    x, y, z = (np.ascontiguousarray(X[:, k]) for k in range(3))
    sizes = np.zeros(4, dtype=np.int32)
    assert shim.rg4_build(N, E, ip(conn_cm), ip(off), ip(cols), ip(n2e_off), ip(n2e), dp(x), dp(y), dp(z), ip(sizes)) == 0
    G, n_inst, acc_max, P = map(int, sizes)
    assert n_inst == E * S and acc_max < 65536
    g_inst_off, g_row_off = np.zeros(G + 1, np.int32), np.zeros(G + 1, np.int32)
    gr_row, gr_acc = np.zeros(N, np.int32), np.zeros(N, np.int32)
    gi_head, gi_ent = np.zeros(2 * n_inst, np.int32), np.zeros(8 * n_inst, np.int32)
    pt, g_pass_off, gr_info = np.zeros(4 * P, np.int32), np.zeros(G + 1, np.int32), np.zeros(4 * N, np.int32)
    shim.rg4_fetch(ip(g_inst_off), ip(g_row_off), ip(gr_row), ip(gr_acc), ip(gi_head), ip(gi_ent), ip(pt), ip(g_pass_off),
                   ip(gr_info))
    gi_head, pt, gr_info = gi_head.reshape(n_inst, 2), pt.reshape(P, 4), gr_info.reshape(N, 4)
    words = gi_ent.view(np.uint16).reshape(n_inst, 4, 4)          # little endian: low half first = p 0, 1 | 2, 3
    assert np.array_equal(np.sort(gr_row), np.arange(N))
    assert np.array_equal(np.sort(gi_head[:, 0]), np.arange(E * S))
    mid = np.array([[-1, 4, 6, 7], [4, -1, 5, 8], [6, 5, -1, 9], [7, 8, 9, -1]])   # FEAT10Data.cu:143
    deg = np.diff(off)
    pi = 0
    for g in range(G):
        rows = gr_row[g_row_off[g]:g_row_off[g + 1]]
        acc = gr_acc[g_row_off[g]:g_row_off[g + 1]]
        assert acc[0] == 0 and np.array_equal(np.diff(acc), 9 * deg[rows[:-1]])
        assert acc[-1] + 9 * deg[rows[-1]] <= acc_max
        w = g_inst_off[g]
        for t, (i, a0) in enumerate(zip(rows, acc)):
            c = cols[off[i]:off[i + 1]]
            assert gr_info[g_row_off[g] + t].tolist() == [a0 | (int(np.searchsorted(c, i)) << 16), off[i], deg[i], i]
            codes = n2e[n2e_off[i]:n2e_off[i + 1]]
            assert np.array_equal(gi_head[w:w + len(codes), 0], codes)       # ascending element
            assert np.all(gi_head[w:w + len(codes), 1] == 3 * deg[i])
            for k, code in enumerate(codes):
                e = code // S
                assert conn[e, code % S] == i
                for n in range(4):
                    for p in range(4):
                        j = n if p == n else mid[n, p]
                        pos = int(np.searchsorted(c, conn[e, j]))
                        assert c[pos] == conn[e, j] and words[w + k, n, p] == a0 // 3 + pos
            w += len(codes)
        assert w == g_inst_off[g + 1]
        i0, i1, nr = g_inst_off[g], g_inst_off[g + 1], len(rows)
        assert g_pass_off[g] == pi
        p0 = i0
        while True:
            inst0, meta, row0, acc_n = pt[pi]
            assert inst0 == p0 and (meta & 31) == min(16, i1 - p0) and (meta >> 8) == nr and row0 == g_row_off[g]
            assert bool(meta & 32) == (p0 == i0) and bool(meta & 64) == (p0 + 16 >= i1)
            assert acc_n == acc[-1] + 9 * deg[rows[-1]]
            pi += 1
            p0 += 16
            if p0 >= i1:
                break
    assert pi == P and g_pass_off[-1] == P
    fill = n_inst / (16.0 * P)
    assert fill > 0.6, fill                                        # passes are mostly full (4 lanes per instance)


@pytest.mark.parametrize("mesh", ["beam_3x2x1", "res2"])
def test_direct_solver_symbolic_factor(shim, mesh):
    """Host set-up of the sparse direct solve (csrc/direct_host.h): the permutation is a bijection that keeps the three
    DOFs of a node together, and the pattern it predicts for the Cholesky factor of Q^T H Q is exactly the pattern a dense
    symbolic elimination of the permuted matrix produces (no entry missing: rocSOLVER's re-factorisation computes on
    this pattern only; none superfluous at node level)."""
    X, conn = load_mesh(mesh)
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    N = X.shape[0]
    off, cols, _, _ = adjacency(conn, N)
    x, y, z = (np.ascontiguousarray(X[:, k]) for k in range(3))
    sizes = np.zeros(2, dtype=np.int32)
    shim.dh_build.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                              C.POINTER(C.c_double), C.c_longlong, C.POINTER(C.c_int)]
    assert shim.dh_build(N, ip(off), ip(cols), dp(x), dp(y), dp(z), 10**9, ip(sizes)) == 0
    n, nnzT = map(int, sizes)
    assert n == 3 * N
    perm, ptrT, indT = np.zeros(n, np.int32), np.zeros(n + 1, np.int32), np.zeros(nnzT, np.int32)
    shim.dh_fetch(ip(perm), ip(ptrT), ip(indT))
    assert np.array_equal(np.sort(perm), np.arange(n))
    assert np.array_equal(perm.reshape(N, 3) % 3, np.tile(np.arange(3), (N, 1)))        # blocks stay together, in order
    assert np.all(perm.reshape(N, 3)[:, 0] // 3 == perm.reshape(N, 3)[:, 2] // 3)
    # dense symbolic elimination of the permuted NODE graph
    order = perm.reshape(N, 3)[:, 0] // 3
    inv = np.empty(N, dtype=np.int64)
    inv[order] = np.arange(N)
    A = np.zeros((N, N), dtype=bool)
    for i in range(N):
        A[inv[i], inv[cols[off[i]:off[i + 1]]]] = True
    L = np.tril(A)
    for k in range(N):
        below = np.where(L[k + 1:, k])[0] + k + 1
        L[np.ix_(below, below)] |= np.tril(np.ones((len(below), len(below)), dtype=bool))
    # expected DOF pattern: full 3x3 blocks below the diagonal, lower triangle of the diagonal blocks
    for k in range(N):
        js = np.where(L[k, :k])[0]
        for r in range(3):
            row = indT[ptrT[3 * k + r]:ptrT[3 * k + r + 1]]
            expect = np.concatenate([(3 * js[:, None] + np.arange(3)[None, :]).reshape(-1), 3 * k + np.arange(r + 1)])
            assert np.array_equal(row, expect), (k, r)
    assert ptrT[-1] == nnzT


@pytest.mark.parametrize("S", [10, 8, 16])
def test_tangent_lane_pair_runs(shim, S):
    """tangent_blocks_kernel's lane -> node-pair assignment (csrc/pair_runs.h): every pair (i <= j) of the element matrix
    is owned by exactly one (lane, slot), a lane's pairs are consecutive columns of ONE block row, the runs fit into the
    64 lanes of a wavefront (T10: 55 pairs / ANCF-3243: 36, one per lane; ANCF-3443: 136 pairs, three per lane)."""
    P = S * (S + 1) // 2
    npl = (P + 63) // 64
    i = np.zeros(64, dtype=np.int32)
    j0 = np.zeros(64, dtype=np.int32)
    cnt = np.zeros(64, dtype=np.int32)
    ip = C.POINTER(C.c_int)
    lanes = shim.pair_runs(S, npl, i.ctypes.data_as(ip), j0.ctypes.data_as(ip), cnt.ctypes.data_as(ip))
    assert lanes <= 64
    assert (cnt[lanes:] == 0).all() and (cnt[:lanes] >= 1).all() and (cnt <= npl).all()
    owner = np.full(P, -1)
    for lane in range(64):
        for n in range(cnt[lane]):
            a, b = int(i[lane]), int(j0[lane]) + n
            assert 0 <= a <= b < S
            p = shim.pair_index_of(S, a, b)
            assert 0 <= p < P and owner[p] == -1
            owner[p] = lane
    assert (owner >= 0).all()
    # the buffer layout: row-major upper triangle
    assert [shim.pair_index_of(S, a, b) for a in range(S) for b in range(a, S)] == list(range(P))
    if S == 16:
        assert npl == 3 and lanes == 51
