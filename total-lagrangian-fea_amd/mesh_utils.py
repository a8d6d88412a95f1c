"""Host plumbing of the T10 path: TetGen readers (reference: lib_utils/cpu_utils.cc:607-754) and the
structured T10 generators used for the BASELINE configs.  Integer outputs are
bit-exact with the reference readers (tests/test_host_plumbing.py)."""
import numpy as np

TETGEN_TO_STANDARD = (0, 1, 2, 3, 6, 7, 9, 5, 8, 4)  # cpu_utils.cc:619
EDGES = ((0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3))  # FEAT10Data.cu:143


def FEAT10_remap_tetgen_indices(tetgen_elem):
    t = np.asarray(tetgen_elem)
    if t.shape[-1] != 10:
        raise ValueError("Element arrays must have size 10 for T10 elements")
    return t[..., list(TETGEN_TO_STANDARD)]


def FEAT10_read_nodes(filename):
    """-> (n_nodes, nodes[n,3]); node ids are shifted by 1 unless the file's smallest id is 0."""
    with open(filename) as f:
        hdr = f.readline().split()
        n, dim = int(hdr[0]), int(hdr[1])
        if dim != 3:
            raise ValueError(f"Only 3D nodes are supported, found {dim}D")
        rows = []
        for _ in range(n):
            p = f.readline().split()
            if len(p) >= 4:
                rows.append((int(p[0]), float(p[1]), float(p[2]), float(p[3])))
    off = 0 if min(r[0] for r in rows) == 0 else 1
    nodes = np.zeros((n, 3))
    for i, x, y, z in rows:
        if 0 <= i - off < n:
            nodes[i - off] = (x, y, z)
    return n, nodes


def FEAT10_read_elements(filename):
    """-> (n_elems, elements[n,10] int32) in the standard mid-node order."""
    with open(filename) as f:
        hdr = f.readline().split()
        m, k = int(hdr[0]), int(hdr[1])
        if k != 10:
            raise ValueError(f"Only T10 elements (10 nodes) are supported, found {k}")
        rows = [[int(p) for p in line.split()[:11]] for line in (f.readline() for _ in range(m)) if line.strip()]
    rows = np.asarray(rows, dtype=np.int64)
    eoff = 0 if rows[:, 0].min() == 0 else 1
    noff = 0 if rows[:, 1:].min() == 0 else 1
    elements = np.zeros((m, 10), dtype=np.int32)
    ids = rows[:, 0] - eoff
    ok = (ids >= 0) & (ids < m)
    elements[ids[ok]] = FEAT10_remap_tetgen_indices(rows[ok, 1:] - noff)
    return m, elements


# 6 tetrahedra per hexahedral cell, all sharing the main diagonal v0-v6 (conforming across cells)
_HEX_TETS = ((0, 1, 2, 6), (0, 2, 3, 6), (0, 3, 7, 6), (0, 7, 4, 6), (0, 4, 5, 6), (0, 5, 1, 6))
_HEX_CORNERS = ((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1))


def structured_t10_box(nx, ny, nz, lx=1.0, ly=1.0, lz=1.0):
    """nx*ny*nz cells x 6 T10 tets on a (2nx+1)(2ny+1)(2nz+1) lattice (corner + mid-edge nodes).
    Returns (nodes[N,3] float64, elements[E,10] int32); unused lattice points are dropped and nodes are
    numbered in lattice (z-major) order, so neighbouring nodes stay close in memory."""
    gx, gy, gz = 2 * nx + 1, 2 * ny + 1, 2 * nz + 1
    # cells in the same (z-major, x fastest) order as the nodes: consecutive elements then gather neighbouring nodes
    # (measured at config C: the thread-per-element residual launch 557 -> 460 us against cells in x-major order)
    import os
    if os.environ.get("TLFEA_BOX_ELEMS") == "xmajor":   # the round-1/2 order, kept for A/B runs
        ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    else:
        kk, jj, ii = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    base = np.stack([ii.ravel(), jj.ravel(), kk.ravel()], axis=1) * 2  # lattice coords of cell corner 0
    corners = np.asarray(_HEX_CORNERS) * 2
    tets = np.asarray(_HEX_TETS)
    # lattice coordinates of the 4 vertices of every tet: [cells, 6, 4, 3]
    v = base[:, None, None, :] + corners[tets][None, :, :, :]
    v = v.reshape(-1, 4, 3)
    # fix orientation so that det J > 0
    d1, d2, d3 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0], v[:, 3] - v[:, 0]
    vol = np.einsum("ij,ij->i", d1, np.cross(d2, d3))
    flip = vol < 0
    v[flip, 1], v[flip, 2] = v[flip, 2].copy(), v[flip, 1].copy()
    mids = np.stack([(v[:, a] + v[:, b]) // 2 for a, b in EDGES], axis=1)
    lat = np.concatenate([v, mids], axis=1)  # [E,10,3]
    if os.environ.get("TLFEA_BOX_ORDER") == "x":   # experiment: x slowest (planes of (2ny+1)(2nz+1) nodes)
        gz = 2 * nz + 1
        lin = (lat[..., 0] * gy + lat[..., 1]) * gz + lat[..., 2]
        used, inv = np.unique(lin.ravel(), return_inverse=True)
        elements = inv.reshape(-1, 10).astype(np.int32)
        zs = (used % gz) * (lz / (2 * nz))
        ys = ((used // gz) % gy) * (ly / (2 * ny))
        xs = (used // (gz * gy)) * (lx / (2 * nx))
        return np.stack([xs, ys, zs], axis=1).astype(np.float64), elements
    lin = (lat[..., 2] * gy + lat[..., 1]) * gx + lat[..., 0]
    used, inv = np.unique(lin.ravel(), return_inverse=True)
    elements = inv.reshape(-1, 10).astype(np.int32)
    xs = (used % gx) * (lx / (2 * nx))
    ys = ((used // gx) % gy) * (ly / (2 * ny))
    zs = (used // (gx * gy)) * (lz / (2 * nz))
    return np.stack([xs, ys, zs], axis=1).astype(np.float64), elements


class GridMeshGenerator:
    """ANCF-3243 beam/net mesh generator (reference: lib_utils/mesh_utils.cc:13-167).  Nodes on an
    (nx+1) x (ny+1) grid with id = j*(nx+1)+i, horizontal elements first; per node 4 coefficient vectors with
    x=[x,1,0,0], y=[1,0,1,0], z=[0,0,0,1] (the beam lies at y=1, as in the reference)."""

    def __init__(self, X, Y, L, include_horizontal=True, include_vertical=True):
        if L <= 0:
            raise ValueError("L must be > 0")
        if abs(round(X / L) * L - X) > 1e-12 or abs(round(Y / L) * L - Y) > 1e-12:
            raise ValueError("X and Y must be exact multiples of L")
        self.X, self.Y, self.L = X, Y, L
        self.nx, self.ny = int(round(X / L)), int(round(Y / L))
        self.include_horizontal = include_horizontal and self.nx > 0
        self.include_vertical = include_vertical and self.ny > 0
        self.nodes, self.elements = [], []

    def node_id(self, i, j):
        if i < 0 or i > self.nx or j < 0 or j > self.ny:
            raise IndexError("(i,j) out of range")
        return j * (self.nx + 1) + i

    def generate_mesh(self):
        self.nodes = [(self.node_id(i, j), i * self.L, j * self.L) for j in range(self.ny + 1)
                      for i in range(self.nx + 1)]
        self.elements = []
        if self.include_horizontal:
            self.elements += [(self.node_id(i, j), self.node_id(i + 1, j)) for j in range(self.ny + 1)
                              for i in range(self.nx)]
        if self.include_vertical:
            self.elements += [(self.node_id(i, j), self.node_id(i, j + 1)) for i in range(self.nx + 1)
                              for j in range(self.ny)]

    def get_num_nodes(self):
        return len(self.nodes)

    def get_num_elements(self):
        return len(self.elements)

    def get_coordinates(self):
        n = len(self.nodes)
        x, y, z = np.zeros(4 * n), np.zeros(4 * n), np.zeros(4 * n)
        for nid, xv, _yv in self.nodes:
            x[4 * nid:4 * nid + 4] = (xv, 1.0, 0.0, 0.0)
            y[4 * nid:4 * nid + 4] = (1.0, 0.0, 1.0, 0.0)
            z[4 * nid:4 * nid + 4] = (0.0, 0.0, 0.0, 1.0)
        return x, y, z

    def get_element_connectivity(self):
        return np.asarray(self.elements, dtype=np.int32).reshape(-1, 2)


def ANCF3443_generate_beam_coordinates(n_beam):
    """Strip of n_beam ANCF-3443 shells, 2 x 1 each (reference: lib_utils/cpu_utils.cc:476-595; the arrays are
    pinned by lib_utest/utest_utils.cc:124-221).  -> (x12, y12, z12, connectivity[n_beam,4])"""
    n_nodes = 4 + 2 * (n_beam - 1)
    x, y, z = np.zeros(4 * n_nodes), np.zeros(4 * n_nodes), np.zeros(4 * n_nodes)
    pos = [(0.0, 0.0), (2.0, 0.0), (2.0, 1.0), (0.0, 1.0)]
    for i in range(1, n_beam):
        pos += [(2.0 * (i + 1), 0.0), (2.0 * (i + 1), 1.0)]
    for nid, (xv, yv) in enumerate(pos):
        x[4 * nid:4 * nid + 4] = (xv, 1.0, 0.0, 0.0)
        y[4 * nid:4 * nid + 4] = (yv, 0.0, 1.0, 0.0)
        z[4 * nid:4 * nid + 4] = (0.0, 0.0, 0.0, 1.0)
    conn = np.zeros((n_beam, 4), dtype=np.int32)
    conn[0] = (0, 1, 2, 3)
    for i in range(1, n_beam):
        conn[i] = (1, 4, 5, 2) if i == 1 else (4 + (i - 2) * 2, 4 + (i - 1) * 2, 4 + (i - 1) * 2 + 1, 5 + (i - 2) * 2)
    return x, y, z, conn


def _b12(kind, L, W, H):
    import ctypes as C

    from . import binding
    S = 8 if kind == 3243 else 16
    out = np.zeros(S * S)
    binding.check(binding.load_library().tlfea_ancf_b12_matrix(kind, C.c_double(L), C.c_double(W), C.c_double(H),
                                                               binding.dp(out)))
    return out


def ANCF3243_B12_matrix(L, W, H):
    """(B^T)^-1 of one beam as an 8 x 8 matrix (cpu_utils.cc:125-188); host arithmetic of the C-ABI, no GPU"""
    return _b12(3243, L, W, H).reshape(8, 8).T.copy()


def ANCF3443_B12_matrix(L, W, H):
    """(B^T)^-1 of one shell as a 16 x 16 matrix (cpu_utils.cc:211-420)"""
    return _b12(3443, L, W, H).reshape(16, 16).T.copy()


def ANCF3243_B12_matrix_flat_per_element(L, W, H):
    """column-major 8 x 8 blocks, one per element (cpu_utils.cc:190-209)"""
    return np.concatenate([_b12(3243, l, w, h) for l, w, h in zip(L, W, H)])


def ANCF3443_B12_matrix_flat_per_element(L, W, H):
    """column-major 16 x 16 blocks, one per element (cpu_utils.cc:422-441)"""
    return np.concatenate([_b12(3443, l, w, h) for l, w, h in zip(L, W, H)])


def ANCF3243_generate_beam_coordinates(n_beam):
    """Chain of n_beam beams of length 2 along x (cpu_utils.cc:443-474) -> (x12, y12, z12), 4 (n_beam + 1) each"""
    n = n_beam + 1
    x, y, z = np.zeros(4 * n), np.zeros(4 * n), np.zeros(4 * n)
    x[0::4] = -1.0 + 2.0 * np.arange(n)
    x[1::4] = 1.0
    y[0::4] = 1.0
    y[2::4] = 1.0
    z[3::4] = 1.0
    return x, y, z


def ANCF3243_calculate_offsets(n_beam):
    """cpu_utils.cc:597-605"""
    start = np.arange(n_beam, dtype=np.int32) * 4
    return start, start + 7


def structured_3443_plate(nx, ny, L, W):
    """nx x ny ANCF-3443 shell elements of size L x W in the z=0 plane (BASELINE config D generator).
    Node id = j*(nx+1)+i; element nodes P1(i,j) P2(i+1,j) P3(i+1,j+1) P4(i,j+1) (cpu_utils.cc:213-217 order);
    coefficients per node r=(x,y,0), r_u=(1,0,0), r_v=(0,1,0), r_w=(0,0,1).  -> (x12, y12, z12, conn[E,4])"""
    ii, jj = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing="xy")
    n = (nx + 1) * (ny + 1)
    x, y, z = np.zeros(4 * n), np.zeros(4 * n), np.zeros(4 * n)
    x[0::4] = (ii * L).reshape(-1)
    y[0::4] = (jj * W).reshape(-1)
    x[1::4] = 1.0
    y[2::4] = 1.0
    z[3::4] = 1.0
    ei, ej = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    n0 = (ej * (nx + 1) + ei).reshape(-1)
    conn = np.stack([n0, n0 + 1, n0 + nx + 2, n0 + nx + 1], axis=1).astype(np.int32)
    return x, y, z, conn


# ---- general linear constraints + ANCF mesh files (mesh_utils.h:105-245, mesh_utils.cc:170-1010) ----------------
class LinearConstraintCSR:
    """c[row] = sum_j values[j] * dof(columns[j]) - rhs[row]; columns = 3*coef + component (mesh_utils.h:104-127)."""

    def __init__(self, offsets=None, columns=None, values=None, rhs=None):
        self.offsets = np.asarray([] if offsets is None else offsets, dtype=np.int32)
        self.columns = np.asarray([] if columns is None else columns, dtype=np.int32)
        self.values = np.asarray([] if values is None else values, dtype=np.float64)
        self.rhs = np.asarray([] if rhs is None else rhs, dtype=np.float64)

    def NumRows(self):
        return int(self.rhs.size)

    def NumNonZeros(self):
        return int(self.columns.size)

    def Empty(self):
        return self.rhs.size == 0


class LinearConstraintBuilder:
    """Row-by-row builder of a LinearConstraintCSR (mesh_utils.cc:173-246): zero coefficients are dropped, an empty
    row or a column outside [0, n_dofs) is an error."""

    def __init__(self, n_dofs, initial=None):
        if n_dofs <= 0:
            raise ValueError("LinearConstraintBuilder: n_dofs must be > 0")
        self._n_dofs = int(n_dofs)
        self._offsets, self._columns, self._values, self._rhs = [0], [], [], []
        if initial is not None:
            if len(initial.offsets) != initial.NumRows() + 1:
                raise ValueError("LinearConstraintBuilder: initial offsets size mismatch")
            if len(initial.values) != initial.NumNonZeros():
                raise ValueError("LinearConstraintBuilder: initial nnz mismatch")
            if initial.offsets[0] != 0 or initial.offsets[-1] != initial.NumNonZeros():
                raise ValueError("LinearConstraintBuilder: initial CSR offsets invalid")
            self._offsets = [int(v) for v in initial.offsets]
            self._columns = [int(v) for v in initial.columns]
            self._values = [float(v) for v in initial.values]
            self._rhs = [float(v) for v in initial.rhs]

    def n_dofs(self):
        return self._n_dofs

    def num_rows(self):
        return len(self._rhs)

    def nnz(self):
        return len(self._columns)

    def AddRow(self, entries, rhs):
        if len(entries) == 0:
            raise ValueError("LinearConstraintBuilder::AddRow: empty row")
        for col, val in entries:
            if col < 0 or col >= self._n_dofs:
                raise IndexError("LinearConstraintBuilder::AddRow: col out of range")
            if val == 0.0:
                continue
            self._columns.append(int(col))
            self._values.append(float(val))
        self._rhs.append(float(rhs))
        self._offsets.append(len(self._columns))
        return len(self._rhs) - 1

    def AddFixedDof(self, col, rhs):
        return self.AddRow([(col, 1.0)], rhs)

    def ToCSR(self):
        return LinearConstraintCSR(self._offsets, self._columns, self._values, self._rhs)


def _ancf_dof_col(node_id, coef_slot, component):
    return (node_id * 4 + coef_slot) * 3 + component


def AppendANCFVectorEqualityConstraint(builder, node_a, node_b, coef_slot):
    """r(b, slot) - r(a, slot) = 0, three rows (mesh_utils.cc:262-275 / :330-343)."""
    if coef_slot < 0 or coef_slot > 3:
        raise IndexError("AppendANCFVectorEqualityConstraint: coef_slot out of range")
    for c in range(3):
        builder.AddRow([(_ancf_dof_col(node_b, coef_slot, c), 1.0), (_ancf_dof_col(node_a, coef_slot, c), -1.0)], 0.0)


def AppendANCFVectorWeldedConstraint(builder, node_a, node_b, coef_slot, Q):
    """r(b, slot) - Q r(a, slot) = 0 with Q row-major 3x3 (mesh_utils.cc:277-299 / :345-367)."""
    if coef_slot < 0 or coef_slot > 3:
        raise IndexError("AppendANCFVectorWeldedConstraint: coef_slot out of range")
    Q = np.asarray(Q, dtype=np.float64).reshape(3, 3)
    for row in range(3):
        entries = [(_ancf_dof_col(node_b, coef_slot, row), 1.0)]
        for k in range(3):
            w = -Q[row, k]
            if w == 0.0:
                continue
            entries.append((_ancf_dof_col(node_a, coef_slot, k), w))
        builder.AddRow(entries, 0.0)


def AppendANCFFixedCoefficient(builder, coef_index, x12_ref, y12_ref, z12_ref):
    """Component-wise equality of one coefficient to the reference arrays (mesh_utils.cc:301-315)."""
    if coef_index < 0 or coef_index >= len(x12_ref):
        raise IndexError("AppendANCFFixedCoefficient: coef_index out of range")
    builder.AddFixedDof(coef_index * 3 + 0, x12_ref[coef_index])
    builder.AddFixedDof(coef_index * 3 + 1, y12_ref[coef_index])
    builder.AddFixedDof(coef_index * 3 + 2, z12_ref[coef_index])


# the reference has one copy of each helper per element family; the DOF numbering is the same
AppendANCF3243VectorEqualityConstraint = AppendANCF3443VectorEqualityConstraint = AppendANCFVectorEqualityConstraint
AppendANCF3243VectorWeldedConstraint = AppendANCF3443VectorWeldedConstraint = AppendANCFVectorWeldedConstraint
AppendANCF3243FixedCoefficient = AppendANCFFixedCoefficient


class ANCFMesh:
    """ANCF3243Mesh / ANCF3443Mesh (mesh_utils.h:165-214)."""

    def __init__(self):
        self.version = 0
        self.grid_nx = self.grid_ny = self.grid_L = self.grid_origin = None
        self.n_nodes = self.n_elements = 0
        self.node_family, self.element_family = [], []
        self.x12 = self.y12 = self.z12 = None
        self.element_L = self.element_W = self.element_H = None
        self.element_connectivity = None
        self.constraints = LinearConstraintCSR()


def _records(path):
    with open(path) as fh:
        for line in fh:
            line = line.split("#", 1)[0].strip()
            if line:
                yield line


def _read_ancf_mesh(path, tag, nn):
    """Shared reader of `.ancf3243mesh` (nn = 2) and `.ancf3443mesh` (nn = 4): sections `nodes N` (id family x0..x3
    y0..y3 z0..z3), `elements M` (3243: id family n0 n1; 3443: id family L W H n0..n3), optional `constraints K`
    (`pinned a b` -> position equality; `welded a b q00..q22` -> position equality + Q-mapped gradients).
    Raises ValueError with the reference's message on malformed input (the reference returns false + message)."""
    fn = "ReadANCF%sMeshFromFile" % tag
    rec = _records(path)
    out = ANCFMesh()

    def nxt(what):
        try:
            return next(rec).split()
        except StopIteration:
            raise ValueError("%s: %s" % (fn, what))

    t = nxt("empty file")
    if len(t) != 2 or t[0] != "ancf%s_mesh" % tag:
        raise ValueError("%s: expected header 'ancf%s_mesh <version>'" % (fn, tag))
    try:
        out.version = int(t[1])
    except ValueError:
        out.version = 0
    if out.version <= 0:
        raise ValueError("%s: invalid mesh version" % fn)
    t = nxt("missing nodes section")
    if t[0] == "grid" and tag == "3243":
        if len(t) != 11 or t[1] != "nx" or t[3] != "ny" or t[5] != "L" or t[7] != "origin":
            raise ValueError("%s: invalid grid line" % fn)
        out.grid_nx, out.grid_ny, out.grid_L = int(t[2]), int(t[4]), float(t[6])
        out.grid_origin = np.array([float(t[8]), float(t[9]), float(t[10])])
        t = nxt("missing nodes section")
    elif t[0] in ("tire", "meta") and tag == "3443":
        t = nxt("missing nodes section")
    if len(t) != 2 or t[0] != "nodes" or not t[1].lstrip("-").isdigit() or int(t[1]) <= 0:
        raise ValueError("%s: invalid nodes header" % fn)
    n_nodes = out.n_nodes = int(t[1])
    out.node_family = [""] * n_nodes
    out.x12, out.y12, out.z12 = np.zeros(4 * n_nodes), np.zeros(4 * n_nodes), np.zeros(4 * n_nodes)
    seen = np.zeros(n_nodes, dtype=bool)
    for _ in range(n_nodes):
        t = nxt("unexpected EOF in nodes")
        if len(t) != 14:
            raise ValueError("%s: invalid node line (expected 14 tokens)" % fn)
        nid = int(t[0])
        if nid < 0 or nid >= n_nodes:
            raise ValueError("%s: node id out of range" % fn)
        if seen[nid]:
            raise ValueError("%s: duplicate node id" % fn)
        seen[nid] = True
        out.node_family[nid] = t[1]
        v = [float(a) for a in t[2:14]]
        out.x12[4 * nid:4 * nid + 4] = v[0:4]
        out.y12[4 * nid:4 * nid + 4] = v[4:8]
        out.z12[4 * nid:4 * nid + 4] = v[8:12]
    t = nxt("missing elements section")
    if len(t) != 2 or t[0] != "elements" or int(t[1]) <= 0:
        raise ValueError("%s: invalid elements header" % fn)
    n_el = out.n_elements = int(t[1])
    out.element_connectivity = np.zeros((n_el, nn), dtype=np.int32)
    out.element_family = [""] * n_el
    if nn == 4:
        out.element_L, out.element_W, out.element_H = np.zeros(n_el), np.zeros(n_el), np.zeros(n_el)
    seen = np.zeros(n_el, dtype=bool)
    ntok = 4 if nn == 2 else 9
    for _ in range(n_el):
        t = nxt("unexpected EOF in elements")
        if len(t) != ntok:
            raise ValueError("%s: invalid element line (expected %d tokens)" % (fn, ntok))
        eid = int(t[0])
        if eid < 0 or eid >= n_el:
            raise ValueError("%s: element id out of range" % fn)
        if seen[eid]:
            raise ValueError("%s: duplicate element id" % fn)
        seen[eid] = True
        out.element_family[eid] = t[1]
        if nn == 4:
            out.element_L[eid], out.element_W[eid], out.element_H[eid] = float(t[2]), float(t[3]), float(t[4])
        nodes = [int(a) for a in t[ntok - nn:]]
        if min(nodes) < 0 or max(nodes) >= n_nodes:
            raise ValueError("%s: element node id out of range" % fn)
        out.element_connectivity[eid] = nodes
    try:
        t = next(rec).split()
    except StopIteration:
        return out  # no constraints section
    if len(t) != 2 or t[0] != "constraints" or int(t[1]) < 0:
        raise ValueError("%s: invalid constraints header" % fn)
    builder = LinearConstraintBuilder(12 * n_nodes)
    for _ in range(int(t[1])):
        t = nxt("unexpected EOF in constraints")
        if t[0] == "pinned":
            if len(t) != 3:
                raise ValueError("%s: pinned expects 'pinned a b'" % fn)
            a, b = int(t[1]), int(t[2])
            if min(a, b) < 0 or max(a, b) >= n_nodes:
                raise ValueError("%s: pinned node id out of range" % fn)
            AppendANCFVectorEqualityConstraint(builder, a, b, 0)
        elif t[0] == "welded":
            if len(t) != 12:
                raise ValueError("%s: welded expects 'welded a b q00..q22'" % fn)
            a, b = int(t[1]), int(t[2])
            if min(a, b) < 0 or max(a, b) >= n_nodes:
                raise ValueError("%s: welded node id out of range" % fn)
            Q = np.array([float(v) for v in t[3:12]]).reshape(3, 3)
            AppendANCFVectorEqualityConstraint(builder, a, b, 0)      # position continuity (no rotation)
            for slot in (1, 2, 3):                                    # gradient continuity with the Q mapping
                AppendANCFVectorWeldedConstraint(builder, a, b, slot, Q)
        else:
            raise ValueError("%s: unknown constraint type '%s'" % (fn, t[0]))
    out.constraints = builder.ToCSR()
    return out


def ReadANCF3243MeshFromFile(path):
    """mesh_utils.cc:444-736"""
    return _read_ancf_mesh(path, "3243", 2)


def ReadANCF3443MeshFromFile(path):
    """mesh_utils.cc:738-1010"""
    return _read_ancf_mesh(path, "3443", 4)
