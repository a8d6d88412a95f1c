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
    ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
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
