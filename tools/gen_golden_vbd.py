#!/usr/bin/env python3
"""Golden vectors for the per-node terms of SyncedVBDSolver from the REFERENCE's NumPy prototype
test-scripts/vbd_proto/alm_vbd_t10_svk.py (importable: its simulation sits under main()): for every node of
beam_3x2x1 at a seeded perturbed state, the internal-force block f_i = sum P h_a dV and the diagonal tangent block
sum K_aa over the incident elements (local_internal_force_and_Kii, :175-222), plus what its greedy colouring yields
(colour count only: its node order comes from numpy's argsort, the C++ reference's from std::sort, so the colours
themselves differ between the two and are checked through validate_coloring instead).
Runs only in the build container; the output is committed under tests/golden/."""
import contextlib
import importlib.util
import io
import os

import numpy as np

os.environ.setdefault("MPLBACKEND", "Agg")
REF = "/root/reference/test-scripts/vbd_proto/alm_vbd_t10_svk.py"
HERE = os.path.dirname(os.path.abspath(__file__))
MESH = os.path.join(HERE, "..", "tests", "golden", "meshes", "beam_3x2x1.1")
OUT = os.path.join(HERE, "..", "tests", "golden", "vbd_proto_beam_3x2x1.npz")


def main():
    spec = importlib.util.spec_from_file_location("ref_vbd_proto", REF)
    mod = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(mod)
    ids, X = mod.read_tetgen_node(MESH + ".node")
    conn = mod.read_tetgen_ele(MESH + ".ele")
    id2idx = {nid: i for i, nid in enumerate(ids)}  # as its main() does (:508-509)
    conn = np.vectorize(lambda nid: id2idx[nid])(conn)
    conn = mod.reorder_t10_elements_to_canon(X, conn)
    pre = mod.tet10_precompute_reference_mesh(X, conn)
    E_mod, nu = 7e8, 0.33
    mu = E_mod / (2 * (1 + nu))
    lam = E_mod * nu / ((1 + nu) * (1 - 2 * nu))
    rng = np.random.default_rng(12345)
    x = X + rng.normal(0.0, 1e-3, X.shape)
    inc = mod.build_incidence(conn, X.shape[0])
    f = np.zeros((X.shape[0], 3))
    K = np.zeros((X.shape[0], 3, 3))
    for i in range(X.shape[0]):
        f[i], K[i] = mod.local_internal_force_and_Kii(i, x, conn, pre, inc[i], lam, mu)
    adj = mod.build_vertex_adjacency(conn, X.shape[0])
    colors = mod.greedy_vertex_coloring(adj)
    mod.validate_coloring(conn, colors)
    np.savez_compressed(OUT, x=x, f_i=f, K_ii=K, lam=lam, mu=mu, n_colors=int(colors.max()) + 1,
                        degree=np.array([len(a) for a in adj], dtype=np.int32))
    print("nodes", X.shape[0], "elements", conn.shape[0], "|f|", np.linalg.norm(f), "|K|", np.linalg.norm(K), "colors",
          int(colors.max()) + 1)


if __name__ == "__main__":
    main()
