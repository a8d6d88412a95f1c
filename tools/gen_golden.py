#!/usr/bin/env python3
"""Generate golden vectors for the T10 hot path from the REFERENCE's own NumPy prototypes.

Runs ONLY in the build container (needs /root/reference); the output .npz files are committed
under tests/golden/ and are the fixtures that pin oracle/ (see oracle/README.md).
The reference scripts are imported with importlib (their simulation code sits under
`if __name__ == "__main__"`), never copied.

Reference functions exercised (test-scripts/T10-tets/):
  f-form-T10-beam-newton.py        : tet10_precompute_reference_mesh(:85), tet10_internal_force_mesh(:121),
                                     tet10_consistent_mass_mesh(:160), tet10_tangent_svk(:221),
                                     newton_inner(:284), alm_newton_step(:335)
  f-form-T10-beam-newton-damped.py : tet10_internal_force_mesh_damped(:143), tet10_viscous_tangent(:316)
  tet_mesh_reader.py               : read_node(:9), read_ele(:20)  (1-based TetGen ids + mid-node remap)
"""
import contextlib
import importlib.util
import io
import os
import sys

import numpy as np

os.environ.setdefault("MPLBACKEND", "Agg")
REF = "/root/reference"
T10 = os.path.join(REF, "test-scripts", "T10-tets")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, T10)


def load(name, fname):
    spec = importlib.util.spec_from_file_location(name, os.path.join(T10, fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    nwt = load("ref_newton", "f-form-T10-beam-newton.py")
    dmp = load("ref_damped", "f-form-T10-beam-newton-damped.py")

    meshes = {
        "cube": "data/meshes/T10/cube.1",
        "beam_3x2x1": "data/meshes/T10/beam_3x2x1.1",
        "res2": "data/meshes/T10/resolution/beam_3x2x1_res2.1",
    }
    lam, mu, rho0 = nwt.lam, nwt.mu, nwt.rho0
    eta_d, lam_d = dmp.eta_damp, dmp.lam_damp

    for tag, rel in meshes.items():
        X = nwt.read_node(os.path.join(REF, rel + ".node"))
        conn = nwt.read_ele(os.path.join(REF, rel + ".ele")).astype(np.int32)
        E, N = conn.shape[0], X.shape[0]
        pre = quiet(nwt.tet10_precompute_reference_mesh, X, conn)
        gradN = np.array([[q["grad_N"] for q in pe] for pe in pre])  # [E,5,10,3]
        detJ = np.array([[q["detJ"] for q in pe] for pe in pre])     # [E,5]
        wq = np.array([q["w"] for q in pre[0]])

        rng = np.random.default_rng(12345)
        x = X + rng.normal(0.0, 1e-3, size=X.shape)
        v = rng.normal(0.0, 1e-1, size=X.shape)

        f_int = nwt.tet10_internal_force_mesh(x, conn, pre, lam, mu)
        Ke = np.array([nwt.tet10_tangent_svk(x[conn[e]], pre[e], lam, mu) for e in range(E)])
        f_int_damped = dmp.tet10_internal_force_mesh_damped(x, v, conn, pre, lam, mu, eta_d, lam_d)
        Cvis = np.array([dmp.tet10_viscous_tangent(x[conn[e]], pre[e], eta_d, lam_d) for e in range(E)])
        M_full = nwt.tet10_consistent_mass_mesh(X, conn, rho0)
        M_scalar = M_full[0::3, 0::3].copy()  # node x node scalar consistent mass

        np.savez_compressed(
            os.path.join(OUT, f"t10_{tag}.npz"),
            X=X, conn=conn, gradN=gradN, detJ=detJ, wq=wq, x=x, v=v,
            lam=lam, mu=mu, rho0=rho0, eta_damp=eta_d, lam_damp=lam_d,
            f_int=f_int, Ke=Ke, f_int_damped=f_int_damped, Cvis=Cvis, M_scalar=M_scalar)
        print(tag, "E", E, "N", N, "min detJ", detJ.min(), "|f_int|", np.linalg.norm(f_int))

    # ---- ALM/Newton steps on beam_3x2x1 exactly as the prototype's __main__ sets them up (:373-397),
    #      but converged to machine precision (tol_R, tol_step passed through newton_inner's own args)
    #      so that the root does not depend on the prototype's looser stopping rule.
    X = nwt.read_node(os.path.join(REF, "data/meshes/T10/beam_3x2x1.1.node"))
    conn = nwt.read_ele(os.path.join(REF, "data/meshes/T10/beam_3x2x1.1.ele"))
    nwt.X_nodes = X  # constraint()/constraint_jacobian() read this module global (:201-215)
    pre = quiet(nwt.tet10_precompute_reference_mesh, X, conn)
    M_full = nwt.tet10_consistent_mass_mesh(X, conn, rho0)
    f_ext = np.zeros(3 * X.shape[0])
    f_ext[3 * 19 + 0] = 1000.0
    h, rho_bb = 1e-3, 1e14
    fixed = nwt.get_fixed_nodes(X)
    q_prev = X.flatten().copy()
    v_prev = np.zeros_like(q_prev)
    v_guess = v_prev.copy()
    lam_guess = np.zeros(3 * len(fixed))
    xs, vs, outers, inners = [], [], [], []
    for step in range(3):
        v = v_guess.copy()
        lam_mult = lam_guess.copy()
        n_outer = n_inner = 0
        for outer in range(5):
            n_outer += 1
            v, nit = quiet(nwt.newton_inner, v, q_prev, v_prev, M_full, nwt.tet10_internal_force_mesh,
                           f_ext, h, X, conn, pre, lam, mu, lam_mult, rho_bb,
                           max_newton=30, tol_R=1e-14, tol_step=1e-15)
            n_inner += nit
            cA = nwt.constraint(q_prev + h * v)
            lam_mult += rho_bb * cA
            if np.linalg.norm(cA) < 1e-6:
                break
        v_guess, lam_guess = v.copy(), lam_mult.copy()
        q_prev = q_prev + h * v_guess
        v_prev = v_guess.copy()
        xs.append(q_prev.reshape(-1, 3).copy()); vs.append(v_guess.copy())
        outers.append(n_outer); inners.append(n_inner)
        print("step", step, "outer", n_outer, "newton", n_inner, "x19", q_prev[3 * 19:3 * 19 + 3])
    np.savez_compressed(
        os.path.join(OUT, "t10_beam_3x2x1_newton.npz"),
        X=X, conn=conn.astype(np.int32), fixed=fixed.astype(np.int32), f_ext=f_ext, h=h, rho=rho_bb,
        lam=lam, mu=mu, rho0=rho0, x_steps=np.array(xs), v_steps=np.array(vs),
        outer_iters=np.array(outers), newton_iters=np.array(inners), lam_mult=lam_guess)


if __name__ == "__main__":
    main()
