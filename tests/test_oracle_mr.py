"""Mooney-Rivlin sub-path of the oracle against derivatives of the strain energy derived symbolically
(tools/gen_golden_mr.py, sympy): the reference holds no vector for this material, so this is an independent consistency
check of the restated closed forms (MooneyRivlin.cuh:45-225 -> oracle/tlfea_oracle.c mr_P / mr_tangent), not a pin by the
reference: **parity unpinned** stands."""
import os

import numpy as np
import pytest

from oracle import orc
from tests.helpers import load_mesh

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "mr_energy_derivatives.npz"))


@pytest.mark.parametrize("k", range(len(G["F"])))
def test_mr_stress_and_tangent_are_derivatives_of_the_energy(k):
    F0, (mu10, mu01, kappa), P_ref, A_ref = G["F"][k], G["params"][k], G["P"][k], G["A"][k]
    X, conn = load_mesh("cube")                                   # 6 elements; a homogeneous map x = F0 X has F = F0 everywhere
    o = orc.T10Oracle(X, conn, orc.mooney_rivlin(mu10, mu01, kappa))
    o.calc_dndu_pre()
    x = X @ F0.T
    o.x, o.y, o.z = (np.ascontiguousarray(x[:, c]) for c in range(3))
    F, P, _, _ = o.compute_p(None)
    # buffers are column-major 3x3 per (element, point): entry [i + 3 j] = M[i][j]
    Fm = F.reshape(-1, 3, 3).transpose(0, 2, 1)
    Pm = P.reshape(-1, 3, 3).transpose(0, 2, 1)
    assert np.abs(Fm - F0).max() < 1e-13
    # F itself carries ~1e-16 of round-off from the reference gradients, which the moduli amplify: the floor is
    # (stiffness) x (a few ulp), next to 1e-12 of the stress
    assert np.abs(Pm - P_ref).max() <= 1e-12 * np.abs(P_ref).max() + 256 * np.finfo(float).eps * (kappa + mu10 + mu01)
    # K_e[(a,d),(b,e)] = sum_q sum_JL gradN_a[J] A[d][J][e][L] gradN_b[L] detJ_q w_q from the SYMBOLIC tangent and the pinned
    # reference gradients (golden-checked in tests/test_oracle_golden.py)
    Ke, _ = o.element_tangents()
    g = o.gradN_a_d()                                              # [E, 5, 10, 3]
    dV = o.detJ * o.qw[None, :]
    K_ref = np.einsum("eqaJ,dJfL,eqbL,eq->eadbf", g, A_ref, g, dV).reshape(o.E, 30, 30)
    assert np.abs(Ke - K_ref).max() <= 1e-11 * np.abs(K_ref).max()
