#!/usr/bin/env python3
"""Independent check vectors for the compressible Mooney-Rivlin sub-path, which the reference
holds no fixture for: P = dW/dF and A = d2W/dF2 derived SYMBOLICALLY (sympy) from the energy the reference's formulas
(MooneyRivlin.cuh:45-225) belong to,

    W(F) = mu10 (J^(-2/3) I1 - 3) + mu01 (J^(-4/3) I2 - 3) + kappa/2 (J - 1)^2,
    C = F^T F,  I1 = tr C,  I2 = ((tr C)^2 - tr C^2) / 2,  J = det F,

and evaluated at a few deformation gradients -> tests/golden/mr_energy_derivatives.npz (numbers only).  It does not turn
"parity unpinned" into pinned (no output of the reference is involved); it shows that the oracle's restatement of the
reference's closed forms IS the gradient / Hessian of this energy.        usage: python tools/gen_golden_mr.py"""
import os

import numpy as np
import sympy as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

Fs = sp.Matrix(3, 3, sp.symbols("F0:9", real=True))
mu10, mu01, kappa = sp.symbols("mu10 mu01 kappa", positive=True)
C = Fs.T * Fs
I1 = C.trace()
I2 = (I1**2 - (C * C).trace()) / 2
J = Fs.det()
W = mu10 * (J ** sp.Rational(-2, 3) * I1 - 3) + mu01 * (J ** sp.Rational(-4, 3) * I2 - 3) + kappa / 2 * (J - 1) ** 2
fvars = list(Fs)
P = [sp.diff(W, f) for f in fvars]                                   # P[3 i + j] = dW/dF_ij
A = [[sp.diff(p, f) for f in fvars] for p in P]                      # A[3 i + j][3 k + l] = dP_ij/dF_kl
fP = sp.lambdify(fvars + [mu10, mu01, kappa], P, modules="mpmath")
fA = sp.lambdify(fvars + [mu10, mu01, kappa], A, modules="mpmath")

import mpmath  # noqa: E402

mpmath.mp.dps = 40
rng = np.random.default_rng(2024)
samples = [np.eye(3)]
for amp in (1e-3, 5e-2, 0.2, 0.35):
    for _ in range(2):
        samples.append(np.eye(3) + amp * rng.normal(size=(3, 3)))
samples.append(np.diag([1.3, 0.8, 1.05]) @ (np.eye(3) + 0.1 * rng.normal(size=(3, 3))))
samples = [F for F in samples if np.linalg.det(F) > 0.2]
params = [(3.0e8 / 2.8 / 2 * 0.8, 3.0e8 / 2.8 / 2 * 0.2, 5.0e8),      # tests/helpers.py "mr"
          (3.0e8 / 2.8 / 2, 0.0, 5.0e8),                              # "neo": config B's material
          (4.0e7, 1.0e7, 5.0e8)]                                      # "mr_damped" elastic part
out_F, out_prm, out_P, out_A = [], [], [], []
for prm in params:
    for F in samples:
        args = [mpmath.mpf(float(v)) for v in F.reshape(-1)] + [mpmath.mpf(p) for p in prm]
        out_F.append(F)
        out_prm.append(prm)
        out_P.append(np.array([float(v) for v in fP(*args)]).reshape(3, 3))
        out_A.append(np.array([[float(v) for v in row] for row in fA(*args)]).reshape(3, 3, 3, 3))
dst = os.path.join(ROOT, "tests", "golden", "mr_energy_derivatives.npz")
np.savez_compressed(dst, F=np.array(out_F), params=np.array(out_prm), P=np.array(out_P), A=np.array(out_A))
print("wrote", dst, len(out_F), "samples")
