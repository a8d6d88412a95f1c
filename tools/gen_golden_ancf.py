#!/usr/bin/env python3
"""Golden vectors for the ANCF-3243 beam and ANCF-3443 shell from the REFERENCE's NumPy prototypes
(test-scripts/3243-beam/f-form-3243-nesterov.py, test-scripts/3443-shell/f-form-3443-nesterov.py).
Those files are script-style (a whole simulation runs at import), so only their set-up part (everything before
`for step in range(Nt):`) and the two force functions defined at the top of that loop are executed, in a
private namespace.  Runs only in the build container; outputs are committed under tests/golden/.
The prototypes hold mass, ds/du, detJ and f_int but NO tangent."""
import contextlib
import io
import os
import textwrap

import numpy as np

os.environ.setdefault("MPLBACKEND", "Agg")
REF = "/root/reference/test-scripts"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_prototype(path):
    src = open(path).read()
    head, tail = src.split("for step in range(Nt):", 1)
    ns = {"__name__": "ref_proto"}
    with contextlib.redirect_stdout(io.StringIO()):
        exec(compile(head, path, "exec"), ns)
        body = tail.split("def alm_nesterov_step", 1)[0]
        exec(compile(textwrap.dedent(body), path + ":force", "exec"), ns)
    return ns


def main():
    for tag, rel, S in (("ancf3243", "3243-beam/f-form-3243-nesterov.py", 8),
                        ("ancf3443", "3443-shell/f-form-3443-nesterov.py", 16)):
        ns = load_prototype(os.path.join(REF, rel))
        gx, gy, gz = ns["gauss_xi"], ns["gauss_eta"], ns["gauss_zeta"]
        Q = len(gx) * len(gy) * len(gz)
        ds = np.zeros((Q, S, 3))
        dj = np.zeros(Q)
        for ix in range(len(gx)):
            for ie in range(len(gy)):
                for iz in range(len(gz)):
                    q = (ix * len(gy) + ie) * len(gz) + iz
                    ds[q] = ns["ds_du_pre"][(ix, ie, iz)]
                    dj[q] = ns["detJ_pre"][(ix, ie, iz)]
        x12, y12, z12 = (np.array(ns[k], dtype=float) for k in ("x12", "y12", "z12"))
        rng = np.random.default_rng(12345)
        xp, yp, zp = (a + rng.normal(0.0, 1e-3, a.shape) for a in (x12, y12, z12))
        with contextlib.redirect_stdout(io.StringIO()):
            f_int = ns["compute_internal_force"](xp, yp, zp)
        extra = {}
        if tag == "ancf3443":
            extra["element_connectivity"] = np.asarray(ns["element_connectivity"], dtype=np.int32)
        np.savez_compressed(os.path.join(OUT, f"{tag}_proto.npz"), L=ns["L"], W=ns["W"], H=ns["H"], E=ns["E"],
                            nu=ns["nu"], rho0=ns["rho0"], B_inv=np.asarray(ns["B_inv"]), mass=np.asarray(ns["m"]),
                            ds_du=ds, detJ_uvw=dj, x12=x12, y12=y12, z12=z12, xp=xp, yp=yp, zp=zp, f_int=f_int,
                            n_elem=int(ns.get("n_beam", ns.get("n_shell", 0))), **extra)
        print(tag, "S", S, "Q", Q, "N_coef", len(x12), "|f_int|", np.linalg.norm(f_int), "detJ", dj.min(), dj.max())


if __name__ == "__main__":
    main()
