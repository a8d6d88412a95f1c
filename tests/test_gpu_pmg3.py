"""Third level of the p-multigrid cycle (rigid-body-mode aggregates of the vertex level): Galerkin operator against
P2^T Hc P2 built on the host, definiteness, and the solve against the Chebyshev-preconditioned one.  The level is opt-in
(TLFEA_PMG_LEVELS=3): measured at config C it trades 10-12 % cheaper CG iterations for 10-17 % more of them."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mesh", ["res4", "bunny"])
def test_third_level_galerkin_and_solve(mesh):
    env = dict(os.environ, TLFEA_PMG_LEVELS="3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pmg3_worker.py"), mesh], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["precond"] == 2 and out["n_aggregates"] > 0 and out["n_aggregates"] * 4 < out["n_vertex"], out
    assert out["galerkin_relerr"] < 1e-12 and out["pattern_covers"], out
    assert out["symmetry"] < 1e-12 and out["min_eig_over_max"] > 0.0, out
    assert out["rel_pmg3"] < 1e-12 and out["solution_relerr"] < 1e-8, out
    assert out["its_pmg3"] < 3 * out["its_cheb"], out
