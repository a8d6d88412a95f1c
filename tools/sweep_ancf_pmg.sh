#!/bin/bash
# config D (256 000 ANCF-3443 shells): Chebyshev(24) polynomial against the two-level cycle on the position coefficients
# (pmg_build_ancf), smoother / coarse polynomial parameters.   tools/sweep_ancf_pmg.sh [tag] [config]
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-sweepD}; CFG=${2:-D}
O=gpurun_out/$TAG
mkdir -p $O
run() { # name precond env...
  name=$1; pre=$2; shift 2
  env "$@" python bench.py --no-cpu-baseline --config $CFG --steps 4 --warmup 2 --max-pcg 400 --precond $pre > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; return; }
  python - <<PY
import json
j=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name", "ms %.2f its %s rel %.2e pcg_ms %.2f"%(j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"],j["config"]["last_solve_rel_residual"],j["stage_ms_per_step"]["pcg"]), flush=True)
PY
}
run cheb24 1
run pmg_ks2 2 TLFEA_PMG_ANCF=1
run pmg_ks2_k16 2 TLFEA_PMG_ANCF=1 TLFEA_PMG_KAPPA_S=16
run pmg_ks3_k16 2 TLFEA_PMG_ANCF=1 TLFEA_PMG_KS=3 TLFEA_PMG_KAPPA_S=16
run pmg_ks3_k30 2 TLFEA_PMG_ANCF=1 TLFEA_PMG_KS=3 TLFEA_PMG_KAPPA_S=30
run pmg_ks4_k50 2 TLFEA_PMG_ANCF=1 TLFEA_PMG_KS=4 TLFEA_PMG_KAPPA_S=50
run pmg_ks2_kc16 2 TLFEA_PMG_ANCF=1 TLFEA_PMG_KC=16 TLFEA_PMG_KAPPA_C=400
run pmg_ks2_kc48 2 TLFEA_PMG_ANCF=1 TLFEA_PMG_KC=48 TLFEA_PMG_KAPPA_C=3500
