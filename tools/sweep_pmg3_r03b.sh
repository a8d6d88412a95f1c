#!/bin/bash
# second sweep of the three-level cycle at config C: more vertex-level smoother terms, level-3 degree
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-sweepP3b}
mkdir -p $O
run() { name=$1; shift
  env TLFEA_PMG_LEVELS=3 "$@" python bench.py --no-cpu-baseline --config C --steps 6 --warmup 2 > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; return; }
  python - <<PY
import json
j=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name", "ms %.2f its %s rel %.2e"%(j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"],j["config"]["last_solve_rel_residual"]), flush=True)
PY
}
run ks6_k40 TLFEA_PMG_KS2=6 TLFEA_PMG_KAPPA_S2=40
run ks6_k90 TLFEA_PMG_KS2=6 TLFEA_PMG_KAPPA_S2=90
run ks8_k100 TLFEA_PMG_KS2=8 TLFEA_PMG_KAPPA_S2=100
run ks8_k150 TLFEA_PMG_KS2=8 TLFEA_PMG_KAPPA_S2=150
run ks10_k150 TLFEA_PMG_KS2=10 TLFEA_PMG_KAPPA_S2=150
run ks12_k220 TLFEA_PMG_KS2=12 TLFEA_PMG_KAPPA_S2=220
run ks6_k60_kc16 TLFEA_PMG_KS2=6 TLFEA_PMG_KAPPA_S2=60 TLFEA_PMG_KC3=16 TLFEA_PMG_KAPPA_C3=400
run ks6_k60_kc32 TLFEA_PMG_KS2=6 TLFEA_PMG_KAPPA_S2=60 TLFEA_PMG_KC3=32 TLFEA_PMG_KAPPA_C3=1500
run ks8_k100_kc16 TLFEA_PMG_KS2=8 TLFEA_PMG_KAPPA_S2=100 TLFEA_PMG_KC3=16 TLFEA_PMG_KAPPA_C3=400
run ks8_k100_fks3 TLFEA_PMG_KS2=8 TLFEA_PMG_KAPPA_S2=100 TLFEA_PMG_KS=3 TLFEA_PMG_KAPPA_S=16
