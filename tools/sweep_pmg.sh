#!/bin/bash
# parameter sweep of the p-multigrid cycle (coarse polynomial degree / interval, smoother interval) on configs B and C
set -o pipefail
mkdir -p gpurun_out/pmg2
run() { # name config steps env...
  name=$1; cfg=$2; st=$3; shift 3
  env "$@" python bench.py --no-cpu-baseline --config $cfg --steps $st --warmup 2 > gpurun_out/pmg2/$name.json 2> gpurun_out/pmg2/$name.err || { tail -5 gpurun_out/pmg2/$name.err; return; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/pmg2/$name.json").read().strip().splitlines()[-1])
print("$name", "value %.3e ms %.2f its %s"%(j["value"],j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"]))
PY
}
run C_kc8_k100 C 3 TLFEA_PMG_KC=8 TLFEA_PMG_KAPPA_C=100
run C_kc12_k200 C 3 TLFEA_PMG_KC=12 TLFEA_PMG_KAPPA_C=200
run C_kc24_k1600 C 3 TLFEA_PMG_KC=24 TLFEA_PMG_KAPPA_C=1600
run C_kc32_k1600 C 3 TLFEA_PMG_KC=32 TLFEA_PMG_KAPPA_C=1600
run C_ks5 C 3 TLFEA_PMG_KAPPA_S=5
run C_ks12 C 3 TLFEA_PMG_KAPPA_S=12
run B_kc8_k100 B 30 TLFEA_PMG_KC=8 TLFEA_PMG_KAPPA_C=100
run B_kc12_k200 B 30 TLFEA_PMG_KC=12 TLFEA_PMG_KAPPA_C=200
run B_kc24_k1600 B 30 TLFEA_PMG_KC=24 TLFEA_PMG_KAPPA_C=1600
run B_ks5 B 30 TLFEA_PMG_KAPPA_S=5
run B_ks12 B 30 TLFEA_PMG_KAPPA_S=12
