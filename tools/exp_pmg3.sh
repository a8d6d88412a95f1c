#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/pmg3
run() { name=$1; cfg=$2; st=$3; shift 3
  env "$@" python bench.py --no-cpu-baseline --config $cfg --steps $st --warmup 2 > gpurun_out/pmg3/$name.json 2> gpurun_out/pmg3/$name.err || { tail -5 gpurun_out/pmg3/$name.err; return; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/pmg3/$name.json").read().strip().splitlines()[-1])
print("$name", "value %.3e ms %.2f its %s"%(j["value"],j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"]))
PY
}
run C_kc32_ks5 C 3 TLFEA_PMG_KC=32 TLFEA_PMG_KAPPA_C=1600 TLFEA_PMG_KAPPA_S=5
run C_kc48_ks5 C 3 TLFEA_PMG_KC=48 TLFEA_PMG_KAPPA_C=3200 TLFEA_PMG_KAPPA_S=5
run C_kc64_ks5 C 3 TLFEA_PMG_KC=64 TLFEA_PMG_KAPPA_C=6400 TLFEA_PMG_KAPPA_S=5
run C_kc48_ks4 C 3 TLFEA_PMG_KC=48 TLFEA_PMG_KAPPA_C=3200 TLFEA_PMG_KAPPA_S=4
run C_kc48_ks3 C 3 TLFEA_PMG_KC=48 TLFEA_PMG_KAPPA_C=3200 TLFEA_PMG_KAPPA_S=3
run B_kc12_ks5 B 30 TLFEA_PMG_KC=12 TLFEA_PMG_KAPPA_C=200 TLFEA_PMG_KAPPA_S=5
run B_kc16_k800 B 30 TLFEA_PMG_KC=16 TLFEA_PMG_KAPPA_C=800
run B_kc10_k150 B 30 TLFEA_PMG_KC=10 TLFEA_PMG_KAPPA_C=150
