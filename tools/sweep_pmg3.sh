#!/bin/bash
# sweep of the third level's parameters on config C (gpurun -- bash tools/sweep_pmg3.sh)
mkdir -p gpurun_out/sweep3
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --config C --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'], j['config']['pcg_outer_iters_per_step'])"; }
export TLFEA_PMG_LEVELS=3
for ks in 16 30 60; do for kc in 12 20; do
  TLFEA_PMG_KC3=$kc TLFEA_PMG_KAPPA_S2=$ks run "kc3=$kc ks2=$ks"
done; done
TLFEA_PMG_KC3=12 TLFEA_PMG_KAPPA_S2=30 TLFEA_PMG_KAPPA_S=12 run "kc3=12 ks2=30 ks=12"
