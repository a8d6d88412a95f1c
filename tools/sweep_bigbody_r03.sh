#!/bin/bash
# bodies larger than config C on ONE GPU (what the weak-scaling series turns into per rank): third-level resolution
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-sweepBig}
mkdir -p $O
run() { name=$1; cfg=$2; shift; shift
  env "$@" python bench.py --no-cpu-baseline --config $cfg --steps 9 --warmup 3 --prewarm-s 0 > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; }
  if grep -q "Memory access fault" $O/$name.err; then echo "GPU fault in $name: stopping"; exit 9; fi
  [ -s $O/$name.json ] || return
  python - <<PY
import json
j=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
ra=j["roofline_all"]
print("$name", "ms %.2f its %s rel %.2e"%(j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"],j["config"]["last_solve_rel_residual"]), j["config"]["preconditioner"][60:200], flush=True)
PY
}
run C2_default C2 X=1
run C2_bins24k C2 TLFEA_PMG_BINS_MAX=24000
run C2_bins48k C2 TLFEA_PMG_BINS_MAX=48000
run C4_default C4 X=1
run C4_bins24k C4 TLFEA_PMG_BINS_MAX=24000
run C4_bins48k C4 TLFEA_PMG_BINS_MAX=48000
