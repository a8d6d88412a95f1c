#!/bin/bash
# fp8 (e4m3) storage of the fine level's streamed copy inside the cycle (round 3 experiment): A/B at configs C, M2, B
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-sweepFp8}
mkdir -p $O
run() { name=$1; cfg=$2; shift; shift
  env "$@" python bench.py --no-cpu-baseline --config $cfg --steps 6 --warmup 2 > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; }
  if grep -q "Memory access fault" $O/$name.err; then echo "GPU fault in $name: stopping"; exit 9; fi
  [ -s $O/$name.json ] || return
  python - <<PY
import json
j=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
k={r["kernel"]:r for r in j.get("roofline_all",[])} if isinstance(j.get("roofline_all"),list) else {}
print("$name", "ms %.2f its %s rel %.2e"%(j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"],j["config"]["last_solve_rel_residual"]), flush=True)
PY
}
run C_fp16 C X=1
run C_fp8 C TLFEA_FINE_BITS=8
run C_fp8_ks3 C TLFEA_FINE_BITS=8 TLFEA_PMG_KS=3
run M2_fp16 M2 X=1
run M2_fp8 M2 TLFEA_FINE_BITS=8
run B_fp16 B X=1
run B_fp8 B TLFEA_FINE_BITS=8
