#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/pmg
for c in B C; do for p in 1 2; do
  st=30; [ $c = C ] && st=3
  python bench.py --no-cpu-baseline --config $c --steps $st --warmup 2 --precond $p > gpurun_out/pmg/${c}_p$p.json 2> gpurun_out/pmg/${c}_p$p.err || { tail -20 gpurun_out/pmg/${c}_p$p.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/pmg/${c}_p$p.json").read().strip().splitlines()[-1])
print("$c precond $p", "value %.3e ms %.2f its %s"%(j["value"],j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"]), {k:(v["avg_us"],v["frac"]) for k,v in j["roofline_all"].items() if k in ("spmv","cheb_step")}, j["stage_ms_per_step"]["pcg"])
PY
done; done
