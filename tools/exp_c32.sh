#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c32
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "preconditioner or linear_solve or newton or full_size" > gpurun_out/c32/pytest.log 2>&1 || { tail -30 gpurun_out/c32/pytest.log; exit 1; }
tail -2 gpurun_out/c32/pytest.log
for c in B C D; do
python bench.py --no-cpu-baseline --config $c --steps 3 --warmup 1 --max-pcg 300 > gpurun_out/c32/$c.json 2> gpurun_out/c32/$c.err || { tail -20 gpurun_out/c32/$c.err; exit 1; }
python - <<PY
import json
j=json.loads(open("gpurun_out/c32/$c.json").read().strip().splitlines()[-1])
print("$c", "value %.3e ms %.2f its %s"%(j["value"],j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"]), {k:(v["avg_us"],v["frac"]) for k,v in j["roofline_all"].items() if k in ("spmv","cheb_step")})
PY
done
