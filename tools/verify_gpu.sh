#!/bin/bash
# full GPU test suite + config B / C bench lines in one gpurun call (gpurun -- bash tools/verify_gpu.sh)
set -o pipefail
mkdir -p gpurun_out/verify
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/verify/pytest.log 2>&1 || { tail -40 gpurun_out/verify/pytest.log; exit 1; }
tail -3 gpurun_out/verify/pytest.log
python bench.py --no-cpu-baseline > gpurun_out/verify/B.json 2> gpurun_out/verify/B.err || { tail -20 gpurun_out/verify/B.err; exit 1; }
python bench.py --no-cpu-baseline --config C --steps 3 --warmup 1 > gpurun_out/verify/C.json 2> gpurun_out/verify/C.err || { tail -20 gpurun_out/verify/C.err; exit 1; }
python - <<'PY'
import json
for c in "BC":
    j=json.loads(open("gpurun_out/verify/%s.json"%c).read().strip().splitlines()[-1])
    print(c, "value %.3e ms %.2f its %s elem %s"%(j["value"],j["ms_per_step"],j["config"]["pcg_outer_iters_per_step"], j["element_stage"]["ms_per_step"]))
    print({k:(v["avg_us"],v["frac"]) for k,v in j["roofline_all"].items()})
PY
