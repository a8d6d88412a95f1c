#!/bin/bash
# round-end verification on the GPU box: full GPU suite, smoke(), profiles (kernel trace, PMC, bench lines), soak run
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/final/smoke.log
bash tools/collect_profiles.sh > gpurun_out/final/collect.log 2>&1; echo "collect rc=$?"; tail -3 gpurun_out/final/collect.log | cut -c1-160
TLFEA_BENCH_VERBOSE=1 python bench.py --no-cpu-baseline --steps 150 --warmup 3 > gpurun_out/final/soak.json 2> gpurun_out/final/soak.err; echo "soak rc=$?"
python - <<'PY'
import json,re
j=json.loads(open("gpurun_out/final/soak.json").read().strip().splitlines()[-1])
print("soak: value %.3e ms %.3f" % (j["value"], j["ms_per_step"]))
t=open("gpurun_out/final/soak.err").read()
m=re.search(r"CG iterations: \[(.*?)\]", t)
its=[int(v) for v in m.group(1).split(",")]
ms=[float(v) for v in re.search(r"per-iteration ms \(warm-up first\): \[(.*?)\]", t).group(1).split(",")]
print("CG iterations min/max", min(its), max(its), " per-iteration ms min/median/max", min(ms), sorted(ms)[len(ms)//2], max(ms))
PY
