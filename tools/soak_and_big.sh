mkdir -p gpurun_out/s2
TLFEA_BENCH_VERBOSE=1 python bench.py --no-cpu-baseline --steps 150 --warmup 3 > gpurun_out/s2/soak.json 2> gpurun_out/s2/soak.err; echo "soak rc=$?"
python - <<'PY'
import json,re
j=json.loads(open("gpurun_out/s2/soak.json").read().strip().splitlines()[-1])
print("soak: value %.3e ms %.3f" % (j["value"], j["ms_per_step"]))
t=open("gpurun_out/s2/soak.err").read()
m=re.search(r"CG iterations: \[(.*?)\]", t)
its=[int(v) for v in m.group(1).split(",")]
ms=[float(v) for v in re.search(r"per-iteration ms \(warm-up first\): \[(.*?)\]", t).group(1).split(",")]
print("CG iterations min/max", min(its), max(its), " per-iteration ms min/median/max", min(ms), sorted(ms)[len(ms)//2], max(ms))
PY
timeout -k 10 400 python3 tools/big_run.py 150 100 50 > gpurun_out/s2/big.txt 2>&1; tail -4 gpurun_out/s2/big.txt
