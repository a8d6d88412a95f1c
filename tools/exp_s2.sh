mkdir -p gpurun_out/s2
python3 bench.py > gpurun_out/s2/r02_bench_configC.json 2> gpurun_out/s2/bench_C.err; echo "C rc=$?"; cut -c1-200 gpurun_out/s2/r02_bench_configC.json
python3 bench.py --config B > gpurun_out/s2/r02_bench_configB.json 2> gpurun_out/s2/bench_B.err; echo "B rc=$?"; cut -c1-200 gpurun_out/s2/r02_bench_configB.json
python3 bench.py --config D --no-cpu-baseline --steps 6 --warmup 2 --max-pcg 300 > gpurun_out/s2/r02_bench_configD.json 2> gpurun_out/s2/bench_D.err; echo "D rc=$?"; cut -c1-200 gpurun_out/s2/r02_bench_configD.json
