mkdir -p gpurun_out/s2
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s2/all.log 2>&1; echo "all rc=$?"; tail -3 gpurun_out/s2/all.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python3 bench.py > gpurun_out/s2/r02_bench_configC.json 2> gpurun_out/s2/bench_C.err; echo "C rc=$?"; cut -c1-160 gpurun_out/s2/r02_bench_configC.json
bash tools/pmc_quick.sh C pmcA2 > gpurun_out/s2/pmcq.log 2>&1; echo "pmc rc=$?"; grep assemble_affine gpurun_out/s2/pmcq.log | head -30
