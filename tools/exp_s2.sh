mkdir -p gpurun_out/s2
timeout -k 10 500 python -m pytest tests/test_gpu_ancf.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/s2/ancf.log 2>&1; echo "ancf+parity rc=$?"; tail -3 gpurun_out/s2/ancf.log
timeout -k 10 240 python3 tools/prof_elem.py D 3 > gpurun_out/s2/profD.log 2>&1; tail -2 gpurun_out/s2/profD.log
