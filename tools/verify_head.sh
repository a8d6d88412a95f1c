#!/bin/bash
# full GPU suite + smoke() on the GPU box (round-end check of HEAD)
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
