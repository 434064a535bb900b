#!/bin/bash
# round 4, GPU call 66: the -s report of the GPU suite on the last commit
set -o pipefail
mkdir -p gpurun_out/r4aq
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -s -x > gpurun_out/r4aq/pytest_s.log 2>&1; echo "pytest -s rc $?"; tail -2 gpurun_out/r4aq/pytest_s.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
