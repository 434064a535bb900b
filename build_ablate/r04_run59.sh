#!/bin/bash
# round 4, GPU call 59: the whole GPU suite and smoke() on the last commit
set -o pipefail
mkdir -p gpurun_out/r4aj
python -m pytest tests -m gpu -q -x > gpurun_out/r4aj/pytest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r4aj/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-250
python -m pytest tests -q -x -m "not gpu" > gpurun_out/r4aj/pytest_cpu.log 2>&1; echo "cpu pytest rc $?"; tail -1 gpurun_out/r4aj/pytest_cpu.log
