#!/bin/bash
# round 4, GPU call 21: the GPU suite with -s after the failed-capture fix, then without
set -u
R=$(pwd); OUT=$R/gpurun_out/r4u; mkdir -p $OUT
timeout -k 10 1100 python3 -X faulthandler -m pytest tests -m gpu -q -s -x > $OUT/pytest_s.log 2>&1; echo "pytest -s rc $?" | tee -a $OUT/summary.txt
grep -E "^\[|passed|failed|rror|what\(\)" $OUT/pytest_s.log | cut -c1-400 | tail -30
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -x > $OUT/pytest_q.log 2>&1; echo "pytest -q rc $?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_q.log
