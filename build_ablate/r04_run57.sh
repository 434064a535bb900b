#!/bin/bash
set -o pipefail
python3 build_ablate/first_block_probe.py 2>&1 | grep -v amdgpu.ids
