"""Workload for the rocprofv3 PMC passes (HBM traffic of the step kernel + calibration kernels).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 profiles/pmc_workload.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 profiles/pmc_workload.py

Calibration (MI355X_MICROARCH.md "HBM": FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950
and is uncalibrated for other widths): two kernels with KNOWN byte counts in the same access style as
the step kernel (8-byte lanes over array-of-structures rows) run in the same process:
  * ssa::propagate_kernel on 2^20 states  : reads 48 B, writes 48 B per state
  * ssa::observe_kernel   on 2^20 objects : reads 48+48+288 B (every line of P is touched), writes 96+32 B
"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, parallel, device

m = int(os.environ.get('M', '20000'))            # M / PROP: the other configurations of profiles/traffic.json (collect_more.sh)
pb = bench.build_problem(m, seed=100)
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer', propagator=os.environ.get('PROP', 'hybrid'))
z = torch.zeros((1, 480, m, 3), dtype=torch.float64, device='cuda')
eng = engine.HotPathEngine(consts, m, 1, pb["trans"], z, history=2)
if os.environ.get('LAYOUT', '1') == '1':      # the env's default storage layout, as bench.py's `value` (LAYOUT=0: the caller's order)
    from ssa_gym_amd.catalogue import regime_order
    eng.set_layout(regime_order(pb["x_true"]))
eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
local = parallel.HipLocalStepper(eng, consts, fast_stats=True)
local.load_schedule(np.arange(400) % m)
for k in range(200):          # healthy part of an episode
    local.step(-1)
torch.cuda.synchronize()
n = 1 << 20
x = torch.randn((n, 6), dtype=torch.float64, device='cuda') * 1e3 + torch.tensor([7e6, 0, 0, 0, 7.5e3, 0], dtype=torch.float64, device='cuda')
P = torch.randn((n, 6, 6), dtype=torch.float64, device='cuda')
for _ in range(3):
    y = device.propagate(x, 20.0)
    device.observe(x, y, P)
torch.cuda.synchronize()
