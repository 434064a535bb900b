"""dynamic instruction count of the SSA_PROP_ELEMENTS chain, stage by stage: ssa_propagate_f64 on 2^20 catalogue states under
rocprofv3 --pmc SQ_INSTS_VALU, with the diagnostic builds of build_ablate/el/ (SSA_EL_CUT = 1..4 cut the chain behind rv2coe /
the initial anomaly / the Kepler solve / the final true anomaly).  LIB selects the build."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import _build, _lib
if os.environ.get("LIB"):
    _build.LIB = os.path.join(ROOT, os.environ["LIB"])
from ssa_gym_amd import device
pb = bench.build_problem(20000, seed=100)
n = 1 << 20
x = torch.as_tensor(np.tile(pb["x"], (n // 20000 + 1, 1))[:n]).cuda()
for _ in range(5):
    device.propagate(x, 20.0, propagator=_lib.PROP_ELEMENTS)
torch.cuda.synchronize()
