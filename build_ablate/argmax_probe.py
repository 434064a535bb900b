"""how long the arg-max head (ssa_masked_argmax_f64) and its neighbours take on their own (HIP events around 200 back-to-back launches)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from ssa_gym_amd import device, _lib
_lib.load()
for n in (2000, 20000, 160000):
    sc = torch.randn(n, dtype=torch.float64, device="cuda")
    mask = (torch.rand(n, device="cuda") > 0.5).to(torch.uint8)
    ws = device.masked_argmax_workspace(n, "cuda")
    for name, fn in (("masked_argmax ws (mask)", lambda: device.masked_argmax_action(sc, mask, ws)), ("masked_argmax (mask)", lambda: device.masked_argmax_action(sc, mask)), ("masked_argmax (no mask)", lambda: device.masked_argmax_action(sc)),
                     ("torch.argmax", lambda: torch.argmax(sc))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            fn()
        b.record(); torch.cuda.synchronize()
        print("n=%6d  %-24s %.2f us per launch (back to back)" % (n, name, a.elapsed_time(b) * 1e3 / 200), flush=True)
