"""ssa_ladder_probe_f64 of a saved tile (the npz of ladder_ab.py) under the library LIB: which rung the build's fused ladder picks for
each of the tile's four matrices, next to which rungs factorise in the kernel's arithmetic."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ssa_gym_amd import host, _lib, _build, device
if os.environ.get("LIB"):
    _build.LIB = os.path.join(ROOT, os.environ["LIB"])
d = np.load(os.environ["CASE"])
_, _, scale = host.merwe_weights(1e-4, 2.0, -3)
P = torch.as_tensor(d["P_tile"]).cuda().contiguous()
rung, mask, U = device.ladder_probe(P, scale)
seq = device.robust_cholesky((scale * P).contiguous())[1]
print("%s: fused rung %s  sequential register ladder %s  status of the tile %s" % (os.environ.get("LIB", "in-tree"), rung.cpu().numpy().tolist(),
                                                                                 seq.cpu().numpy().tolist(), d["status_tile"].tolist()))
for k, m in enumerate(mask.cpu().numpy()):
    print("   object %d: plain %d  rungs 0..15 %s" % (int(d["obj"]) // 4 * 4 + k, (int(m) >> 16) & 1, format(int(m) & 0xffff, '016b')[::-1]))
Un = U.cpu().numpy()
if os.environ.get("UOUT"):
    np.save(os.environ["UOUT"], Un)
if os.environ.get("UREF"):
    R = np.load(os.environ["UREF"])
    same = np.array_equal(Un.view(np.int64), R.view(np.int64))
    print("   factor rows bit-identical to the other build's: %s" % same)
    if not same:
        k = np.argwhere(Un.view(np.int64) != R.view(np.int64))
        print("   differing entries (object, row, col):", k[:12].tolist())
        j = int(k[0][0])
        np.set_printoptions(precision=17, linewidth=220)
        print("   this build:\n", Un[j], "\n   other build:\n", R[j])
        for nm, F in (("this", Un[j]), ("other", R[j])):
            M = F.T @ F - scale * d["P_tile"][j]
            print("   %s: U^T U - scale P  diagonal = %s" % (nm, np.diag(M)))
