"""decodes the operand / result layouts of the fp64 MFMA forms and their issue cost (diagnostic, GPU box)"""
import ctypes as C, itertools, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
hip = C.CDLL("libamdhip64.so")
lib = C.CDLL(os.path.join(here, "libmfma_probe.so"))
class dim3(C.Structure): _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("z", C.c_uint)]
hip.hipLaunchKernel.argtypes = [C.c_void_p, dim3, dim3, C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]
def launch(name, args):
    fn = getattr(lib, name)      # host stub address == kernel handle for hipLaunchKernel
    arr = (C.c_void_p * len(args))(*[C.cast(C.pointer(a), C.c_void_p) for a in args])
    rc = hip.hipLaunchKernel(C.cast(fn, C.c_void_p), dim3(1, 1, 1), dim3(64, 1, 1), arr, 0, None)
    assert rc == 0, (name, rc)
    torch.cuda.synchronize()
rs = np.random.RandomState(0)
A = torch.as_tensor(rs.normal(size=64)).cuda(); B = torch.as_tensor(rs.normal(size=64)).cuda()
pA, pB = C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr())
a, b = A.cpu().numpy(), B.cpu().numpy()
D = torch.zeros(256, dtype=torch.float64, device="cuda")
launch("probe16", [pA, pB, C.c_void_p(D.data_ptr())])
d = D.cpu().numpy().reshape(64, 4)
lane = np.arange(64)
cands = {"i=l%16,k=l//16": (lane % 16, lane // 16), "i=l//4,k=l%4": (lane // 4, lane % 4)}
for na, (ia, ka) in cands.items():
    for nb, (jb, kb) in cands.items():
        Am = np.zeros((16, 4)); Bm = np.zeros((4, 16))
        Am[ia, ka] = a; Bm[kb, jb] = b
        Cm = Am @ Bm
        for nd, f in (("col=l%16,row=l//16+4r", lambda l, r: (l // 16 + 4 * r, l % 16)), ("col=l%16,row=4*(l//16)+r", lambda l, r: (4 * (l // 16) + r, l % 16))):
            ok = all(abs(Cm[f(l, r)] - d[l, r]) < 1e-12 for l in range(64) for r in range(4))
            if ok: print("16x16x4: A", na, "| B (j,k)", nb, "| D", nd, flush=True)
D4 = torch.zeros(64, dtype=torch.float64, device="cuda")
launch("probe4", [pA, pB, C.c_void_p(D4.data_ptr())])
d4 = D4.cpu().numpy()
found = False
fields = {"lo": lambda l: l % 4, "mid": lambda l: (l // 4) % 4, "hi": lambda l: l // 16}
for (bi, ii, ki) in itertools.permutations(fields, 3):
    for (bj, jj, kj) in itertools.permutations(fields, 3):
        for (bd, id_, jd) in itertools.permutations(fields, 3):
            Am = np.zeros((4, 4, 4)); Bm = np.zeros((4, 4, 4))
            Am[fields[bi](lane), fields[ii](lane), fields[ki](lane)] = a
            Bm[fields[bj](lane), fields[kj](lane), fields[jj](lane)] = b
            Cm = np.einsum('bik,bkj->bij', Am, Bm)
            got = Cm[fields[bd](lane), fields[id_](lane), fields[jd](lane)]
            if np.allclose(got, d4, atol=1e-12):
                print("4x4x4_4b: A block=%s i=%s k=%s | B block=%s j=%s k=%s | D block=%s i=%s j=%s" % (bi, ii, ki, bj, jj, kj, bd, id_, jd), flush=True); found = True
if not found:
    print("4x4x4: no simple layout matched; dump:", a.tolist(), b.tolist(), d4.tolist())
cyc = torch.zeros(2, dtype=torch.int64, device="cuda")
for name in ("time16", "time4", "timefma"):
    n = 4096
    for _ in range(2):
        launch(name, [pA, pB, C.c_void_p(D.data_ptr()), C.c_void_p(cyc.data_ptr()), C.c_int(n)])
    c = cyc.cpu().numpy()
    print("%s: dependent chain %.2f clock64 ticks/instr, 4 independent accumulators %.2f ticks/instr" % (name, c[0] / n, c[1] / n), flush=True)
