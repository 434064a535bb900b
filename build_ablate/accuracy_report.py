"""prints the three-way accuracy statistics quoted in DESIGN.md section 4 (GPU vs fp64 oracle vs 80-bit oracle)."""
import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT,'tests'), os.path.join(ROOT,'oracle')): sys.path.insert(0,p)
import torch, oracle as orc
import test_hip_step as T
class H: pass
from ssa_gym_amd import _lib, device, host, engine
h=H(); h.torch,h.lib,h.dev,h.host,h.engine = torch,_lib,device,host,engine
o=orc.Oracle(); ol=orc.Oracle(True)
def q(v): return ' '.join('%.2e'%np.quantile(v,p) for p in (0.5,0.9,0.99,1.0))
m=2000
xt,x,P,g = T.make_batch(m, seed=1)
for alpha in (1e-3,1e-4):
  f64=T.run_oracle(o,xt,x,P,g,-1,1,alpha,z_noise3=np.zeros(3)); ld=T.run_oracle(ol,xt,x,P,g,-1,1,alpha,centred=True,z_noise3=np.zeros(3))
  rp,rv,rP=T.errs(f64,ld)
  print('alpha %g  reference arithmetic vs exact: pos'%alpha,q(rp),'| P',q(rP))
  for prop in ('fg','elements'):
    gpu=T.run_gpu(h,xt,x,P,g,[-1],1,alpha,propagator=prop)
    gp,gv,gP=T.errs(gpu,ld); ep,ev,eP=T.errs(gpu,f64)
    print('   GPU(%s) vs exact: pos'%prop,q(gp),'| P',q(gP),'|| vs reference-order: pos',q(ep))
