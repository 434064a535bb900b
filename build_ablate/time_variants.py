"""times the fused step kernel of every ablation build (diagnostic; not part of the product)."""
import sys, os, glob, ctypes as C, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
sys.argv=['bench.py']
import bench
from ssa_gym_amd import _lib, _build, host, engine, parallel
m=int(os.environ.get('M','20000'))
FAST=os.environ.get('FAST','0')=='1'   # the bench's path: statistics by sharded atomics, folded by an extra wavefront of the next launch
_sh=torch.zeros((2,1,128,16),dtype=torch.int64,device='cuda')
def fast(p,k):
    if FAST:
        p.stat_shards=_sh[k&1].data_ptr(); p.stat_shards_prev=_sh[(k+1)&1].data_ptr(); p.stats_prev=p.stats; p.launch_mask=8
pb=bench.build_problem(m, seed=100)
res={}
ROUNDS=int(os.environ.get('ROUNDS','3'))   # interleaved rounds over all builds: box / clock drift hits every build alike
for rnd in range(ROUNDS):
  for path in sorted(glob.glob(os.path.join(ROOT,'build_ablate','*.so'))):
      _lib._lib=None; _build.LIB=path
      sig=dict(_lib.SIGNATURES)
      import ctypes
      probe=ctypes.CDLL(path)
      _lib.SIGNATURES={k:v for k,v in sig.items() if hasattr(probe,k)}   # older ablation builds lack the newest entry points
      _lib.ABI_VERSION=probe.ssa_abi_version()
      lib=_lib.load()
      _lib.SIGNATURES=sig
      for prop in os.environ.get('PROPS','fg,elements').split(','):
          consts=host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi/2, pb["obs_lla"], obs_type='aer', propagator=prop)
          z=torch.zeros((1,480,m,3),dtype=torch.float64,device='cuda')
          eng=engine.HotPathEngine(consts,m,1,pb["trans"],z,history=2)
          eng.load_state(0,pb["x_true"],pb["x"],np.broadcast_to(pb["P0"],(m,6,6)))
          snap=eng.snapshot(0)
          sched=torch.full((64,),-1,dtype=torch.int32,device='cuda'); sched[:]=torch.arange(64,dtype=torch.int32,device="cuda")*313 % m
          p=eng._p; s=torch.cuda.current_stream().cuda_stream
          best=1e9
          for rep in range(5):
              eng.restore(0,snap); torch.cuda.synchronize()
              e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
              e0.record()
              for k in range(100):
                  tick=k+1; p.time_offset=tick; sin,sout=(tick-1)%2,tick%2
                  p.x_true_in,p.x_true_out=eng._bx_t+sin*eng._sx,eng._bx_t+sout*eng._sx
                  p.x_in,p.x_out=eng._bx+sin*eng._sx,eng._bx+sout*eng._sx
                  p.P_in,p.P_out=eng._bP+sin*eng._sP,eng._bP+sout*eng._sP
                  p.obs=eng._bo; p.metrics=eng._bm; p.upd=eng._bu; p.stats=eng._bs; p.launch_mask=int(os.environ.get('MASK','1')); fast(p,k)
                  p.actions=sched.data_ptr()+4*(k%64)
                  lib.ssa_env_step_f64(eng._cref,eng._pref,s)
              e1.record(); torch.cuda.synchronize(); best=min(best,e0.elapsed_time(e1)/100)
          # late-episode states (diverged predict-only filters: hyperbolic / near-parabolic sigma points): steps 330-430
          eng.restore(0,snap); torch.cuda.synchronize()
          def run(k0,n):
              for k in range(k0,k0+n):
                  tick=k+1; p.time_offset=tick; sin,sout=(tick-1)%2,tick%2
                  p.x_true_in,p.x_true_out=eng._bx_t+sin*eng._sx,eng._bx_t+sout*eng._sx
                  p.x_in,p.x_out=eng._bx+sin*eng._sx,eng._bx+sout*eng._sx
                  p.P_in,p.P_out=eng._bP+sin*eng._sP,eng._bP+sout*eng._sP
                  p.obs=eng._bo; p.metrics=eng._bm; p.upd=eng._bu; p.stats=eng._bs; p.launch_mask=int(os.environ.get('MASK','1')); fast(p,k)
                  p.actions=sched.data_ptr()+4*(k%64)
                  lib.ssa_env_step_f64(eng._cref,eng._pref,s)
          run(0,330); torch.cuda.synchronize()
          e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
          e0.record(); run(330,100); e1.record(); torch.cuda.synchronize(); late=e0.elapsed_time(e1)/100
          # stats kernel
          res.setdefault((os.path.basename(path), prop), []).append((best*1e3, late*1e3, int((eng.status!=0).sum().item())))
          del eng, z; torch.cuda.empty_cache()
for (name, prop), v in res.items():
    a=np.array(v)
    print(name, prop, 'step_kernel us min %.2f med %.2f'%(a[:,0].min(), np.median(a[:,0])), 'late-episode us min %.2f med %.2f'%(a[:,1].min(), np.median(a[:,1])), 'failed', int(a[-1,2]), flush=True)
