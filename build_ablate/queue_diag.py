import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
sys.argv=['bench.py']
import bench
from ssa_gym_amd import host, engine, parallel
m=20000
pb=bench.build_problem(m, seed=100)
for prop in ('fg',):
    consts=host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi/2, pb["obs_lla"], obs_type='aer', propagator=prop)
    z=torch.zeros((1,480,m,3),dtype=torch.float64,device='cuda')
    eng=engine.HotPathEngine(consts,m,1,pb["trans"],z,history=2)
    eng.load_state(0,pb["x_true"],pb["x"],np.broadcast_to(pb["P0"],(m,6,6)))
    local=parallel.HipLocalStepper(eng,consts); local.load_schedule(np.arange(600)%m)
    p=eng._p; s=torch.cuda.current_stream().cuda_stream
    mu=398600441800000.0
    for k in range(1,480):
        tick=k; p.time_offset=tick; sin,sout=(tick-1)%2,tick%2
        p.x_true_in,p.x_true_out=eng._bx_t+sin*eng._sx,eng._bx_t+sout*eng._sx
        p.x_in,p.x_out=eng._bx+sin*eng._sx,eng._bx+sout*eng._sx
        p.P_in,p.P_out=eng._bP+sin*eng._sP,eng._bP+sout*eng._sP
        p.obs=eng._bo+sout*eng._so; p.metrics=eng._bm+sout*eng._sm; p.upd=eng._bu+sout*eng._su; p.stats=eng._bs+sout*eng._ss
        p.actions=local._sched.data_ptr()+4*(k%600)
        report = k in (1,100,200,250,300,350,400,450,479)
        if report:
            x=eng.x_filter[sin].cpu().numpy(); P=eng.P_filter[sin].cpu().numpy(); st=eng.status.cpu().numpy()
        e0,e1,e2=[torch.cuda.Event(enable_timing=True) for _ in range(3)]
        p.launch_mask=1; e0.record(); eng._lib.ssa_env_step_f64(eng._cref,eng._pref,s); e1.record()
        torch.cuda.synchronize(); nq=int(eng.work[0].item())
        lst=eng.work[4:4+nq].cpu().numpy() if report else None
        p.launch_mask=6; eng._lib.ssa_env_step_f64(eng._cref,eng._pref,s); e2.record(); torch.cuda.synchronize()
        if report:
            r=np.linalg.norm(x[:,:3],axis=1); v=np.linalg.norm(x[:,3:],axis=1); en=0.5*v*v-mu/r
            unb=en>=0
            ok=st==0
            # PD check of scale*P for queued objects
            q=np.zeros(m,bool); q[lst]=True
            npd=0
            for j in lst[:2000]:
                try: np.linalg.cholesky(P[j])
                except np.linalg.LinAlgError: npd+=1
            print('step',k,'queue',nq,'unbound(all)',int(unb[ok].sum()),'queued&unbound',int((q&unb).sum()),'queued nonPD (of first 2000)',npd,'failed',int((~ok).sum()),'fast us %.1f post+final us %.1f'%(e0.elapsed_time(e1)*1e3,e1.elapsed_time(e2)*1e3), flush=True)
    p.launch_mask=0
