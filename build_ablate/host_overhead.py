import os, sys, time, numpy as np, torch, torch.distributed as dist
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
sys.argv=['bench.py']
import bench
from ssa_gym_amd import host, engine, parallel
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
m=20000
pb=bench.build_problem(m, seed=100)
consts=host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi/2, pb["obs_lla"])
z=torch.zeros((1,480,m,3),dtype=torch.float64,device='cuda')
eng=engine.HotPathEngine(consts,m,1,pb["trans"],z,history=2)
eng.load_state(0,pb["x_true"],pb["x"],np.broadcast_to(pb["P0"],(m,6,6)))
local=parallel.HipLocalStepper(eng,consts); local.load_schedule(np.arange(4000)%m)
send=torch.zeros(4*m+8,dtype=torch.float64,device='cuda'); recv=torch.zeros(4*m+8,dtype=torch.float64,device='cuda')
def t(f,n=300):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    return (t1-t0)/n*1e6, (t2-t0)/n*1e6
print('local.step host/total us', t(lambda: local.step(-1, send[:4*m], send[4*m:])))
print('all_gather sync host/total us', t(lambda: dist.all_gather_into_tensor(recv, send)))
ws=[]
def ag():
    w=dist.all_gather_into_tensor(recv, send, async_op=True); ws.append(w)
print('all_gather async host/total us', t(ag)); [w.wait() for w in ws]
s=torch.cuda.current_stream().cuda_stream
print('ctypes env_step only host/total', t(lambda: eng._lib.ssa_env_step_f64(eng._cref, eng._pref, s)))
dist.destroy_process_group()
