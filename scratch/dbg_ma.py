import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyflyt_amd.pz_envs import MAQuadXHoverEnv
def run(disable):
    if disable: os.environ["PF_DISABLE_FAST"]="1"
    else: os.environ.pop("PF_DISABLE_FAST",None)
    env = MAQuadXHoverEnv(num_envs=64, seed=4, flight_dome_size=2.5, max_duration_seconds=1.0)
    env.reset(seed=4)
    e = env.engine
    rng = np.random.default_rng(0)
    out=[]
    for k in range(45):
        a = torch.tensor(np.concatenate([rng.uniform(-1,1,size=(256,3)), rng.uniform(0.2,0.7,size=(256,1))],1).astype(np.float32), device="cuda")
        o,r,t,u = e.env_step(a)
        out.append((o.clone(), r.clone(), t.clone(), u.clone(), e.state.clone()))
    return out
A=run(False); B=run(True)
for k,(x,y) in enumerate(zip(A,B)):
    dr=(x[1]-y[1]).abs(); do=(x[0]-y[0]).abs().max(1).values
    bad=(dr>1e-2).nonzero().flatten()
    if len(bad):
        i=int(bad[0]); print("step",k,"lane",i,"r fast",x[1][i].item(),"r gen",y[1][i].item(),"term",x[2][i].item(),y[2][i].item(),"trunc",x[3][i].item(),y[3][i].item())
        print(" p fast",x[4][0][i], " p gen", y[4][0][i]); print(" ints fast", x[4][6][i].view(torch.int32), "gen", y[4][6][i].view(torch.int32))
        break
    print(k, dr.max().item(), do.max().item())
