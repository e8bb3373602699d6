import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from oracle import oracle as O
from test_gpu_parity import _engine, _oracle, QUAD_LOW, QUAD_HIGH, relerr, obs_groups
n=4096; seed=23
eng=_engine("quadx","hover",n,noise="philox",autoreset="next_step",seed=seed)
orc=_oracle("hover",n,"philox",seed=seed)
rng=np.random.default_rng(seed+1)
G=obs_groups(eng.obs_dim,True,4,0,3)
og=eng.env_reset().cpu().numpy().astype(np.float64); orr=orc.reset()
print("reset err", relerr(og,orr,G).max())
def mixed(rng,n):
    a=rng.uniform(QUAD_LOW,QUAD_HIGH,size=(n,4)); g=np.concatenate([rng.uniform(-0.3,0.3,size=(n,3)),rng.uniform(0.33,0.40,size=(n,1))],axis=1); a[:n//2]=g[:n//2]; return a.astype(np.float32)
for k in range(1000):
    a=mixed(rng,n)
    o,r,t,tr=eng.env_step(torch.tensor(a,device="cuda:0"))
    ro,rr,rt,rtr,_=orc.step(a,autoreset=1)
    e=relerr(o.cpu().numpy().astype(np.float64),ro,G).max(axis=1)
    bad=np.nonzero((e>1e-3) & ~((rt|rtr) & (t.cpu().numpy()|tr.cpu().numpy())))[0]
    if len(bad):
        l=int(bad[0])
        st=eng.state.cpu()
        print("step",k,"bad lanes",bad[:8],len(bad),"e",e[l])
        print("dev obs",o[l].cpu().numpy()[:13]); print("orc obs",ro[l][:13])
        print("dev ints",st[6,l].view(torch.int32).tolist(),"g7",st[7,l].tolist(),st[7,l].view(torch.int32).tolist())
        L=orc.lanes[l]; print("orc step_count",L.step_count,"rng",L.rng_ctr,"key",L.reset_key,"term",L.terminated,L.truncated)
        break
else: print("clean")
