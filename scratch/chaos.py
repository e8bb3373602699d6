import numpy as np, sys
sys.path.insert(0,'.')
from oracle import oracle as O
def run(env, n, steps, low, high, eps, seed=17, gentle=False):
    A = O.OracleBatch(O.make_params(env, noise_mode=O.NOISE_PHILOX, seed=seed), n)
    B = O.OracleBatch(O.make_params(env, noise_mode=O.NOISE_PHILOX, seed=seed), n)
    A.reset(); B.reset()
    rng = np.random.default_rng(0)
    for i in range(n):
        for k in range(3):
            B.lanes[i].v[k] *= (1 + eps*rng.standard_normal())
            B.lanes[i].p[k] *= (1 + eps*rng.standard_normal())
    ok = np.ones(n,bool)
    rng = np.random.default_rng(seed+1)
    for s in range(steps):
        a = rng.uniform(low, high, size=(n,4)).astype(np.float32)
        if gentle: a = np.concatenate([rng.uniform(-0.3,0.3,size=(n,3)), rng.uniform(-0.2,0.8,size=(n,1))],axis=1).astype(np.float32)
        oa,ra,ta,tra,_ = A.step(a, autoreset=1); ob,rb,tb,trb,_ = B.step(a, autoreset=1)
        e = (np.abs(oa-ob)/np.maximum(1,np.abs(oa))).max(axis=1)
        ok &= (ta==tb)&(tra==trb)&(e<1e-4)
        if s%30==29: print(env, "step",s+1,"dropped",1-ok.mean(), "median err of ok lanes", np.median(e[ok]))
run("fixedwing_waypoints", 1024, 150, -np.ones(4), np.ones(4), 6e-8)
run("fixedwing_waypoints", 512, 300, -np.ones(4), np.ones(4), 6e-8, gentle=True)
run("hover", 1024, 150, np.array([-np.pi]*3+[0]), np.array([np.pi]*3+[0.8]), 6e-8)
