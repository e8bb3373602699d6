import numpy as np, torch, sys
sys.path.insert(0,'.')
from oracle import oracle as O
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
np.set_printoptions(linewidth=250, precision=4, suppress=True)
n=512; seed=19
P = build_params("fixedwing","waypoints",noise="philox",autoreset="next_step",seed=seed,goal_reach_distance=40.0)
eng = BatchEngine(P,n)
orc = O.OracleBatch(O.make_params("fixedwing_waypoints", noise_mode=O.NOISE_PHILOX, seed=seed, goal_reach_distance=40.0), n)
rng = np.random.default_rng(seed+1)
og = eng.env_reset().cpu().numpy(); orr = orc.reset()
L=478
print("reset", np.abs(og-orr).max())
for k in range(7):
    a = np.concatenate([rng.uniform(-0.3,0.3,size=(n,3)), rng.uniform(-0.2,0.8,size=(n,1))],axis=1).astype(np.float32)
    og, rg, tg, trg = eng.env_step(torch.tensor(a,device="cuda:0"))
    og=og.cpu().numpy(); ints = eng.ints().cpu().numpy()
    orr, rr, tr, trr, fin = orc.step(a, autoreset=1)
    print(k, "gpu nleft", ints[L], "flags", tg[L].item(), trg[L].item(), "orc nleft", orc.lanes[L].n_targets_left, tr[L], trr[L], "rew", rg[L].item(), rr[L])
    print("  gpu", og[L,23:]); print("  orc", orr[L,23:])
    print("  newdist gpu", eng.state[0,L,3].item(), "orc", orc.lanes[L].new_dist, orc.lanes[L].old_dist)
