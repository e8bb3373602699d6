import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_golden as T
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
name, vehicle, task, over = next(c for c in T.ENVS if c[0] == "env_quadx_waypoints_mode7")
g = T.load(name)
P = build_params(vehicle, task, noise="inject", autoreset="off", **over)
eng = BatchEngine(P, T.N, device=T.DEV)
print("groups", eng.state.shape[0], "specialised", eng.lib.pf_ctx_is_specialised(eng._ctx))
D = eng.obs_dim; nt = P.num_targets
G = T.obs_groups(D, bool(P.angle_repr), 4, nt, 4 if P.use_yaw_targets else 3)
resets = set(int(k) for k in g["reset_before"]); ri = 0
def do_reset():
    global ri
    ut = T.dev_cols(g["reset_u"][ri])
    obs = eng.env_reset(xi_reset=T.dev_cols(g["reset_xi"][ri]), u_targets=ut).double().cpu().numpy()
    print("reset", ri, "err", T.vec_err(obs, g["reset_obs"][ri], G)); ri += 1
do_reset()
for k in range(len(g["action"])):
    if k in resets: do_reset()
    a = torch.tensor(np.repeat(g["action"][k][None], T.N, axis=0), dtype=torch.float32, device=T.DEV).contiguous()
    obs, rew, term, trunc = eng.env_step(a, xi=T.dev_cols(g["xi"][k]))
    o = obs.double().cpu().numpy()
    e = T.vec_err(o, g["obs"][k], G)
    if k % 20 == 0 or k < 5 or e > 3e-4:
        d = np.abs(o[0] - g["obs"][k]); j = int(d.argmax())
        print(k, "%.2e" % e, "worst col", j, "dev %.7f ref %.7f" % (o[0][j], g["obs"][k][j]), "action", g["action"][k])
