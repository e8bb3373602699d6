"""debug: the mode-7 waypoint fixture replayed on the device (fp64-state instantiation) and on the oracle side by side; per step the
largest difference of each STATE component (device master = float32 word + remainder word)."""
import ctypes as C, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_golden as T
from oracle import oracle as O
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
name, vehicle, task, over = next(c for c in T.ENVS if c[0] == "env_quadx_waypoints_mode7")
g = T.load(name)
P = build_params(vehicle, task, noise="inject", autoreset="off", **over)
eng = BatchEngine(P, T.N, device=T.DEV)
OP = O.make_params("quadx_waypoints", noise_mode=O.NOISE_INJECT, flight_mode=7, goal_reach_distance=0.4)
lib = O.lib(); L = O.Lane()
dp = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))
resets = set(int(k) for k in g["reset_before"]); ri = 0
def dev_state():
    s = eng.state[:, 0, :].double().cpu().numpy()
    hi = lambda grp, w: s[grp, w]
    p = np.array([s[0,0]+s[16,0], s[0,1]+s[16,1], s[0,2]+s[16,2]])
    q = np.array([s[1,0]+s[16,3], s[1,1]+s[17,0], s[1,2]+s[17,1], s[1,3]+s[17,2]])
    v = np.array([s[2,0]+s[17,3], s[2,1]+s[18,0], s[2,2]+s[18,1]])
    w = np.array([s[2,3]+s[18,2], s[3,0]+s[18,3], s[3,1]+s[19,0]])
    thr = np.array([s[3,2]+s[19,1], s[3,3]+s[19,2], s[4,0]+s[19,3], s[4,1]+s[21,2]])
    rI = np.array([s[4,2]+s[20,0], s[4,3]+s[20,1], s[5,0]+s[20,2]]); rE = np.array([s[5,1]+s[20,3], s[5,2]+s[21,0], s[5,3]+s[21,1]])
    casc = np.concatenate([s[7]+s[22], s[8]+s[23], s[9]+s[24], s[10]+s[25], (s[11]+s[26])[:2]])
    return p, q, v, w, thr, rI, rE, casc
def orc_state():
    f = lambda n, k: np.array(list(getattr(L, n))[:k]) if not hasattr(getattr(L, n)[0], "__len__") else None
    p = np.array(L.p[:3]); q = np.array(L.q[:4]); v = np.array(L.v[:3]); w = np.array(L.w[:3]); thr = np.array(L.throttle[:4])
    I = np.array([list(r) for r in L.pid_I]); E = np.array([list(r) for r in L.pid_E]); zI = np.array(L.zpid_I[:2]); zE = np.array(L.zpid_E[:2])
    casc = np.concatenate([I[1], E[1], I[2][:2], E[2][:2], I[3][:2], E[3][:2], zI, zE])
    return p, q, v, w, thr, I[0], E[0], casc
def reset():
    global ri
    eng.env_reset(xi_reset=T.dev_cols(g["reset_xi"][ri]), u_targets=T.dev_cols(g["reset_u"][ri]))
    lib.orc_env_reset(C.byref(OP), C.byref(L), 0, dp(g["reset_xi"][ri]), dp(g["reset_u"][ri])); ri += 1
reset()
names = ["p", "q", "v", "w", "thr", "rateI", "rateE", "cascade"]
for k in range(len(g["action"])):
    if k in resets: reset()
    a = torch.tensor(np.repeat(g["action"][k][None], T.N, axis=0), dtype=torch.float32, device=T.DEV).contiguous()
    eng.env_step(a, xi=T.dev_cols(g["xi"][k]))
    lib.orc_env_step(C.byref(OP), C.byref(L), dp(g["action"][k].astype(np.float32) if os.environ.get("F32_ACTIONS") else g["action"][k]), dp((np.nan_to_num(g["xi"][k]).astype(np.float32)) if os.environ.get("F32_ACTIONS") else np.nan_to_num(g["xi"][k])))
    if k < 3 or k % 20 == 0 or 165 <= k <= 180:
        d = [float(np.abs(x - y).max()) for x, y in zip(dev_state(), orc_state())]
        print(k, " ".join(f"{n} {e:.1e}" for n, e in zip(names, d)))
