"""Exploration: a dead aircraft comes to rest on the ground (DF_AT_REST), then a second dead aircraft slides into it. Device vs oracle."""
import os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from oracle import oracle as O
from tests.test_gpu_dogfight import _engine, _set_spawn, GOLD

g = np.load(os.path.join(GOLD, "env_dogfight_crash.npz"))
eng, A = _engine(1, "inject", max_duration_seconds=30.0)
_set_spawn(eng, g["start_pos"], g["start_orn"])
eng.env_reset(xi_reset=torch.zeros(g["reset_xi"].shape, dtype=torch.float32, device="cuda:0").contiguous())
W = O.OracleDogfight(g["start_pos"], g["start_orn"], noise_mode=O.NOISE_OFF, max_duration_seconds=30.0)
W.reset()
n_xi = g["xi"][0].shape[0]

def place(i, p, q, v, w, dead=True):
    s = eng.state
    s[0, i, :3] = torch.tensor(p, dtype=torch.float32, device="cuda:0"); s[1, i, :] = torch.tensor(q, dtype=torch.float32, device="cuda:0")
    s[2, i, :3] = torch.tensor(v, dtype=torch.float32, device="cuda:0"); s[2, i, 3] = float(w[0]); s[3, i, 0] = float(w[1]); s[3, i, 1] = float(w[2])
    L = W.Ls[i]
    for k in range(3): L.p[k] = float(np.float32(p[k])); L.v[k] = float(np.float32(v[k])); L.w[k] = float(np.float32(w[k]))
    for k in range(4): L.q[k] = float(np.float32(q[k]))
    if dead:
        s[6, i, 0] = 0.0
        fl = s[6, :, 3].view(torch.int32); fl[i] = int(fl[i]) & ~1
        W.D.alive[i] = 0; W.D.health[i] = 0.0

place(0, [10.0, 0.0, 0.4], [0, 0, 0, 1], [0, 0, 0], [0, 0, 0])
rest_at = None
for k in range(int(os.environ.get("STEPS", "200"))):
    eng.env_step(torch.zeros(A, 4, device="cuda:0"), xi=torch.zeros(n_xi, A, device="cuda:0"))
    W.step(np.zeros((A, 4)))
    f0 = int(eng.state[6, 0, 3].view(torch.int32))
    pd = eng.state[0, 0, :3].cpu().numpy(); po = np.array(W.Ls[0].p[:])
    vd = eng.state[2, 0, :3].cpu().numpy()
    if k % 10 == 0 or (f0 & 8192): print(k, "flags", hex(f0), "dev p", pd, "v", vd, "orc p", po, "inactive", W.D.inactive[0])
    if f0 & 8192:
        rest_at = k; break
print("rest_at", rest_at)
if rest_at is not None:
    p0 = eng.state[0, 0, :3].cpu().numpy().astype(np.float64)
    q0 = eng.state[1, 0, :].cpu().numpy()
    print("wreck rests at", p0, "q", q0, "oracle q", np.array(W.Ls[0].q[:]), "oracle v", np.array(W.Ls[0].v[:]), "w", np.array(W.Ls[0].w[:]))
    dx = float(os.environ.get("DX", "3.0")); dz = float(os.environ.get("DZ", "0.2")); sp = float(os.environ.get("SPEED", "8.0"))
    place(1, [p0[0] - dx, p0[1], p0[2] + dz], [0, 0, 0, 1], [sp, 0, 0], [0, 0, 0])
    for k in range(30):
        eng.env_step(torch.zeros(A, 4, device="cuda:0"), xi=torch.zeros(n_xi, A, device="cuda:0"))
        W.step(np.zeros((A, 4)))
        f0 = int(eng.state[6, 0, 3].view(torch.int32))
        pd = eng.state[0, :2, :3].cpu().numpy().astype(np.float64); po = np.array([W.Ls[i].p[:] for i in range(2)])
        print(k, "flags", hex(f0), "wreck dev", pd[0], "orc", po[0], "| mover dev", pd[1], "orc", po[1], "| err wreck %.2e mover %.2e" % (np.abs(pd[0] - po[0]).max(), np.abs(pd[1] - po[1]).max()),
              "peer_contact orc", [W.Ls[i].peer_contact for i in range(2)])
