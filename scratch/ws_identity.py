"""scratch: Fixedwing WS vs single-wave variants must be bit-identical."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
def run(ws):
    os.environ["PF_WS"] = str(ws)
    from pyflyt_amd.gym_envs import make_vec
    env = make_vec("PyFlyt/Fixedwing-Waypoints-v4", 5000, seed=5, flatten=True)
    o = [env.reset(seed=5)[0].clone()]
    for k in range(200):
        ob, r, t, u, _ = env.step(env.sample_actions(k))
        o += [ob.clone(), r.clone(), t.clone(), u.clone()]
    st = env.engine.state.clone()
    env.close()
    return o, st
a, sa = run(0); b, sb = run(1)
print("identical:", all(torch.equal(x, y) for x, y in zip(a, b)) and torch.equal(sa, sb))
