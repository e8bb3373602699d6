"""profiling tool: seed fuzz of the env parity harness (tests/test_gpu_parity.run_env_parity) -- rare paths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import test_gpu_parity as T
worst_all = {}
for seed in range(100, 112):
    for veh, task, name, lo, hi in (("quadx", "hover", "hover", T.QUAD_LOW, T.QUAD_HIGH),
                                    ("quadx", "waypoints", "quadx_waypoints", T.QUAD_LOW, T.QUAD_HIGH),
                                    ("fixedwing", "waypoints", "fixedwing_waypoints", T.FW_LOW, T.FW_HIGH)):
        for ar in ("next_step", "same_step"):
            w, nd = T.run_env_parity(veh, task, name, 2048, 200, "philox", ar, lo, hi, seed=seed)
            worst_all[(name, ar)] = max(worst_all.get((name, ar), 0.0), w)
print({k: f"{v:.2e}" for k, v in worst_all.items()})
