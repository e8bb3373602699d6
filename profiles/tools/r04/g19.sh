#!/bin/bash
# several lane slices of one GPU's batch on their own streams (one engine + graph each): do the launches' ramps and tails overlap?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04h
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04h/slices_per_gpu.txt
import sys, time, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
for env in ("hover", "quadx_waypoints", "fixedwing_waypoints"):
    for S in (1, 2, 4, 8):
        N, G, REP = 65536, 100, 10
        n = N // S
        engs = [bench.make_engine(env, n, dev, lane_offset=k * n, noise="philox") for k in range(S)]
        rings = []
        for e in engs:
            r = [torch.empty(n, 4, device=dev) for _ in range(G)]
            for i, a in enumerate(r):
                e.sample_actions(a, i)
            rings.append(r)
            e.env_reset()
            bench.preroll(e, r, 400, 1 << 20)
            e.env_step(r[0])
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        graphs = []
        for e, r, st in zip(engs, rings, streams):
            with torch.cuda.stream(st):
                st.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    for i in range(G):
                        e.env_step(r[i])
                g.replay()
                st.synchronize()
            graphs.append(g)
        def run(reps):
            for _ in range(reps):
                for g, st in zip(graphs, streams):
                    with torch.cuda.stream(st):
                        g.replay()
        run(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(REP)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{env}: {S} slice(s) of {n} lanes, each on its own stream: {dt / (REP * G) * 1e6:.2f} us per step of all {N} lanes")
        del graphs, engs, rings
PY
