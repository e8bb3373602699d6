R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests -m gpu -q --tb=short -k "rollout or dogfight" 2>&1 | tail -8
cd /tmp; python - <<'PY'
import sys, os, torch, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import bench
n=65536
eng = bench.make_engine("dogfight", n, torch.device("cuda:0"), 0, "philox")
ring=[torch.empty(n,4,device="cuda:0") for _ in range(50)]
for i,a in enumerate(ring):
    eng.sample_actions(a,i); a.mul_(0.15); a[:,3]+=0.4
seq=torch.stack(ring)
eng.env_reset(); torch.cuda.synchronize()
for rep in range(2):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(50): eng.env_step(ring[i])
    e1.record(); torch.cuda.synchronize()
    print("per-step launches: %.1f us/step"%(e0.elapsed_time(e1)/50*1e3))
eng.env_reset(); torch.cuda.synchronize()
eng.rollout(50, actions=seq); torch.cuda.synchronize()
eng.env_reset(); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record(); eng.rollout(50, actions=seq); e1.record(); torch.cuda.synchronize()
print("state-resident rollout: %.1f us/step"%(e0.elapsed_time(e1)/50*1e3))
PY
