import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    from exp import timeit
    timeit(sys.argv[1], n=int(sys.argv[2]), noise="philox", autoreset="next_step")
else:
    for dbg in (0, 1, 2, 3, 4, 7):
        subprocess.run([sys.executable, __file__, f"dbg={dbg}", "65536"], env=dict(os.environ, PF_DBG=str(dbg)))
