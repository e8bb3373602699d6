import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    from exp import timeit
    timeit(sys.argv[1], n=int(sys.argv[2]), noise="philox", autoreset="next_step")
else:
    for n in (16384, 65536, 131072):
        for lpw in (64, 32):
            subprocess.run([sys.executable, __file__, f"lpw={lpw}", str(n)], env=dict(os.environ, PF_LPW=str(lpw)))
