import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from exp import timeit
    timeit(sys.argv[1], n=int(sys.argv[2]), noise="philox", autoreset="next_step")
else:
    for n in (65536, 262144):
        for lpw, wps in ((64,2),(64,4),(32,2),(32,4),(16,2),(16,4)):
            env = dict(os.environ, PF_LPW=str(lpw), PF_WPS=str(wps))
            subprocess.run([sys.executable, __file__, f"fast lpw={lpw} wps={wps}", str(n)], env=env)
    env = dict(os.environ, PF_DISABLE_FAST="1")
    subprocess.run([sys.executable, __file__, "generic", "65536"], env=env)
