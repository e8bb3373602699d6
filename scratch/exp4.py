import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from exp import timeit
os.environ.setdefault("PF_LPW","64")
for n in (65536, 262144):
    timeit("fast full", n=n, noise="philox", autoreset="next_step")
    timeit("fast settle=0", n=n, noise="philox", autoreset="next_step", settle=0)
    timeit("fast settle=0 noise off", n=n, noise="off", autoreset="next_step", settle=0)
    timeit("fast noise off", n=n, noise="off", autoreset="next_step")
