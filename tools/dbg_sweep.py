import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shw_amd as shw
from size_sweep import rate
for cfg in [(2048, True, 2), (2048, True, 2), (512, False, 2), (512, False, 2), (512, True, 1), (512, True, 1), (2048, True, 2)]:
    N, train, p = cfg
    r, ms = rate(N, 512, 64, train, p)
    print(cfg, "%.3f ms" % ms, flush=True)
