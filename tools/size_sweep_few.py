"""Developer aid: a few mid sizes, p = 2 (variant libraries)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.size_sweep import rate  # noqa: E402

tag = os.path.basename(os.environ.get("SHW_LIB_PATH", "default"))
for N in (1200, 1500, 1700, 2000):
    f, fm = rate(N, 512, 64, False, 2)
    t, tm = rate(N, 512, 64, True, 2)
    print("%-18s %6d | loss %7.3f ms | train %7.3f ms" % (tag, N, fm, tm), flush=True)
