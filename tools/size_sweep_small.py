"""Developer aid: size sweep restricted to 33..512 points (one wave per slice), p = 2."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.size_sweep import rate  # noqa: E402

print("%6s %5s | %-24s | %-24s" % ("N", "B", "loss only", "loss + input gradients"))
for N in (48, 64, 100, 128, 200, 256, 300, 400, 500, 512):
    f, fm = rate(N, 512, 256, False, 2)
    t, tm = rate(N, 512, 256, True, 2)
    print("%6d %5d | %9.3e pp/s %7.3f ms | %9.3e pp/s %7.3f ms" % (N, 256, f, fm, t, tm), flush=True)
