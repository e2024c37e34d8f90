"""Developer aid: p = 1 (level median) size sweep up to 2048 points, loss and training step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.size_sweep import rate  # noqa: E402

print("%6s %5s | %-24s | %-24s" % ("N", "B", "loss only", "loss + input gradients"))
for N in (300, 512, 600, 1000, 1024, 1200, 1500, 2000, 2048):
    f, fm = rate(N, 512, 64, False, 1)
    t, tm = rate(N, 512, 64, True, 1)
    print("%6d %5d | %9.3e pp/s %7.3f ms | %9.3e pp/s %7.3f ms" % (N, 64, f, fm, t, tm), flush=True)
