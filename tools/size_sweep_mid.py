"""Developer aid: size sweep restricted to 513..2048 points (the keys-per-lane classes of round 3), p = 2."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.size_sweep import rate  # noqa: E402

print("%6s %5s | %-24s | %-24s" % ("N", "B", "loss only", "loss + input gradients"))
for N in (600, 768, 1000, 1024, 1200, 1280, 1500, 1536, 1700, 1792, 2000, 2048):
    f, fm = rate(N, 512, 64, False, 2)
    t, tm = rate(N, 512, 64, True, 2)
    print("%6d %5d | %9.3e pp/s %7.3f ms | %9.3e pp/s %7.3f ms" % (N, 64, f, fm, t, tm), flush=True)
