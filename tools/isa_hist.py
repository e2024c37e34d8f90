"""Instruction histogram of one kernel in a hipcc -save-temps .s file (developer tool)."""
import collections, re, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = None
for i, l in enumerate(lines):
    if l.startswith('_Z') and pat in l.split(':')[0]:
        start = i
        break
ops = collections.Counter()
for l in lines[start + 1:]:
    if l.startswith('.Lfunc_end'):
        break
    t = l.strip()
    if not t or t[0] in '.;/' or t.endswith(':'):
        continue
    ops[t.split()[0]] += 1
print(lines[start].split(':')[0], sum(ops.values()))
for k, v in ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"  {k:34s}{v}")
