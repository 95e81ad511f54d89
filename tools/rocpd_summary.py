"""Per-forward kernel summary from a rocprofv3 rocpd database (the default output format of
`rocprofv3 --kernel-trace --stats`): python3 tools/rocpd_summary.py <db> <n_forwards> [anchor-kernel] [per-forward-count]
The last n_forwards*count occurrences of the anchor kernel delimit the timed region."""
import collections
import sqlite3
import sys

db, nf = sys.argv[1], int(sys.argv[2])
anchor = sys.argv[3] if len(sys.argv) > 3 else "k_spectrum"
per = int(sys.argv[4]) if len(sys.argv) > 4 else 3
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
first = idx[-nf * per]
# back up to the start of that forward: the kernels between the previous forward's last anchor-block and this one
prev = idx[-nf * per - 1] if len(idx) > nf * per else 0
span = [r for r in rows[first:]]
agg = collections.OrderedDict()
for name, s, e in span:
    a = agg.setdefault(name[:110], [0, 0])
    a[0] += 1
    a[1] += e - s
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%9.1f us/fwd  x%5.1f  %s" % (v[1] / nf / 1e3, v[0] / nf, k))
print("GPU busy per forward: %.1f us; wall span per forward: %.1f us" % (tot / nf / 1e3, (span[-1][2] - span[0][1]) / nf / 1e3))
