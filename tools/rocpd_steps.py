"""Dev tool: per-kernel averages and per-step busy/wall medians from a rocprofv3 rocpd database (…_results.db)."""
import sqlite3, collections, statistics as S, sys
c = sqlite3.connect(sys.argv[1])
for r in c.execute("select name, count(*), avg(end-start)/1000.0, sum(end-start)/1e6 from kernels group by name order by 4 desc limit 22"):
    print(f"{r[0][:84]:84s} {r[1]:6d} {r[2]:9.1f} us {r[3]:9.2f} ms")
rows = list(c.execute("select name,start,end from kernels order by start"))
steps, cur = [], None
for n, s, e in rows:
    if "step_select" in n:
        if cur: steps.append(cur)
        cur = []
    if cur is not None: cur.append((n, s, e))
agg = collections.defaultdict(list)
for i in range(len(steps) - 1):
    st = steps[i]
    kind = "chain" if any("pw_chain" in n for n, _, _ in st) else "layers"
    wall = (steps[i + 1][0][1] - st[0][1]) / 1000
    busy = sum(e - s for _, s, e in st) / 1000
    big = sum(e - s for n, s, e in st if "256, 256" in n) / 1000
    if wall < 2 * busy: agg[kind].append((wall, busy, big, busy - big, len(st)))
for k, v in agg.items():
    print(k, len(v), "median wall/busy/big-GEMM/small/launches:", [round(S.median(x[i] for x in v), 1) for i in range(5)])
