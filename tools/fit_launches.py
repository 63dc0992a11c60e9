"""The launches of one C2 fit from a rocprofv3 --kernel-trace CSV (tools/gap_run.sh): offset, gap to the previous launch,
duration, name; then the per-kernel totals.  usage: fit_launches.py <kernel_trace.csv> [fit index from the end, default 2]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:44]) for r in rows))
ap = [i for i, e in enumerate(ev) if e[2].startswith("k_lev_ap<false>")]
# (the bench's later blocks fit with other samplers: take the back-th fit from the end that draws the legacy way)
legacy = [(x, y) for x, y in zip(ap, ap[1:]) if any(e[2].startswith("k_tr_steps") for e in ev[x:y])]
a, b = legacy[-back]
fit = ev[a:b]
t0, prev = fit[0][0], None
for i, (s, e, n) in enumerate(fit):
    print("%3d %8.1f gap %6.1f dur %6.1f %s" % (i, (s - t0) / 1e3, (s - prev) / 1e3 if prev else 0.0, (e - s) / 1e3, n))
    prev = e
busy = sum(e - s for s, e, n in fit)
print("launches %d, span %.1f us, busy %.1f us" % (len(fit), (fit[-1][1] - t0) / 1e3, busy / 1e3))
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in fit:
    tot[n][0] += 1; tot[n][1] += e - s
for n, (k, d) in sorted(tot.items(), key=lambda x: -x[1][1]):
    print("  %-46s x%-3d %8.1f us" % (n, k, d / 1e3))
