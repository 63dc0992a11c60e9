"""Kernel sequence of the last C2 fit in a rocprofv3 --kernel-trace CSV: start offset, duration, gap before, name.
usage: fit_sequence.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]) for r in rows))
sid = [i for i, e in enumerate(ev) if e[2].startswith("k_sid")]
def fit_begin(i):
    j = i
    while j > 0 and (ev[j - 1][2].startswith("k_runmin") or ev[j - 1][2].startswith("k_lev_a") or "copyBuffer" in ev[j - 1][2] or "fillBuffer" in ev[j - 1][2] or ev[j - 1][2].startswith("k_fill") or ev[j - 1][2].startswith("k_anchor_rank")):
        j -= 1
    return j
fa, fb = fit_begin(sid[-2]), fit_begin(sid[-1])
fit = ev[fa:fb]
t0 = fit[0][0]
prev = None
for s, e, n in fit:
    print("%8.1f us  dur %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, n))
    prev = e
