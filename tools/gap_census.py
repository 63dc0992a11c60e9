"""Idle gaps of the GPU inside one C2 fit, from a rocprofv3 --kernel-trace CSV: which kernel
boundaries does the device wait at, and for how long?  usage: gap_census.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in rows))
# fits are separated by the k_sid launches (one per fit): take the last complete one
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_runmin_argmax") or e[2].startswith("void k_lev_r") or e[2].startswith("k_lev_a")]
sid = [i for i, e in enumerate(ev) if e[2].startswith("k_sid")]
a, b = sid[-2], sid[-1]
# walk back from k_sid to the first anchor-round kernel of that fit
def fit_begin(i):
    j = i
    while j > 0 and (ev[j - 1][2].startswith("k_runmin") or ev[j - 1][2].startswith("void k_lev_r") or ev[j - 1][2].startswith("k_lev_a") or "copyBuffer" in ev[j - 1][2] or "fillBuffer" in ev[j - 1][2]):
        j -= 1
    return j
fa, fb = fit_begin(a), fit_begin(b)
fit = ev[fa:fb]
busy = sum(e[1] - e[0] for e in fit)
span = fit[-1][1] - fit[0][0]
print("kernels %d, span %.3f ms, busy %.3f ms, idle %.3f ms" % (len(fit), span / 1e6, busy / 1e6, (span - busy) / 1e6))
gaps = []
for x, y in zip(fit, fit[1:]):
    g = y[0] - x[1]
    if g > 0:
        gaps.append((g, x[2], y[2]))
tot = collections.Counter()
for g, x, y in gaps:
    tot["<5us" if g < 5000 else "5-20us" if g < 20000 else ">=20us"] += g
print({k: round(v / 1e6, 3) for k, v in tot.items()})
for g, x, y in sorted(gaps, reverse=True)[:28]:
    print("  %8.1f us  after %-40s before %s" % (g / 1e3, x, y))
