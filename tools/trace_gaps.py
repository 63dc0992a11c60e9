"""Idle intervals of the device in a rocprofv3 kernel + memory-copy trace (csv): python tools/trace_gaps.py <dir> [min_ms]"""
import csv
import glob
import os
import sys


def rows(path):
    with open(path, newline="") as f:
        yield from csv.DictReader(f)


def main():
    d = sys.argv[1]
    min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    ops = []
    for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in rows(p):
            ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
    for p in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in rows(p):
            ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") ))
    ops.sort()
    if not ops:
        print("no ops")
        return
    # the last fit: what follows the last host-to-device copy of more than 2 ms (the constructor's upload of the rows)
    start_i = 0
    for i, (s_, e_, n_) in enumerate(ops):
        if n_.startswith("COPY") and "HOST_TO_DEVICE" in n_ and e_ - s_ > 2e6:
            start_i = i + 1
    ops = ops[start_i:]
    print("ops in the last fit:", len(ops), "span ms:", (ops[-1][1] - ops[0][0]) / 1e6)
    busy_end = ops[0][1]
    busy = ops[0][1] - ops[0][0]
    prev = ops[0][2]
    gaps = []
    small = 0.0
    for s, e, n in ops[1:]:
        if s > busy_end:
            g = (s - busy_end) / 1e6
            if g >= min_ms:
                gaps.append((g, prev, n))
            else:
                small += g
        if e > busy_end:
            busy += (e - max(s, busy_end))
            busy_end = e
            prev = n
    print("busy ms: %.1f   gaps >= %.1f ms: %.1f ms in %d   smaller gaps: %.1f ms" % (busy / 1e6, min_ms, sum(g for g, _, _ in gaps), len(gaps), small))
    for g, a, b in gaps:
        print("%8.2f ms   after %-60s before %s" % (g, a, b))
    cp = {}
    for s, e, n in ops:
        if n.startswith("COPY"):
            cp.setdefault(n, [0, 0.0])
            cp[n][0] += 1
            cp[n][1] += (e - s) / 1e6
    print("copies:", cp)


if __name__ == "__main__":
    main()
