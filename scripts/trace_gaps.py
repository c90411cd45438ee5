#!/usr/bin/env python
"""Per-stream view of the last training step in a rocprofv3 kernel trace: busy time and idle gaps
of every (hardware queue, stream), the collectives, and the largest gaps of the main stream.

    python scripts/trace_gaps.py gpurun_out/prof_forced/vm/123_kernel_trace.csv
"""
import collections
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1], newline="")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    sgd = [i for i, r in enumerate(rows) if "fused_sgd" in r["Kernel_Name"] or "fused_lars" in r["Kernel_Name"]]
    ends = [i for k, i in enumerate(sgd)
            if k + 1 == len(sgd) or int(rows[sgd[k + 1]]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) > 2e6]
    win = rows[ends[-2] + 1: ends[-1] + 1]
    t0 = int(win[0]["Start_Timestamp"])
    ms = lambda t: round((int(t) - t0) / 1e6, 3)  # noqa: E731
    print("last step: %.3f ms, %d kernels" % ((int(win[-1]["End_Timestamp"]) - t0) / 1e6, len(win)))
    byq = collections.defaultdict(list)
    for r in win:
        byq[(r["Queue_Id"], r["Stream_Id"])].append(r)
    for q, l in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in l)
        print("queue %s stream %s: %d kernels, busy %.3f ms, first %.3f, last %.3f" % (
            q[0], q[1], len(l), busy / 1e6, ms(l[0]["Start_Timestamp"]), ms(l[-1]["End_Timestamp"])))
    for r in win:
        n = r["Kernel_Name"]
        if "ccl" in n.lower() or "allreduce" in n.lower():
            print("collective: queue %s start %.3f ms dur %.1f us grid %s %s" % (
                r["Queue_Id"], ms(r["Start_Timestamp"]),
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], n[:60]))
    mainq = max(byq, key=lambda k: len(byq[k]))
    l = byq[mainq]
    gaps = [(int(l[i + 1]["Start_Timestamp"]) - int(l[i]["End_Timestamp"]), i) for i in range(len(l) - 1)]
    big = [g for g in gaps if g[0] > 20000]
    print("main stream: %d gaps > 20 us totalling %.3f ms; all gaps %.3f ms" % (
        len(big), sum(g[0] for g in big) / 1e6, sum(max(g[0], 0) for g in gaps) / 1e6))
    short = lambda s: re.sub(r"\(anonymous namespace\)::", "", s)[:48]  # noqa: E731
    for g, i in sorted(big, reverse=True)[:10]:
        print("  %.1f us at %.3f ms: %s -> %s" % (g / 1e3, ms(l[i]["End_Timestamp"]),
                                                 short(l[i]["Kernel_Name"]), short(l[i + 1]["Kernel_Name"])))


if __name__ == "__main__":
    main()
