#!/usr/bin/env python
"""Phase view of the last training step in a rocprofv3 kernel trace: forward / backward / update
windows of the main stream (busy time, gaps, kernel classes) and how the weight-gradient stream
fills the backward window.

    python scripts/trace_phases.py gpurun_out/r2prof/step/<host>/<pid>_kernel_trace.csv
"""
import collections
import csv
import re
import sys


def short(s):
    s = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", s)
    s = re.sub(r"^void ", "", s)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", s)
    return (m.group(1) + (m.group(2) or ""))[:44] if m else s[:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1], newline="")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    # a step = from one loader gather to the next (with early bucket updates the optimizer kernels are spread over
    # backward, so they no longer delimit steps); without a device loader: the old rule (gaps after the optimizer)
    gath = [i for i, r in enumerate(rows) if "pool_gather_kernel" in r["Kernel_Name"]]
    if len(gath) >= 3:
        win = rows[gath[-2]: gath[-1]]
    else:
        sgd = [i for i, r in enumerate(rows) if "fused_sgd" in r["Kernel_Name"] or "fused_lars" in r["Kernel_Name"]]
        ends = [i for k, i in enumerate(sgd) if k + 1 == len(sgd) or rows[sgd[k + 1]]["s"] - rows[i]["e"] > 2e6]
        win = rows[ends[-2] + 1: ends[-1] + 1]
    t0 = win[0]["s"]
    byq = collections.defaultdict(list)
    for r in win:
        byq[(r["Queue_Id"], r["Stream_Id"])].append(r)
    mainq = max(byq, key=lambda k: len(byq[k]))
    main_rows = byq[mainq]
    side_rows = [r for q, l in byq.items() if q != mainq for r in l]
    xent = next(r for r in main_rows if "xent" in r["Kernel_Name"] or "cross_entropy" in r["Kernel_Name"])
    upd = [r for r in main_rows if "fused_sgd" in r["Kernel_Name"] or "fused_lars" in r["Kernel_Name"]]
    last_bwd = max((r for r in main_rows if "conv_" in r["Kernel_Name"] or "stem_" in r["Kernel_Name"]), key=lambda r: r["e"])
    first_sgd = next((r for r in upd if r["s"] >= last_bwd["e"]), win[-1])
    phases = [("forward", t0, xent["e"]), ("backward", xent["e"], first_sgd["s"]), ("update", first_sgd["s"], max(r["e"] for r in win))]
    print("last step %.3f ms, %d kernels (%d main stream, %d other streams)" % (
        (win[-1]["e"] - t0) / 1e6, len(win), len(main_rows), len(side_rows)))
    if len(sys.argv) > 2 and sys.argv[2] == "--list":   # every kernel of the step in start order
        for r in win:
            print("%9.3f ms %s %8.1f us  grid %-8s %s" % ((r["s"] - t0) / 1e6, "main " if (r["Queue_Id"], r["Stream_Id"]) == mainq else "other",
                                                        (r["e"] - r["s"]) / 1e3, r["Grid_Size_X"], short(r["Kernel_Name"])))
        return
    for name, a, b in phases:
        mr = [r for r in main_rows if a <= r["s"] < b]
        sr = [r for r in side_rows if a <= r["s"] < b]
        busy = sum(r["e"] - r["s"] for r in mr)
        gaps = sum(max(mr[i + 1]["s"] - mr[i]["e"], 0) for i in range(len(mr) - 1))
        sbusy = sum(r["e"] - r["s"] for r in sr)
        print("\n%s: %.3f ms window; main stream %d kernels busy %.3f ms, gaps %.3f ms; other streams %d kernels busy %.3f ms" % (
            name, (b - a) / 1e6, len(mr), busy / 1e6, gaps / 1e6, len(sr), sbusy / 1e6))
        if sr and mr:
            print("  other streams: first start %.3f ms, last end %.3f ms after window start; main's last kernel ends %.3f ms" % (
                (sr[0]["s"] - a) / 1e6, (max(r["e"] for r in sr) - a) / 1e6, (mr[-1]["e"] - a) / 1e6))
        for label, l in (("main", mr), ("other", sr)):
            agg = collections.defaultdict(lambda: [0, 0])
            for r in l:
                k = short(r["Kernel_Name"])
                agg[k][0] += r["e"] - r["s"]
                agg[k][1] += 1
            for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
                print("    %-5s %-46s %4d x  %8.3f ms" % (label, k, n, t / 1e6))
        # gap histogram of the main stream
        gl = [mr[i + 1]["s"] - mr[i]["e"] for i in range(len(mr) - 1)]
        for g, i in sorted(((g, i) for i, g in enumerate(gl) if g > 20000), reverse=True)[:40]:
            during = [r for r in sr if r["s"] < mr[i + 1]["s"] and r["e"] > mr[i]["e"]]
            print("    gap %6.1f us at %7.3f ms: %s -> %s | other streams meanwhile: %s" % (
                g / 1e3, (mr[i]["e"] - a) / 1e6, short(mr[i]["Kernel_Name"])[:30], short(mr[i + 1]["Kernel_Name"])[:30],
                ", ".join("%s(%.0f us, ends %+.1f us vs gap end)" % (short(r["Kernel_Name"])[:26], (r["e"] - r["s"]) / 1e3,
                                                                    (r["e"] - mr[i + 1]["s"]) / 1e3) for r in during[:3])))
        if gl:
            bins = [(0, 1000), (1000, 2000), (2000, 4000), (4000, 8000), (8000, 20000), (20000, 10 ** 9)]
            print("  main-stream gaps: " + ", ".join("%d-%s us: %d (%.3f ms)" % (
                lo // 1000, "inf" if hi > 10 ** 8 else hi // 1000, sum(1 for g in gl if lo <= g < hi),
                sum(g for g in gl if lo <= g < hi) / 1e6) for lo, hi in bins))


if __name__ == "__main__":
    main()
