#!/usr/bin/env python
"""Every conv launch of one training step against its roofline.

  run  : (GPU box, under ``rocprofv3 --kernel-trace``) a few eager flagship steps; the LAST step's conv
         launches are logged with their shapes (``ops/kernels.py::LAUNCH_LOG``) into a JSON file
  join : (anywhere) kernel trace CSV + that JSON -> one row per launch: shape, duration, TFLOP/s,
         compulsory bytes, TB/s, and the share of the launch's roofline time it achieves
         (roofline = max(flops / PEAK_TF, bytes / PEAK_TBS))

  rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python3 scripts/conv_roofline.py run OUT/launches.json
  python scripts/conv_roofline.py join OUT/t_kernel_trace.csv OUT/launches.json > profiles/r3_conv_roofline.txt
"""
import csv
import json
import sys

PEAK_TF = 1000.0   # what the vendor library reaches on large bf16 GEMMs here (profiles/r1_kernel_bench_vs_library.log)
PEAK_TBS = 6.3     # streaming kernels of this repo (profiles/r2_notes.md section 10)


def run(out, model_name="resnet50", batch=256, steps=5):
    import os
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from distributeddeeplearning_amd.data.synthetic import DeviceSyntheticLoader
    from distributeddeeplearning_amd.models import build_model
    from distributeddeeplearning_amd.ops import functional as ops, kernels as K
    from distributeddeeplearning_amd.optim import FusedSGD

    dev = torch.device("cuda", 0)
    torch.manual_seed(42)
    model = build_model(model_name, num_classes=1000).to_compute(dev, torch.bfloat16).train()
    opt = FusedSGD(model.named_parameters(), lr=0.001, momentum=0.9)
    loader = DeviceSyntheticLoader(batch, length=batch * 64, device=dev, dtype=torch.bfloat16)
    for s in range(steps):
        if s == steps - 1:
            torch.cuda.synchronize()
            K.LAUNCH_LOG = []
        x, y = loader.batch()
        opt.zero_grad()
        loss = ops.softmax_cross_entropy(model(x), y)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    with open(out, "w") as fh:
        json.dump(K.LAUNCH_LOG, fh)
    print("logged", len(K.LAUNCH_LOG), "conv launches; loss", float(loss))


def _cost(rec):
    kind, B, Hin, Win, C, Hout, Wout, N, R, S, ostride, idil, M = rec[:13]
    if kind == "conv_igemm":
        addend, stats, ep_y, ep_mask, ep_y2, bwd, c2 = rec[13:20]
        Kred = R * S * C + c2
        flops = 2.0 * M * N * Kred
        # compulsory bytes: input tensor (its visited part), output, filter, epilogue streams
        in_elems = B * Hin * Win * C if (ostride == 1 or R > 1) else M * C
        by = 2.0 * (in_elems + M * N + N * Kred) + 2.0 * M * c2
        by += 2.0 * M * N * (addend + ep_y + ep_y2) + (M * N / 8.0) * ep_mask
        tag = ("dgrad" if bwd else "fwd") + ("+st" if stats and not ep_y else "") + ("+bnbwd" if ep_y else "") + \
              ("+mask" if ep_mask else "") + ("+add" if addend else "") + ("+y2" if ep_y2 else "") + ("+seg2" if c2 else "")
    elif kind == "conv_rows":
        bwd = rec[13]
        flops = 2.0 * M * N * R * S * C
        by = 2.0 * (M * C + M * N + N * R * S * C) + 2.0 * M * N * bwd
        tag = "rows " + ("dgrad+bnbwd" if bwd else "fwd+st")
    else:
        splits = rec[13]
        Kred = M
        flops = 2.0 * M * N * R * S * C
        in_elems = B * Hin * Win * C if (ostride == 1 or R > 1) else M * C
        by = 2.0 * (in_elems + M * N) + 4.0 * N * R * S * C * splits
        tag = "wgrad x%d" % splits
    return flops, by, tag


def join(trace_csv, log_json):
    rows = list(csv.DictReader(open(trace_csv)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    last = max(i for i, r in enumerate(rows) if "pool_gather" in r["Kernel_Name"])
    step = rows[last:]
    log = json.load(open(log_json))
    out = []
    for kind in ("conv_igemm", "conv_wgrad", "conv_rows"):
        ks = [r for r in step if kind + "_kernel" in r["Kernel_Name"] or (kind == "conv_wgrad" and ("conv_wgrad256_kernel" in r["Kernel_Name"] or "conv_wgrad_rows_kernel" in r["Kernel_Name"]))
              or (kind == "conv_rows" and "conv_rows3x3_kernel" in r["Kernel_Name"])]
        ls = [r for r in log if r[0] == kind]
        if len(ks) != len(ls):
            print("# %s: %d kernels in the trace, %d launches logged - not joined" % (kind, len(ks), len(ls)))
            continue
        for k, l in zip(ks, ls):
            dur = (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e3
            flops, by, tag = _cost(l)
            roof = max(flops / (PEAK_TF * 1e12), by / (PEAK_TBS * 1e12)) * 1e6
            name = k["Kernel_Name"]
            tmpl = name[name.index("<"):name.index(">") + 1] if "<" in name else ""
            out.append((int(k["Start_Timestamp"]), k["Stream_Id"], kind, l, tag, tmpl, dur, flops / dur / 1e6, by / dur / 1e6,
                        roof, int(k["Grid_Size_X"]) // max(int(k["Workgroup_Size_X"]), 1) * int(k["Grid_Size_Y"]) * int(k["Grid_Size_Z"])))
    out.sort()
    t0 = int(step[0]["Start_Timestamp"])
    print("# one step, conv launches in start order; roofline = max(flops / %.0f TF/s, bytes / %.1f TB/s)" % (PEAK_TF, PEAK_TBS))
    print("# %8s %2s %-11s %-30s %-22s %6s %8s %7s %6s %7s %5s  %s" % ("t_ms", "st", "kind", "B,Hin,C->Hout,N RxS/s", "what", "blocks", "dur_us",
                                                                      "TF/s", "TB/s", "roof_us", "eff%", "template"))
    tot = {}
    for (ts, st, kind, l, tag, tmpl, dur, tf, tbs, roof, blocks) in out:
        shape = "%d,%d,%d->%d,%d %dx%d/%d" % (l[1], l[2], l[4], l[5], l[7], l[8], l[9], l[10])
        print("  %8.3f %2s %-11s %-30s %-22s %6d %8.1f %7.0f %6.2f %7.1f %5.0f  %s" % ((ts - t0) / 1e6, st, kind, shape, tag, blocks, dur, tf, tbs,
                                                                                     roof, 100.0 * roof / dur, tmpl[:60]))
        key = (kind, "stage%d" % {56: 1, 28: 2, 14: 3, 7: 4}.get(l[5], 0) if l[5] in (56, 28, 14, 7) else "other")
        a = tot.setdefault(key, [0.0, 0.0, 0])
        a[0] += dur
        a[1] += roof
        a[2] += 1
    print("#\n# totals by kind and output resolution (stage): launches, measured us, roofline us")
    for key in sorted(tot):
        a = tot[key]
        print("#   %-11s %-7s %3d  %8.1f  %8.1f  (%.0f%%)" % (key[0], key[1], a[2], a[0], a[1], 100.0 * a[1] / a[0]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], *(sys.argv[3:4]))
    else:
        join(sys.argv[2], sys.argv[3])
