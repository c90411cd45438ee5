#!/usr/bin/env python
"""Condense rocprofv3 --pmc counter_collection.csv files (one directory per pass, as written by
scripts/prof_kernel.sh) into one markdown table: a row per (kernel, grid) of our own kernels with
the mean of every counter plus derived figures (MFMA-busy share, LDS conflict share, HBM bytes).

    python scripts/pmc_summary.py gpurun_out/pmc_a3_c4 > profiles/pmc_a3_c4.md
"""
import collections
import csv
import glob
import os
import re
import sys

OURS = ("conv_igemm_kernel", "conv_wgrad", "lin_fused", "gemm256", "step_aux", "bn_lin", "conv_rows", "splitk_reduce", "pool_gather", "subsample2", "bump_counters", "stem_", "lars_", "bn_", "maxpool", "label_rank", "segmented_sqnorms", "nchw_to_nhwc", "colsum", "bias_add", "stem_pool",
        "avgpool", "softmax_xent", "fused_sgd", "fused_lars", "weight_flip", "stem_pack",
        "xgmi_allreduce", "pack_kernel", "philox")


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?([A-Za-z0-9_]+(?:<[^(]*>)?)", name)
    s = m.group(1) if m else name[:60]
    m2 = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(.*?)EEv", name)
    if m2:
        s = m2.group(1) + "<" + m2.group(2) + ">"
    return s[:70]


def main():
    root = sys.argv[1]
    data = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for path in sorted(glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv"))):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if not any(t in k for t in OURS):
                    continue
                key = (short(k), int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1))
                data[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row["Counter_Name"] in ("SQ_WAVES", "FETCH_SIZE", "WRITE_SIZE"):
                    data[key]["dur_us"].append((int(row["End_Timestamp"]) -
                                                int(row["Start_Timestamp"])) / 1e3)
                meta[key] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["LDS_Block_Size"],
                             row["Scratch_Size"])
    ctrs = sorted({c for v in data.values() for c in v})
    print("# rocprofv3 PMC summary: %s\n" % os.path.basename(root.rstrip("/")))
    print("Mean per dispatch.  `dur_us` = dispatch end - start while counting (slower than an "
          "unprofiled run); `mfma_util%` = SQ_VALU_MFMA_BUSY_CYCLES / (dur x 2.4 GHz x 1024 SIMDs) "
          "— the share of the chip's matrix-pipe cycles in use (16 busy cycles per "
          "v_mfma_f32_16x16x32_bf16); `TF/s` = SQ_INSTS_MFMA x 16384 FLOP / dur; `wait%` = "
          "SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (wave time spent waiting on any counter); `lds_conf%` "
          "= SQ_LDS_BANK_CONFLICT / SQ_WAVE_CYCLES; FETCH/WRITE_SIZE are KiB of HBM traffic "
          "(reads served by L2 / the 256 MB infinity cache are not in FETCH_SIZE); `GB/s` = "
          "(FETCH+WRITE) / dur.\n")
    hdr = ["kernel", "blocks", "vgpr", "agpr", "lds", "scratch"] + ctrs + ["mfma_util%", "TF/s", "wait%",
                                                                           "lds_conf%", "hbm_MiB", "GB/s"]
    print("| " + " | ".join(hdr) + " |")
    print("|" + "---|" * len(hdr))
    for key in sorted(data):
        v = {c: sum(x) / len(x) for c, x in data[key].items()}
        row = [key[0], str(key[1])] + list(meta[key])
        row += ["%.4g" % v[c] if c in v else "" for c in ctrs]
        dur = v.get("dur_us")
        row.append("%.1f" % (100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur * 2400.0 * 1024))
                   if dur and "SQ_VALU_MFMA_BUSY_CYCLES" in v else "")
        row.append("%.0f" % (v["SQ_INSTS_MFMA"] * 16384 / dur / 1e6)
                   if dur and v.get("SQ_INSTS_MFMA") else "")
        wc = v.get("SQ_WAVE_CYCLES")
        row.append("%.0f" % (100.0 * v["SQ_WAIT_INST_ANY"] / wc)
                   if wc and "SQ_WAIT_INST_ANY" in v else "")
        row.append("%.2f" % (100.0 * v["SQ_LDS_BANK_CONFLICT"] / wc)
                   if wc and "SQ_LDS_BANK_CONFLICT" in v else "")
        hbm = (v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) / 1024.0
        row.append("%.1f" % hbm if hbm else "")
        row.append("%.0f" % (hbm * 1.048576 / dur * 1e3) if hbm and dur else "")
        print("| " + " | ".join(row) + " |")


if __name__ == "__main__":
    main()
