#!/usr/bin/env python3
"""MFMA pipe utilisation table from scripts/pmc_summary.py's per-(kernel, grid) counter averages.

usage: python scripts/pmc_busy_table.py <pmc_mfma_busy_raw.txt> [<out.json> [<kernel_stats_by_grid.csv> [<source_sha16.txt>]]] > profiles/rNN_pmc_mfma_busy.txt

busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) per (kernel, grid); only kernels that issue MFMAs are listed.
out.json (bench.py copies it into its roofline object for the build it was measured on):
  mfma_busy                      the dominant kernel (conv_mfma_h8_kernel), its grids weighted by the TIME they take in the
                                 un-counted kernel-trace pass (kernel_stats_by_grid.csv of the same make_profiles.sh run; without
                                 that file: dispatch-weighted, as rounds 2-4 reported it)
  time_share                     the dominant kernel's share of the iteration's kernel time
  conv_mfma_busy_time_weighted   every convolution kernel that issues MFMAs (forward, data and filter gradients, image-end, fused
                                 stages): sum(time x busy) / sum(time) -- the utilisation of conv2d as a whole, not of its best kernel
  conv_time_share                those kernels' share of the iteration's kernel time
source_sha16.txt: the hash make_profiles.sh recorded WHEN IT MEASURED; refused if it is not the current tree's."""
import csv
import json
import os
import re
import sys


def norm(name):
    return re.sub(r"\s+", "", re.sub(r"^void ", "", name))


def main():
    rows, cur = [], None
    for line in open(sys.argv[1]):
        m = re.match(r"(\S.*) grid \((\d+), (\d+), (\d+)\) \((\d+) dispatches\)", line)
        if m:
            cur = {"name": m.group(1), "grid": "(%s,%s)" % (m.group(2), m.group(3)), "n": int(m.group(5))}
            rows.append(cur)
            continue
        m = re.match(r"\s+(\w+)\s+([\d.]+)", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    # time per (kernel, grid) from the kernel-trace pass
    times, total_time = {}, 0.0
    if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
        for r in csv.DictReader(open(sys.argv[3])):
            m = re.match(r"(.*) grid \((\d+),(\d+)\)$", r["Name"])
            t = float(r["TotalDurationNs"])
            total_time += t
            if m:
                times[(norm(m.group(1)), "(%s,%s)" % (m.group(2), m.group(3)))] = t
    print("# MFMA pipe utilisation of the matrix-core kernels (one MI355X, bench.py default workload)")
    print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES "
          "SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline")
    print("# busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), averaged over the dispatches of a (kernel, grid); "
          "v_mfma_f32_16x16x32_bf16 = 16 cycles each; time share = the (kernel, grid)'s part of the iteration's kernel time in the un-counted kernel-trace pass")
    print("%-54s %-12s %10s %12s %14s %12s %12s" % ("kernel", "grid", "dispatches", "MFMA busy %", "MFMA instr", "VALU instr", "time share %"))
    dom, conv = [], []
    for r in sorted(rows, key=lambda r: (r["name"], r["grid"])):
        if r.get("SQ_INSTS_MFMA", 0) <= 0 or r.get("SQ_BUSY_CU_CYCLES", 0) <= 0:
            continue
        busy = r["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * r["SQ_BUSY_CU_CYCLES"])
        name = re.sub(r"^void ", "", r["name"])
        t = times.get((norm(name), r["grid"]))
        print("%-54s %-12s %10d %12.1f %14d %12d %12s" % (name[:54], r["grid"], r["n"], 100.0 * busy, r["SQ_INSTS_MFMA"], r["SQ_INSTS_VALU"],
                                                        "%.2f" % (100.0 * t / total_time) if t is not None and total_time else "-"))
        item = dict(kernel=name, grid=r["grid"], dispatches=r["n"], mfma_busy=busy, time_ns=t)
        if name.startswith("conv_"):
            conv.append(item)
        if "conv_mfma_h8_kernel" in name:
            dom.append(item)
    if len(sys.argv) > 2 and dom:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import rcgan_amd  # noqa: F401
        from rcgan_amd import _lib
        sha = _lib.source_hash()
        if len(sys.argv) > 4:
            if not os.path.exists(sys.argv[4]):      # no record of what the run measured: never stamp it with the tree's hash
                raise SystemExit("pmc_busy_table.py: %s is missing -- the run directory does not say which sources it measured: not published" % sys.argv[4])
            measured = open(sys.argv[4]).read().strip()
            if measured != sha:
                raise SystemExit("pmc_busy_table.py: the counters were measured on sources %s, the tree is %s: not published" % (measured, sha))

        def weighted(items):
            if items and all(i["time_ns"] is not None for i in items):
                return sum(i["time_ns"] * i["mfma_busy"] for i in items) / sum(i["time_ns"] for i in items), "time"
            return sum(i["dispatches"] * i["mfma_busy"] for i in items) / sum(i["dispatches"] for i in items), "dispatches"
        out = {"source_sha16": sha, "kernel": "conv_mfma_h8_kernel"}
        out["mfma_busy"], out["mfma_busy_weighting"] = weighted(dom)
        out["definition"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) per (kernel, grid), bench.py default workload; "
                             "grids combined by their time in the kernel-trace pass")
        timed = [c for c in conv if c["time_ns"] is not None]
        if timed and total_time:
            out["time_share"] = sum(i["time_ns"] for i in dom if i["time_ns"] is not None) / total_time
            out["conv_mfma_busy_time_weighted"] = sum(c["time_ns"] * c["mfma_busy"] for c in timed) / sum(c["time_ns"] for c in timed)
            out["conv_time_share"] = sum(c["time_ns"] for c in timed) / total_time
            out["conv_kernels_counted"] = len(timed)
            out["conv_kernels_without_time"] = [c["kernel"] + " " + c["grid"] for c in conv if c["time_ns"] is None]
        out["per_grid"] = [{"kernel": i["kernel"], "grid": i["grid"], "dispatches": i["dispatches"], "mfma_busy": i["mfma_busy"],
                            "time_share": (i["time_ns"] / total_time if i["time_ns"] is not None and total_time else None)} for i in dom]
        json.dump(out, open(sys.argv[2], "w"), indent=1)
        if "conv_mfma_busy_time_weighted" in out:
            print("# all convolution kernels that issue MFMAs (%d (kernel, grid) rows, %.1f %% of the iteration's kernel time): time-weighted MFMA busy %.1f %%"
                  % (len(timed), 100 * out["conv_time_share"], 100 * out["conv_mfma_busy_time_weighted"]))
            print("# dominant kernel conv_mfma_h8_kernel: %.1f %% of the kernel time, time-weighted MFMA busy %.1f %%" % (100 * out["time_share"], 100 * out["mfma_busy"]))


if __name__ == "__main__":
    main()
