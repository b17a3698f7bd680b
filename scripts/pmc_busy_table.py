#!/usr/bin/env python3
"""MFMA pipe utilisation table from scripts/pmc_summary.py's per-(kernel, grid) counter averages.

usage: python scripts/pmc_busy_table.py <pmc_mfma_busy_raw.txt> [<out.json>] > profiles/rNN_pmc_mfma_busy.txt
out.json: the dominant kernel's (conv_mfma_h8_kernel) dispatch-weighted MFMA-busy fraction + the hash of the kernel sources it was measured
on -- bench.py reports it as roofline.mfma_busy for that build only.
busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); only kernels that issue MFMAs are listed."""
import re
import sys


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
    print("# MFMA pipe utilisation of the matrix-core kernels (one MI355X, bench.py default workload)")
    print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES "
          "SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline")
    print("# busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), averaged over the dispatches of a (kernel, grid); "
          "v_mfma_f32_16x16x32_bf16 = 16 cycles each")
    print("%-54s %-12s %10s %12s %14s %12s" % ("kernel", "grid", "dispatches", "MFMA busy %", "MFMA instr", "VALU instr"))
    dom = []
    for r in sorted(rows, key=lambda r: (r["name"], r["grid"])):
        if r.get("SQ_INSTS_MFMA", 0) <= 0 or r.get("SQ_BUSY_CU_CYCLES", 0) <= 0:
            continue
        busy = 100.0 * r["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * r["SQ_BUSY_CU_CYCLES"])
        name = re.sub(r"^void ", "", r["name"])
        print("%-54s %-12s %10d %12.1f %14d %12d" % (name[:54], r["grid"], r["n"], busy, r["SQ_INSTS_MFMA"], r["SQ_INSTS_VALU"]))
        if "conv_mfma_h8_kernel" in name:
            dom.append((name, r["grid"], r["n"], r["SQ_VALU_MFMA_BUSY_CYCLES"], r["SQ_BUSY_CU_CYCLES"]))
    if len(sys.argv) > 2 and dom:
        import json
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import rcgan_amd  # noqa: F401
        from rcgan_amd import _lib
        num = sum(n * b for _, _, n, b, _ in dom)
        den = sum(n * 4.0 * c for _, _, n, _, c in dom)
        json.dump({"source_sha16": _lib.source_hash(), "kernel": "conv_mfma_h8_kernel",
                   "mfma_busy": num / den, "definition": "sum over its dispatches of SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), bench.py default workload",
                   "per_grid": [{"kernel": k, "grid": g, "dispatches": n, "mfma_busy": b / (4.0 * c)} for k, g, n, b, c in dom]},
                  open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
