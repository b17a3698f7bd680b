#!/usr/bin/env python3
"""HBM traffic per launch of the dominant kernel from the two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
summarised by scripts/pmc_summary.py  ->  the JSON that bench.py reports as roofline.traffic.

usage: python scripts/pmc_traffic_json.py <pmc_FETCH_SIZE_conv_h8.txt> <pmc_WRITE_SIZE_conv_h8.txt> <out.json> [<source_sha16.txt>]
source_sha16.txt: the hash scripts/make_profiles.sh recorded WHEN IT MEASURED; refused if it is not the current tree's.

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is reported in KB and counts the 128-byte requests of a wide
coalesced stream at 64 bytes on gfx950 -> x2; WRITE_SIZE (KB) as reported."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(path):
    # (several template instances of the kernel may share a grid -- the plain and the sub-pixel form at 1280 / 512 workgroups: merged,
    # dispatch-weighted)
    acc, cur = {}, None
    for line in open(path):
        m = re.match(r"(\S.*) grid \((\d+), (\d+), (\d+)\) \((\d+) dispatches\)", line)
        if m:
            cur = (int(m.group(2)), int(m.group(5)))
            continue
        m = re.match(r"\s+(\w+)\s+([\d.]+)", line)
        if m and cur:
            n0, s0 = acc.get(cur[0], (0, 0.0))
            acc[cur[0]] = (n0 + cur[1], s0 + cur[1] * float(m.group(2)))
    return {g: (n, s / n) for g, (n, s) in acc.items()}


def main():
    fetch, write = parse(sys.argv[1]), parse(sys.argv[2])
    n = sum(v[0] for v in fetch.values())
    f_avg = sum(v[0] * v[1] for v in fetch.values()) / n
    w_avg = sum(v[0] * v[1] for v in write.values()) / n
    # algorithmic bytes of the launch mix at the end of round 2 (the up blocks' shortcuts run on the low-resolution grid of other
    # kernels; G.Block.3.Conv1's data gradient in its sub-pixel form on a 128-workgroup grid): 512 workgroups = G step, n = 128
    # at 32x32 -- Conv1 forward (sub-pixel: reads the 16x16 input), Conv2 forward (reads 32x32 + the quarter-size shortcut in its
    # epilogue), Conv2's data gradient (reads 32x32); 1280 workgroups = Conv1 and Conv2 forward of the 5 critic batches, n = 320
    px = {512: 128 * 1024, 1280: 320 * 1024}
    alg = {}
    for g, (cnt, _) in fetch.items():
        full = px[g] * 256 * 2
        c1, c2, d2 = full / 4, full + full / 4, full
        reads = ((c1 + c2) / 2 if g == 1280 else (c1 + c2 + d2) / 3) + 1.2e6
        alg[g] = reads + full
    alg_avg = sum(fetch[g][0] * alg[g] for g in fetch) / n
    import rcgan_amd  # noqa: F401
    from rcgan_amd import _lib
    if len(sys.argv) > 4:
        if not os.path.exists(sys.argv[4]):      # no record of what the run measured: never stamp it with the tree's hash
            raise SystemExit("pmc_traffic_json.py: %s is missing -- the run directory does not say which sources it measured: not published" % sys.argv[4])
        measured = open(sys.argv[4]).read().strip()
        if measured != _lib.source_hash():
            raise SystemExit("pmc_traffic_json.py: the counters were measured on sources %s, the tree is %s: not published" % (measured, _lib.source_hash()))
    out = {
        "source_sha16": _lib.source_hash(),      # bench.py attaches the figure to a run only on the same build of the kernels
        "kernel": "conv_mfma_h8_kernel (256 x 256 tile, pixel operand as an LDS patch fetched once per 64-channel chunk; the upsample-3x3 layers in sub-pixel form, shortcuts before the upsample)",
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (two separate passes; scripts/make_profiles.sh)",
        "fetch_size_kb_avg": f_avg, "write_size_kb_avg": w_avg, "launches": n,
        "per_grid": {str(g): {"launches": fetch[g][0], "fetch_kb": fetch[g][1], "write_kb": write[g][1], "algorithmic_bytes": alg[g]} for g in sorted(fetch)},
        "correction": "FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported",
        "traffic_bytes_per_launch": (2 * f_avg + w_avg) * 1024,
        "algorithmic_bytes_per_launch": alg_avg,
        "algorithmic_note": "reads = the conv input at its STORED resolution (the nearest-2x upsample is folded into the load) + 1.2 MB of filters; writes = the output",
    }
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
