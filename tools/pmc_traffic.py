#!/usr/bin/env python3
"""HBM-side bytes per launch from rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section) over the program named in the
optional 4th argument (default: "bench.py itself"; tools/profile_gemm_pmc.sh passes "tools/gemm_ab.py at the block's four shapes
(bench.py passes fault inside librocprofiler-sdk)" -- the string goes into the bench line's traffic_source, so the record says what
was measured).

FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots: 3 + 2).  Run, program directly after `--`:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python3 tools/pmc_traffic.py gpurun_out/r02/pmc_fetch gpurun_out/r02/pmc_write profiles/r02/bench_pmc_traffic.json
Units and gfx950 corrections: both counters are KiB at the L2 <-> fabric boundary (Infinity-Cache hits included);
FETCH_SIZE tallies 128-B requests at 64 B, so wide coalesced reads are DOUBLED; WRITE_SIZE is exact for 16-B stores.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname, counter):
    per = defaultdict(lambda: [0, 0.0])          # kernel -> [dispatches, sum of counter]
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {dirname}")
    for fn in files:
        with open(fn, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = per[row["Kernel_Name"]]
                k[0] += 1
                k[1] += float(row["Counter_Value"])
    return per


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")          # demangled names: strip these BEFORE cutting at the argument list
    return name.split("(")[0].strip()[:120]


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    program = sys.argv[4] if len(sys.argv) > 4 else "bench.py itself"
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    rows, tot = [], {"launches": 0, "read": 0.0, "write": 0.0}
    for name in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(name, [0, 0.0])
        nw, w = write.get(name, [0, 0.0])
        n = max(nf, nw)
        read_b, write_b = 2.0 * f * 1024.0, w * 1024.0
        rows.append({"kernel": short(name), "launches": n, "read_bytes_per_launch": read_b / max(nf, 1), "write_bytes_per_launch": write_b / max(nw, 1)})
        if "gemm_pp_kernel" in name:
            tot["launches"] += n
            tot["read"] += read_b
            tot["write"] += write_b
    rows.sort(key=lambda r: -(r["read_bytes_per_launch"] + r["write_bytes_per_launch"]) * r["launches"])
    doc = {"kernel": f"gemm_pp_kernel (K6, all epilogue modes), averaged over its launches in {program}",
           "program": program,
           "launches": tot["launches"],
           "traffic_bytes_per_launch": (tot["read"] + tot["write"]) / max(tot["launches"], 1),
           "read_bytes_per_launch": tot["read"] / max(tot["launches"], 1), "write_bytes_per_launch": tot["write"] / max(tot["launches"], 1),
           "how_short": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over {program}, FETCH_SIZE x2 (gfx950), per launch",
           "how": __doc__.strip().splitlines()[0] + "  FETCH_SIZE doubled, KiB -> bytes; counted at the L2<->fabric boundary (Infinity-Cache hits included).",
           "per_kernel": rows[:40]}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps({k: doc[k] for k in ("launches", "traffic_bytes_per_launch", "read_bytes_per_launch", "write_bytes_per_launch")}))


if __name__ == "__main__":
    main()
