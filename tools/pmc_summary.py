#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE summaries of tools/pmc.sh -> profiles/pmc_<family>.json (what bench.py reads back as
`roofline.traffic`), stamped with the fingerprint of the kernel sources the passes ran on.

    python tools/pmc_summary.py FAMILY KERNEL_REGEX FETCH.txt WRITE.txt --batch 1024 --dtype bf16 [--out profiles/pmc_FAMILY.json]

Means are over every launch of every kernel whose name matches KERNEL_REGEX (KB per launch, as rocprofv3 reports)."""
import argparse
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mean_kb(path, regex, counter):
    calls, total = 0, 0.0
    for line in open(path):
        parts = line.split()
        if len(parts) < 6 or parts[4] != counter or not re.search(regex, " ".join(parts[5:])):
            continue
        calls += int(parts[0])
        total += float(parts[2])
    if calls == 0:
        raise SystemExit(f"{path}: no {counter} rows match {regex!r}")
    return calls, total / calls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("family")
    ap.add_argument("regex")
    ap.add_argument("fetch")
    ap.add_argument("write")
    ap.add_argument("--batch", type=int, required=True)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--out", default=None)
    ap.add_argument("--source", default=None)
    args = ap.parse_args()
    from bench import kernel_sources_sha16
    n_f, fetch = mean_kb(args.fetch, args.regex, "FETCH_SIZE")
    n_w, write = mean_kb(args.write, args.regex, "WRITE_SIZE")
    out = {"kernel": args.family, "batch_per_gpu": args.batch, "dtype": args.dtype, "launches": n_f,
           "fetch_kb_per_launch": round(fetch, 1), "write_kb_per_launch": round(write, 1),
           "kernel_sources_sha16": kernel_sources_sha16(),
           "source": args.source or f"{args.fetch} + {args.write}: tools/pmc.sh (rocprofv3 --kernel-trace --pmc <counter> "
                                    "--kernel-include-regex grafp, separate passes; mean over every launch of the family; "
                                    "FETCH_SIZE is doubled by bench.py per MI355X_MICROARCH.md HBM section)"}
    path = args.out or os.path.join("profiles", f"pmc_{args.family}.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
