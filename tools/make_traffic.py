#!/usr/bin/env python3
"""Turn a pmc_summary.json (tools/pmc_summary.py) into profiles/pmc_traffic.json,
the per-launch HBM traffic of the dominant kernel that bench.py reports.
Usage: make_traffic.py <pmc_summary.json> <kernel-substring> <records> <source-note>"""
import json
import sys


def main():
    summary, substr, records, source = sys.argv[1], sys.argv[2], int(float(sys.argv[3])), sys.argv[4]
    with open(summary) as f:
        s = json.load(f)
    name = next(k for k in s if substr in k)
    fetch, write = s[name]["FETCH_SIZE"], s[name]["WRITE_SIZE"]
    out = {
        "kernel": name,
        "records": records,
        "FETCH_SIZE_KiB": fetch,
        "WRITE_SIZE_KiB": write,
        "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
        "correction": "FETCH_SIZE doubled (gfx950 tallies 128-byte reads as 64 B); both counters are KiB; "
                      "separate --pmc passes",
        "algorithmic_bytes_per_launch": records * 32,
        "source": source,
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
