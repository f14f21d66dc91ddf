#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, the mean counter value per
dispatch.  Usage: pmc_summary.py <dir-with-*counter_collection.csv> [...]
Prints one JSON object {kernel: {counter: mean_per_dispatch, "dispatches": n}}."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for root in sys.argv[1:]:
        for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    name = row.get("Kernel_Name", "?")
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
        out[k]["dispatches"] = max(len(v) for v in cs.values())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
