#!/usr/bin/env python3
"""Average PMC counter values per dispatch of one kernel from rocprofv3 --pmc csv output.
Usage: python tools/pmc_summary.py <kernel-name-substring> <dir> [<dir> ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    pat = sys.argv[1]
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc, n = defaultdict(float), defaultdict(int)
            for row in csv.DictReader(open(f)):
                if pat not in row["Kernel_Name"]:
                    continue
                acc[row["Counter_Name"]] += float(row["Counter_Value"])
                n[row["Counter_Name"]] += 1
            for k in sorted(acc):
                print("%-28s %16.0f  (avg of %d dispatches)" % (k, acc[k] / n[k], n[k]))


if __name__ == "__main__":
    main()
