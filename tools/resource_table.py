#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: one line per kernel."""
import re
import subprocess
import sys

KEYS = [("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("ScratchSize [bytes/lane]", "scratch"), ("VGPRs Spill", "spill"),
        ("Occupancy [waves/SIMD]", "occ"), ("LDS Size [bytes/block]", "lds"), ("SGPRs", "sgpr")]


def main(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, short in KEYS:
            m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
            if m:
                cur[short] = int(m.group(1))
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(rows, names):
        n = re.sub(r"\(tavb::\w+\)$", "", n.replace("tavb::", "")).replace("void ", "")
        print(f"{n:78s} " + " ".join(f"{s}={r.get(s)}" for _, s in KEYS))


if __name__ == "__main__":
    main(sys.argv[1])
