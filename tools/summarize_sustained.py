#!/usr/bin/env python3
"""Per-kernel durations of the search-round launches over windows of a long traced run (tools/profile_sustained.sh):
mean / p50 / p99 / max per kernel for the first 200 rounds and for the last 1000."""
import csv
import glob
import json
import os
import sys

src = sys.argv[1]
f = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
names = ("k_noise", "k_sim", "k_advance", "k_resblock", "k_input_conv")
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        n = next((k for k in names if k in r["Kernel_Name"]), None)
        if n:
            rows.append((int(r["Start_Timestamp"]), n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
# label the search launches of a round: k_sim(BACKUP), k_advance, k_noise, k_sim(SELECT)   (round 3: one k_noise launch)
seq = [(n, d) for _, n, d in rows if n in ("k_noise", "k_sim", "k_advance")]
labels = ["k_sim(BACKUP)", "k_advance", "k_noise", "k_sim(SELECT)"]
L = len(labels)
first = next((i for i, x in enumerate(seq) if x[0] == "k_sim"), 0)
seq = seq[first:]
rounds = [seq[i:i + L] for i in range(0, len(seq) - len(seq) % L, L)]
rounds = [r for r in rounds if [x[0] for x in r] == ["k_sim", "k_advance", "k_noise", "k_sim"]]


def stats(v):
    v = sorted(v)
    return {"mean": sum(v) / len(v), "p50": v[len(v) // 2], "p99": v[int(len(v) * 0.99)], "max": v[-1], "n": len(v)}


out = {"rounds_traced": len(rounds), "unit": "us"}
for name, sl in (("first_200_rounds", rounds[24:224]), ("last_1000_rounds", rounds[-1000:])):
    out[name] = {labels[j]: stats([r[j][1] for r in sl]) for j in range(L)}
    out[name]["sum_of_means"] = sum(out[name][labels[j]]["mean"] for j in range(L))
rb = [d for _, n, d in rows if n == "k_resblock"]
if rb:
    out["k_resblock_last_7000"] = stats(rb[-7000:])
print(json.dumps(out, indent=1))
