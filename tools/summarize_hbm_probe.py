#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on known byte counts (tools/probes/hbm_counter_probe.hip):

    python tools/summarize_hbm_probe.py gpurun_out/hbm_probe > profiles/r06_hbm_counter_calibration.json

reads <dir>/probe.json (the probe's own stdout: bytes each kernel touches) and the counter_collection.csv files under
<dir>/fetch and <dir>/write (separate --pmc passes), and prints counter bytes (KiB x 1024, no correction) / true bytes per
kernel: the factor a measured counter has to be DIVIDED by for that access pattern."""
import collections
import csv
import glob
import json
import os
import sys


def counters(d):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"].split("(")[0].strip()
                out[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return out


def main():
    src = sys.argv[1]
    truth = json.loads(open(os.path.join(src, "probe.json")).read().strip().splitlines()[-1])
    rows = {}
    for sub, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for (k, c), v in counters(os.path.join(src, sub)).items():
            if c != cname:
                continue
            key = next((t for t in truth if isinstance(truth[t], dict) and k.replace("void ", "").startswith(t.split("<")[0])
                        and (("<" not in t) or t.split("<")[1].rstrip(">") in k)), None)
            if key is None:
                continue
            v = v[1:] if len(v) > 1 else v                     # (the first launch of a kernel also faults its pages in)
            mean = sum(v) / len(v) * 1024.0
            e = rows.setdefault(key, dict(truth[key]))
            e[cname + "_bytes_raw"] = mean
            want = truth[key].get("read" if cname == "FETCH_SIZE" else "write")
            if want:
                e[cname + "_over_true"] = mean / want
    print(json.dumps({"boards": truth["boards"], "kernels": rows,
                      "note": "raw = counter KiB x 1024 with no correction; *_over_true = raw / bytes the kernel really "
                              "reads (FETCH) or writes (WRITE); span = bytes of the address range a write pattern with holes covers"},
                     indent=1))


if __name__ == "__main__":
    main()
