#!/usr/bin/env python3
"""GPU: runs tools/probes/mfma_mix_probe (register-resident MFMA loops, see its header) for every arithmetic mix and samples
clock / power meanwhile.  Build first:  hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_mix_probe.hip -o tools/probes/mfma_mix_probe

    python tools/mfma_mix.py [--seconds 4]     ->  gpurun_out/mfma_mix.json + a table on stdout
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from clock_power import smi_row  # noqa: E402

NAMES = {0: "bf16 x3 (36 MFMAs per K=64 block of 3 tiles)", 1: "f16 + 2 x fp8 e4m3 (12 + 6)", 2: "bf16 x1 (12)",
         3: "fp8 only (6)", 4: "f16 + 2 x fp6 e2m3 (12 + 6)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    a = ap.parse_args()
    exe = os.path.join(ROOT, "tools", "probes", "mfma_mix_probe")
    out = {"tool": "tools/mfma_mix.py", "modes": {}}
    chk = subprocess.run([exe, "check"], capture_output=True, text=True)
    out["check"] = {"rc": chk.returncode, "lines": chk.stdout.strip().splitlines(), "stderr": chk.stderr[-500:]}
    print(chk.stdout, chk.stderr[-300:])
    for mode in (0, 1, 2, 3, 4, 0, 1):
        samples, stop = [], threading.Event()

        def sampler():
            while not stop.is_set():
                samples.append(smi_row())
                time.sleep(0.1)
        th = threading.Thread(target=sampler, daemon=True)
        th.start()
        p = subprocess.run([exe, str(mode), str(a.seconds)], capture_output=True, text=True)
        stop.set()
        th.join(timeout=10)
        res = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
        rec = {"name": NAMES[mode], "rc": p.returncode, "launches": [ln.strip() for ln in p.stdout.splitlines() if "launch" in ln]}
        if res:
            f = res[0].split()
            rec["ns_per_block_first"] = float(f[4])
            rec["ns_per_block_settled"] = float(f[6])
        loaded = sorted([s for s in samples if s.get("power_w")], key=lambda s: -s["power_w"])
        loaded = loaded[:max(1, len(loaded) // 2)]
        if loaded:
            rec["power_w"] = round(sum(s["power_w"] for s in loaded) / len(loaded), 1)
            rec["sclk_mhz"] = round(sum(s.get("sclk_mhz", 0) for s in loaded) / len(loaded), 1)
        key = f"mode{mode}" + ("_again" if f"mode{mode}" in out["modes"] else "")
        out["modes"][key] = rec
        print(key, {k: v for k, v in rec.items() if k != "launches"}, flush=True)
    base = out["modes"]["mode0"].get("ns_per_block_settled")
    if base:
        for k, r in out["modes"].items():
            if r.get("ns_per_block_settled"):
                r["speed_vs_bf16x3"] = round(base / r["ns_per_block_settled"], 3)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "mfma_mix.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: (r.get("ns_per_block_settled"), r.get("speed_vs_bf16x3"), r.get("power_w"), r.get("sclk_mhz"))
                      for k, r in out["modes"].items()}))


if __name__ == "__main__":
    main()
