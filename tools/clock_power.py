#!/usr/bin/env python3
"""GPU: samples the shader clock, the socket power and the temperature (rocm-smi, a few samples per second) for some
seconds of idle, while the normal bench runs its timed rounds, and after it.  What it answers: is the residual tower
running at the clock the nominal MFMA peak assumes (2.4 GHz), or at a power- / thermally-limited one?

    python tools/clock_power.py [--steps 400] [--config normal]      ->  gpurun_out/clock_power.json + a table on stdout
"""
import argparse
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def smi_row():
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--json"],
                             capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = d[next(iter(d))]
    except Exception as e:                                   # noqa: BLE001
        return {"error": repr(e)}
    row = {}
    for k, v in card.items():
        kl = k.lower()
        m = re.search(r"([-+]?\d+(\.\d+)?)", str(v))
        if not m:
            continue
        x = float(m.group(1))
        if "sclk" in kl or "mclk" in kl:
            if "mhz" in str(v).lower():
                row["sclk_mhz" if "sclk" in kl else "mclk_mhz"] = x
        elif "power" in kl and "socket" in kl or "average graphics package power" in kl or kl.startswith("current socket"):
            row["power_w"] = x
        elif "temperature" in kl and ("hotspot" in kl or "junction" in kl):
            row["temp_hotspot_c"] = x
        elif "temperature" in kl and "mem" in kl:
            row["temp_mem_c"] = x
    if not smi_row.raw:
        smi_row.raw.update(card)
    return row


smi_row.raw = {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--config", default="normal")
    ap.add_argument("--idle", type=float, default=3.0)
    a = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    samples, stop, t0 = [], threading.Event(), time.time()

    def sampler():
        while not stop.is_set():
            r = smi_row()
            r["t"] = round(time.time() - t0, 2)
            samples.append(r)
            time.sleep(0.1)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    time.sleep(a.idle)
    marks = {"bench_start": round(time.time() - t0, 2)}
    env = dict(os.environ, TMPDIR="/tmp")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", a.config, "--steps", str(a.steps),
                        "--warmup", "10", "--sustained-rounds", "0", "--no-micro", "--no-cpu-baseline", "--no-other-configs",
                        "--no-dist"], capture_output=True, text=True, env=env, cwd=ROOT)
    marks["bench_end"] = round(time.time() - t0, 2)
    time.sleep(a.idle)
    stop.set()
    th.join(timeout=15)
    line = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 and p.stdout.strip() else None
    timed_s = line["ms_per_step"] * line["steps"] / 1e3 if line else 0.0
    # the timed rounds are the last `timed_s` seconds before the bench process prints and tears down (~1-2 s)
    out = {"tool": "tools/clock_power.py", "config": a.config, "steps": a.steps, "marks": marks,
           "bench": None if not line else {"value": line["value"], "ms_per_step": line["ms_per_step"],
                                           "avg_block_launch_ms": line["roofline"].get("avg_launch_ms")},
           "timed_seconds": timed_s, "first_raw_sample": smi_row.raw, "samples": samples}
    busy = [s for s in samples if s.get("power_w") and marks["bench_start"] < s["t"] < marks["bench_end"]]
    if busy:
        top = sorted(busy, key=lambda s: -s["power_w"])[:max(3, len(busy) // 3)]      # the loaded third of the window
        out["under_load"] = {k: round(sum(s.get(k, 0.0) for s in top) / len(top), 1)
                             for k in ("sclk_mhz", "power_w", "temp_hotspot_c")}
        out["under_load"]["samples"] = len(top)
    idle = [s for s in samples if s.get("power_w") and s["t"] < marks["bench_start"]]
    if idle:
        out["idle"] = {k: round(sum(s.get(k, 0.0) for s in idle) / len(idle), 1) for k in ("sclk_mhz", "power_w", "temp_hotspot_c")}
    with open(os.path.join(ROOT, "gpurun_out", "clock_power.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out.get(k) for k in ("bench", "idle", "under_load", "marks", "timed_seconds")}))
    for s in samples[:: max(1, len(samples) // 40)]:
        print(s)


if __name__ == "__main__":
    main()
