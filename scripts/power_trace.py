#!/usr/bin/env python
"""Samples socket power, shader clock and temperature of GPU 0 at ~20 Hz while a command runs, and writes the trace as JSON
(is the pipeline at its power cap during the ViT?).  Reads the amdgpu hwmon / pp_dpm files directly (a rocm-smi
process per sample is too slow for 10 Hz); falls back to `rocm-smi --showpower --showclocks --json` if sysfs is not there.
Usage: python scripts/power_trace.py OUT.json -- <command ...>"""
import glob
import json
import os
import subprocess
import sys
import threading
import time


def all_cards():
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
        if hw and os.path.exists(os.path.join(card, "pp_dpm_sclk")):
            out.append((card, hw[0]))
    return out


def find_sysfs():
    """The card whose power rises under a short GPU load (a box may expose the sysfs files of GPUs this container cannot use: the first call of
    round 5 happened to read the right card, the second read an idle neighbour at 94 MHz).  Falls back to the first card."""
    cards = all_cards()
    if len(cards) <= 1:
        return cards[0] if cards else (None, None)
    try:
        import torch
        idle = [sample_sysfs(c, h).get("power_in_uW", sample_sysfs(c, h).get("power_uW", 0)) for c, h in cards]
        x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
        t0 = time.time()
        busy = [0] * len(cards)
        while time.time() - t0 < 1.5:
            for _ in range(20):
                x @ x
            torch.cuda.synchronize()
            for i, (c, h) in enumerate(cards):
                s = sample_sysfs(c, h)
                busy[i] = max(busy[i], s.get("power_in_uW", s.get("power_uW", 0)))
        best = max(range(len(cards)), key=lambda i: busy[i] - idle[i])
        return cards[best]
    except Exception:  # noqa: BLE001
        return cards[0]


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def sample_sysfs(card, hw):
    s = {}
    for key, name in (("power_uW", "power1_average"), ("power_in_uW", "power1_input"), ("sclk_hz", "freq1_input"), ("mclk_hz", "freq2_input"),
                      ("temp_mC", "temp1_input"), ("cap_uW", "power1_cap")):
        v = read(os.path.join(hw, name))
        if v is not None:
            try:
                s[key] = int(v)
            except ValueError:
                pass
    dpm = read(os.path.join(card, "pp_dpm_sclk"))
    if dpm:
        for line in dpm.splitlines():
            if line.rstrip().endswith("*"):
                s["dpm_sclk"] = line.strip()
    busy = read(os.path.join(card, "gpu_busy_percent"))
    if busy is not None:
        s["busy"] = busy
    return s


def sample_smi():
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        return {"smi": json.loads(out)}
    except Exception as e:  # noqa: BLE001
        return {"smi_error": str(e)}


def main():
    out, cmd = sys.argv[1], sys.argv[sys.argv.index("--") + 1:]
    card, hw = find_sysfs()
    samples, stop = [], threading.Event()

    def loop():
        while not stop.is_set():
            t = time.time()
            s = sample_sysfs(card, hw) if card else sample_smi()
            s["t"] = t
            samples.append(s)
            stop.wait(0.05 if card else 0.5)

    th = threading.Thread(target=loop, daemon=True)
    t0 = time.time()
    th.start()
    rc = subprocess.call(cmd)
    stop.set()
    th.join()
    json.dump({"cmd": cmd, "rc": rc, "t0": t0, "t1": time.time(), "source": "sysfs" if card else "rocm-smi", "card": card, "samples": samples}, open(out, "w"))
    pw = [s.get("power_uW", s.get("power_in_uW")) for s in samples if s.get("power_uW") or s.get("power_in_uW")]
    ck = [s["sclk_hz"] for s in samples if s.get("sclk_hz")]
    if pw:
        print(f"power_trace: {len(samples)} samples, power W min/mean/max {min(pw)/1e6:.0f}/{sum(pw)/len(pw)/1e6:.0f}/{max(pw)/1e6:.0f}, "
              + (f"sclk MHz min/mean/max {min(ck)/1e6:.0f}/{sum(ck)/len(ck)/1e6:.0f}/{max(ck)/1e6:.0f}" if ck else "no sclk"))
    else:
        print(f"power_trace: {len(samples)} samples from {'sysfs' if card else 'rocm-smi'} (no power field parsed)")
    sys.exit(rc)


if __name__ == "__main__":
    main()
