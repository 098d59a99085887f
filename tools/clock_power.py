#!/usr/bin/env python
"""Shader clock and package power under each kind of launch of the path (GPU box, from the repo root):

    python tools/clock_power.py [seconds per workload]

For each workload a loop of launches runs for a few seconds while a thread samples the GPU's hwmon files (sclk, package power; rocm-smi
where sysfs does not show them).  What it answers: does the materialisation kernel (float64 arithmetic AND 4.4 TB/s of stores) run at the
clock the fused render gets, or does the package limit (1400 W) take the clock down when HBM writes are added to full float64 issue?
"""
import glob
import os
import statistics
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _find_hwmon(pci=None):
    """hwmon files of the GPU with PCI bus id `pci` (the box shows several cards in sysfs; one is ours)."""
    cands = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        dev = os.path.realpath(os.path.join(d, "..", ".."))
        p = [os.path.join(d, n) for n in ("power1_average", "power1_input")]
        p = [x for x in p if os.path.exists(x)]
        f = os.path.join(d, "freq1_input")
        if p and os.path.exists(f):
            cands.append((dev, p[0], f))
    for dev, p, f in cands:
        if pci and dev.lower().endswith(pci.lower()):
            return p, f
    print("no hwmon for %s among %s" % (pci, [c[0] for c in cands]))
    return None, None


class Sampler(threading.Thread):
    def __init__(self, period=0.1, pci=None):
        super().__init__(daemon=True)
        self.period, self.rows, self.stop_flag = period, [], False
        self.power_file, self.freq_file = _find_hwmon(pci)

    def sample(self):
        if self.power_file:
            try:
                return float(open(self.freq_file).read()) / 1e6, float(open(self.power_file).read()) / 1e6
            except (OSError, ValueError):
                pass
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=10).stdout
            import re
            w = re.search(r"([0-9.]+)\s*$", out.splitlines()[1].split(",")[-1])
            m = re.search(r"\((\d+)Mhz\)", out)
            return (float(m.group(1)) if m else float("nan")), (float(w.group(1)) if w else float("nan"))
        except Exception:
            return float("nan"), float("nan")

    def run(self):
        while not self.stop_flag:
            self.rows.append((time.perf_counter(),) + self.sample())
            time.sleep(self.period)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    from synthesizer_amd import _native as N
    from synthesizer_amd import oscillators as G
    from synthesizer_amd import workloads as W
    from synthesizer_amd.mixer import VoiceBank
    N.ensure_init(0)
    SR, NV, F2 = 48000, 1024, 480000
    v, g = W.additive_voices(G, NV, SR, seed=0, partials=16, adsr={"sustain": 1.0e6})
    bank = VoiceBank(v, gains=g)
    vbuf = N.DeviceBuffer(NV * F2 * 4)
    bus = N.DeviceBuffer(F2 * 8)
    ring = [N.DeviceBuffer(SR * 8) for _ in range(4)]
    rows16 = vbuf.view(0, NV * F2 * 2)
    pos = [5]

    def render():
        bank.render_device(SR, pos[0] * SR, bus_f32=ring[pos[0] & 3])
        pos[0] += 1
    work = [
        ("fused render (headline), 1024 x 48000 per launch", render, NV * SR),
        ("generate float32 rows 1024 x 480000", lambda: bank.generate_device(F2, 5 * SR, out=vbuf), NV * F2),
        ("generate int16 rows 1024 x 480000", lambda: bank.generate_i16_device(F2, 5 * SR, out=rows16, stride=F2, check=False), NV * F2),
        ("mix float32 rows -> bus (HBM read)", lambda: bank.mix_device(vbuf, F2, bus_f32=bus), NV * F2),
        ("fill 1.97 GB (hipMemsetAsync)", lambda: vbuf.zero(), NV * F2),
    ]
    s = Sampler(pci=N.device_pci())
    print("hwmon: %s %s" % (s.power_file, s.freq_file))
    s.start()
    time.sleep(1.0)
    marks = []
    for name, call, units in work:
        call()
        N.sync()
        t0 = time.perf_counter()
        n = 0
        N.timer_start()
        while time.perf_counter() - t0 < secs:
            for _ in range(8):
                call()
            n += 8
            if n % 64 == 0:
                N.sync()
        ms = N.timer_stop() / n
        t1 = time.perf_counter()
        marks.append((name, t0, t1, ms, units))
        time.sleep(1.5)
    s.stop_flag = True
    s.join()
    try:
        bank.overflow_check()
    except Exception as e:
        print("overflow:", e)
    idle = [r for r in s.rows if r[0] < marks[0][1] - 0.2]
    print("idle: sclk %.0f MHz, package %.0f W" % (statistics.median(r[1] for r in idle), statistics.median(r[2] for r in idle)))
    print("%-52s %10s %10s %10s %12s" % ("workload", "ms/launch", "sclk MHz", "package W", "T units/s"))
    for name, t0, t1, ms, units in marks:
        rows = [r for r in s.rows if t0 + 0.6 < r[0] < t1 - 0.1]
        clk = statistics.median(r[1] for r in rows) if rows else float("nan")
        pw = statistics.median(r[2] for r in rows) if rows else float("nan")
        print("%-52s %10.4f %10.0f %10.0f %12.4f   (%d samples; clk %.0f..%.0f, W %.0f..%.0f)" % (
            name, ms, clk, pw, units / (ms / 1e3) / 1e12, len(rows),
            min(r[1] for r in rows) if rows else 0, max(r[1] for r in rows) if rows else 0,
            min(r[2] for r in rows) if rows else 0, max(r[2] for r in rows) if rows else 0))


if __name__ == "__main__":
    main()
