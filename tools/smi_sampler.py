#!/usr/bin/env python3
"""Clock / power sampler for the warm-up transient probe (tools/transient_probe.sh): one line per sample
`unix_time sclk_MHz mclk_MHz power_W` read from sysfs (no HIP call: safe to run beside the bench as its own process)."""
import glob
import sys
import time


def first(pattern):
    g = sorted(glob.glob(pattern))
    return g[0] if g else None


def cur_level(path):
    try:
        for ln in open(path):
            if "*" in ln:
                return ln.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", "")
    except OSError:
        pass
    return "nan"


def read(path):
    try:
        return open(path).read().strip()
    except (OSError, TypeError):
        return "nan"


def main():
    period = float(sys.argv[1]) if len(sys.argv) > 1 else 0.002
    dev = None
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if glob.glob(d + "/pp_dpm_sclk"):
            dev = d
            break
    if dev is None:
        print("# no amdgpu sysfs node with pp_dpm_sclk", flush=True)
        return
    power = first(dev + "/hwmon/hwmon*/power1_average") or first(dev + "/hwmon/hwmon*/power1_input")
    freq = first(dev + "/hwmon/hwmon*/freq1_input")
    print(f"# dev {dev} power {power} freq {freq}", flush=True)
    while True:
        t = time.time()
        s = read(freq) if freq else cur_level(dev + "/pp_dpm_sclk")
        if freq and s != "nan":
            s = str(int(s) // 1000000)
        m = cur_level(dev + "/pp_dpm_mclk")
        p = read(power)
        if p != "nan":
            p = str(int(p) // 1000000)
        print(f"{t:.4f} {s} {m} {p}", flush=True)
        time.sleep(period)


if __name__ == "__main__":
    main()
