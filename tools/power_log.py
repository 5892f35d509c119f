"""Sample the GPU's power, shader clock and temperature (amdgpu sysfs / hwmon) while a command runs.

    python tools/power_log.py --out gpurun_out/power.txt -- python bench.py --single-mode

Evidence for DESIGN section 4's "the pipelined step is held by the power budget, not by issue slots": the socket power
during the timed region against its cap, and the shader clock the chip holds there against the 2.4 GHz the peak figures
are quoted at.  Reads whatever of these files the box has (none -> says so and just runs the command):
  hwmon*/power1_average | power1_input (uW), power1_cap (uW), freq1_input (Hz, sclk), freq2_input (Hz, mclk),
  temp*_input (m degC), and rocm-smi as a fallback for power / sclk when hwmon has none.
"""
import argparse
import os
import subprocess
import sys


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import telemetry   # noqa: E402


def hip_bus_id(device=0):
    """PCI address of HIP device `device` as this process tree sees it (asked in a child: the sampler holds no GPU context)."""
    code = ("import ctypes; h = ctypes.CDLL('libamdhip64.so'); b = ctypes.create_string_buffer(64); "
            "rc = h.hipDeviceGetPCIBusId(b, 64, %d); print(b.value.decode() if rc == 0 else '')" % device)
    try:
        return subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120).stdout.strip().lower()
    except Exception:
        return ''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--period', type=float, default=0.05)
    ap.add_argument('--device', type=int, default=0, help='HIP device index whose sensors to read')
    ap.add_argument('cmd', nargs=argparse.REMAINDER)
    args = ap.parse_args()
    cmd = args.cmd[1:] if args.cmd and args.cmd[0] == '--' else args.cmd
    bdf = hip_bus_id(args.device)
    hw = telemetry.hwmon_of_pci(bdf)
    card = '/sys/bus/pci/devices/%s' % bdf if hw else None
    sampler = telemetry.PowerSampler(hw, args.period).start()
    rc = subprocess.call(cmd)
    rows = sampler.stop()
    lines = ['# %s' % ' '.join(cmd), '# HIP device %d = pci %s -> %s, %s' % (args.device, bdf or '?', card, hw)]
    if not hw:
        lines.append('# no amdgpu hwmon files on this box: nothing sampled')
    for key in ('power_w', 'sclk_mhz', 'mclk_mhz', 'temp_c'):
        vals = [r[key] for r in rows if key in r]
        if not vals:
            continue
        s = sorted(vals)
        # "loaded" = samples in the upper half of the power range: the timed region, not import / packing / the CPU baseline
        lines.append('%-9s n %5d  min %9.1f  median %9.1f  p90 %9.1f  max %9.1f' %
                     (key, len(s), s[0], s[len(s) // 2], s[int(0.9 * (len(s) - 1))], s[-1]))
    caps = [r['cap_w'] for r in rows if 'cap_w' in r]
    if caps:
        lines.append('cap_w     %.1f' % caps[0])
    pw = [r.get('power_w') for r in rows]
    if any(p is not None for p in pw):
        hi = max(p for p in pw if p is not None)
        lo = min(p for p in pw if p is not None)
        loaded = [r for r in rows if r.get('power_w') is not None and r['power_w'] >= lo + 0.6 * (hi - lo)]
        if loaded:
            lines.append('# samples with power >= idle + 0.6 (max - idle): %d (%.1f s)' % (len(loaded), len(loaded) * args.period))
            for key in ('power_w', 'sclk_mhz', 'temp_c'):
                v = [r[key] for r in loaded if key in r]
                if v:
                    lines.append('loaded %-9s mean %9.1f  min %9.1f  max %9.1f' % (key, sum(v) / len(v), min(v), max(v)))
    lines.append('# time series (every 4th sample): t_s power_w sclk_mhz temp_c')
    for r in rows[::4]:
        lines.append('%8.2f %8.1f %8.1f %6.1f' % (r['t'], r.get('power_w', float('nan')), r.get('sclk_mhz', float('nan')),
                                                   r.get('temp_c', float('nan'))))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:14]), file=sys.stderr)
    sys.exit(rc)


if __name__ == '__main__':
    main()
