"""Socket power, shader clock and temperature of one GPU, read from the amdgpu hwmon files in sysfs.

The 1080p step in `f16x3` runs the chip AT its package power cap: every matrix product is three MFMAs, and the shader
clock the firmware holds under that load (about 1.9 GHz) is what the roofline's 2.4 GHz peak is not.  `bench.py` samples
these sensors over its timed region so the line carries the evidence (`power`), `tools/power_log.py` wraps any command.

`hwmon_dir(device)` follows the device's PCI address (terran_amd.affinity.device_cpus -> ta_device_pci_bus_id) to
/sys/bus/pci/devices/<bdf>/hwmon/hwmon*; everything is None / a no-op when the box exposes no such files.
"""
import glob
import os
import threading
import time


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def hwmon_of_pci(bdf):
    """hwmon directory of the amdgpu function at PCI address `bdf` ('0000:0d:00.0'), or None."""
    if not bdf:
        return None
    for h in sorted(glob.glob('/sys/bus/pci/devices/%s/hwmon/hwmon*' % bdf.lower())):
        try:
            names = os.listdir(h)
        except OSError:
            continue
        if any(n.startswith(('power1_', 'freq1_')) for n in names):
            return h
    return None


def hwmon_dir(device):
    from . import affinity
    return hwmon_of_pci(affinity.device_cpus(device).get('pci'))


def sample(hw):
    """One reading: {'power_w', 'cap_w', 'sclk_mhz', 'mclk_mhz', 'temp_c'} (keys the box does not have are absent)."""
    row = {}
    for key, names in (('power_w', ('power1_average', 'power1_input')), ('cap_w', ('power1_cap',)),
                       ('sclk_mhz', ('freq1_input',)), ('mclk_mhz', ('freq2_input',))):
        for n in names:
            v = _read_int(os.path.join(hw, n))
            if v is not None:
                row[key] = v / 1e6
                break
    temps = [v / 1e3 for v in (_read_int(p) for p in glob.glob(os.path.join(hw, 'temp*_input'))) if v is not None]
    if temps:
        row['temp_c'] = max(temps)
    return row


class PowerSampler:
    """Background thread reading the sensors every `period` seconds between start() and stop()."""

    def __init__(self, hw, period=0.05):
        self.hw = hw
        self.period = period
        self.rows = []
        self._stop = threading.Event()
        self._thread = None

    def start(self):
        if not self.hw or self._thread:
            return self
        self._stop.clear()
        self.rows = []

        def loop():
            t0 = time.perf_counter()
            while not self._stop.is_set():
                r = sample(self.hw)
                r['t'] = time.perf_counter() - t0
                self.rows.append(r)
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=loop, name='ta-power', daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if self._thread:
            self._stop.set()
            self._thread.join(timeout=2.0)          # a sensor read that hangs must not hang the caller (daemon thread)
            self._thread = None
        return list(self.rows)

    def summary(self, skip_seconds=0.0):
        """Means over the samples taken at least `skip_seconds` after start (the sensor averages over a window of its
        own: the first readings of a region still contain what ran before it).  None when nothing was sampled."""
        rows = [r for r in self.rows if r['t'] >= skip_seconds] or self.rows
        if not rows:
            return None
        out = {'samples': len(rows), 'period_s': self.period}
        for key in ('power_w', 'sclk_mhz', 'temp_c'):
            v = [r[key] for r in rows if key in r]
            if v:
                out[key + '_mean'] = round(sum(v) / len(v), 1)
                out[key + '_min'] = round(min(v), 1)
                out[key + '_max'] = round(max(v), 1)
        caps = [r['cap_w'] for r in rows if 'cap_w' in r]
        if caps:
            out['cap_w'] = round(caps[0], 1)
        return out
