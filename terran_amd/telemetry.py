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


# -- accumulating counters through the SMI library -------------------------------------------------------------------------
# hwmon's power1_average is the firmware's moving average of the socket power.  Two things it cannot say: how many JOULES a
# region cost (the figure that decides between kernel variants once the step sits at the power cap: DESIGN section 4), and
# WHY the clock is where it is.  The firmware keeps accumulators for both (gpu_metrics: `energy_accumulator`, and per throttler
# a residency counter that advances, in units of `accumulation_counter`, while that limiter holds the clock down: PPT = package
# power tracking, socket / VR / HBM thermal, PROCHOT).  libamd_smi reads them; its Python package ships with ROCm under
# /opt/rocm/share/amd_smi (not on sys.path by default).  Everything here returns None when the library, the driver or a
# field is missing.
_SMI = {}
_RESIDENCIES = ('ppt_residency_acc', 'socket_thm_residency_acc', 'vr_thm_residency_acc', 'hbm_thm_residency_acc', 'prochot_residency_acc')


def _smi():
    if 'mod' not in _SMI:
        _SMI['mod'] = None
        try:
            import sys
            for p in (os.environ.get('ROCM_PATH', '/opt/rocm') + '/share/amd_smi',):
                if os.path.isdir(p) and p not in sys.path:
                    sys.path.append(p)
            import amdsmi
            amdsmi.amdsmi_init()
            _SMI['mod'] = amdsmi
        except Exception:                      # no library / no driver / no permission: telemetry is optional
            _SMI['mod'] = None
    return _SMI['mod']


class SmiCounters:
    """Energy and throttler-residency accumulators of the GPU at PCI address `bdf`: `read()` -> one reading (dict),
    `SmiCounters.delta(a, b)` -> what a region between two readings cost."""

    def __init__(self, bdf):
        self.handle = None
        m = _smi() if bdf else None
        if m is None:
            return
        try:
            for h in m.amdsmi_get_processor_handles():
                if str(m.amdsmi_get_gpu_device_bdf(h)).lower() == bdf.lower():
                    self.handle = h
                    break
        except Exception:
            self.handle = None

    def read(self):
        if self.handle is None:
            return None
        m = _smi()
        row = {'t': time.perf_counter()}
        try:
            e = m.amdsmi_get_energy_count(self.handle)
            acc = e.get('energy_accumulator', e.get('power'))
            row['energy_uj'] = float(acc) * float(e.get('counter_resolution', 15.259))
        except Exception:
            pass
        try:
            g = m.amdsmi_get_gpu_metrics_info(self.handle)
            for k in _RESIDENCIES + ('accumulation_counter', 'throttle_status', 'indep_throttle_status'):
                v = g.get(k)
                if isinstance(v, int) and v not in (0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF):
                    row[k] = v
        except Exception:
            pass
        return row if len(row) > 1 else None

    @staticmethod
    def delta(a, b):
        """{'seconds', 'energy_j', 'power_w_from_energy', 'throttle_residency_pct': {limiter: % of the region it was active},
        'throttle_status'} between two readings (keys the box does not report are absent); None without readings."""
        if not a or not b:
            return None
        out = {'seconds': round(b['t'] - a['t'], 4)}
        if 'energy_uj' in a and 'energy_uj' in b and b['energy_uj'] >= a['energy_uj']:
            out['energy_j'] = round((b['energy_uj'] - a['energy_uj']) * 1e-6, 2)
            if out['seconds'] > 0:
                out['power_w_from_energy'] = round(out['energy_j'] / out['seconds'], 1)
        ticks = b.get('accumulation_counter', 0) - a.get('accumulation_counter', 0)
        if ticks > 0:
            out['throttle_residency_pct'] = {k[:-len('_residency_acc')]: round(100.0 * (b[k] - a[k]) / ticks, 2)
                                             for k in _RESIDENCIES if k in a and k in b}
        for k in ('throttle_status', 'indep_throttle_status'):
            if k in b:
                out[k] = b[k]
        return out if len(out) > 1 else None


class PowerSampler:
    """Background thread reading the sensors every `period` seconds between start() and stop(); with `bdf` (the GPU's PCI
    address) the SMI accumulators are read at start() and stop() as well (`region`: energy, throttler residencies)."""

    def __init__(self, hw, period=0.05, bdf=None):
        self.hw = hw
        self.period = period
        self.rows = []
        self._stop = threading.Event()
        self._thread = None
        self.smi = SmiCounters(bdf) if bdf else None
        self._smi0 = None
        self.region = None

    def start(self):
        if self.smi is not None:
            self._smi0 = self.smi.read()
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
        if self.smi is not None and self._smi0 is not None:
            self.region = SmiCounters.delta(self._smi0, self.smi.read())
            self._smi0 = None
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
            return {'region': self.region} if self.region else None
        out = {'samples': len(rows), 'period_s': self.period}
        if self.region:
            out['region'] = self.region             # the WHOLE region between start() and stop() (no skip): accumulators
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
