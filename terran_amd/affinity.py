"""CPU / NUMA placement of the host threads that feed one GPU.

An MI355X node has two sockets; a rank (or a `StreamPipeline` lane) that stages 1080p batches through pinned memory -- 199 MB
per 32 frames, ~16 GB/s of H2D per GPU at 2 600 frames/s -- should run on the cores of the socket its GPU hangs off, so
that the reader's memcpy into the pinned buffer, the buffer itself (first touch) and the DMA out of it stay on one NUMA
node.  The reference leaves this to the OS (terran/io/video/reader.py:88-117 reads into pageable numpy arrays).

`device_cpus(d)` reads the GPU's PCI address from the library (ta_device_pci_bus_id) and its `numa_node` / `local_cpulist`
from sysfs; `bind(d)` applies them to the CALLING THREAD (Linux affinities are per thread; threads started afterwards inherit
them).  Everything degrades to a no-op when sysfs says nothing (containers without /sys/bus/pci, node -1) or
TERRAN_AMD_NO_AFFINITY is set.
"""
import os

from . import lib

_cache = {}


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def device_cpus(device):
    """-> {'pci': '0000:c1:00.0', 'numa_node': n or None, 'cpus': sorted list or None}"""
    device = int(device)
    if device in _cache:
        return _cache[device]
    info = {'pci': None, 'numa_node': None, 'cpus': None}
    try:
        import ctypes
        buf = ctypes.create_string_buffer(64)
        if lib.load().ta_device_pci_bus_id(device, buf, 64) == lib.OK:
            info['pci'] = buf.value.decode().lower()
            base = '/sys/bus/pci/devices/%s/' % info['pci']
            with open(base + 'numa_node') as f:
                node = int(f.read().strip())
            if node >= 0:
                info['numa_node'] = node
                with open('/sys/devices/system/node/node%d/cpulist' % node) as f:
                    cpus = _parse_cpulist(f.read())
            else:
                with open(base + 'local_cpulist') as f:
                    cpus = _parse_cpulist(f.read())
            allowed = os.sched_getaffinity(0) if hasattr(os, 'sched_getaffinity') else cpus
            cpus &= set(allowed)
            if cpus:
                info['cpus'] = sorted(cpus)
    except (OSError, ValueError, lib.TerranAmdError):
        pass
    _cache[device] = info
    return info


def bind(device):
    """Restrict the calling thread to the CPUs local to `device`.  Returns the placement that was applied (or found)."""
    info = dict(device_cpus(device), bound=False)
    if info['cpus'] and not os.environ.get('TERRAN_AMD_NO_AFFINITY') and hasattr(os, 'sched_setaffinity'):
        try:
            os.sched_setaffinity(0, info['cpus'])
            info['bound'] = True
        except OSError:
            pass
    return info


def describe(info):
    """Short text for logs / bench config: 'pci 0000:c1:00.0 numa 1 cpus 64-127,192-255 (bound)'."""
    cpus = info.get('cpus')
    if cpus:
        runs, start, prev = [], cpus[0], cpus[0]
        for c in cpus[1:] + [None]:
            if c is None or c != prev + 1:
                runs.append('%d-%d' % (start, prev) if prev > start else '%d' % start)
                start = c
            prev = c
        text = ','.join(runs)
    else:
        text = 'unknown'
    return 'pci %s numa %s cpus %s (%s)' % (info.get('pci'), info.get('numa_node'), text,
                                           'bound' if info.get('bound') else 'not bound')
