"""What the SMI library reports on this box for device 0: the raw fields terran_amd.telemetry.SmiCounters reads (energy
accumulator, throttler residencies, accumulation counter), the violation status, and two readings one second of load apart."""
import json
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import affinity, telemetry   # noqa: E402

bdf = affinity.device_cpus(0).get('pci')
print('device 0 pci', bdf, 'hwmon', telemetry.hwmon_of_pci(bdf))
m = telemetry._smi()
print('amdsmi module', m)
c = telemetry.SmiCounters(bdf)
print('handle', c.handle)
if c.handle is not None:
    for fn in ('amdsmi_get_energy_count', 'amdsmi_get_violation_status', 'amdsmi_get_gpu_metrics_header_info', 'amdsmi_get_power_info'):
        try:
            v = getattr(m, fn)(c.handle)
            print(fn, json.dumps({k: (x if not isinstance(x, list) else str(x)[:120]) for k, x in v.items()}, default=str)[:1500])
        except Exception as e:
            print(fn, 'failed:', type(e).__name__, e)
    try:
        g = m.amdsmi_get_gpu_metrics_info(c.handle)
        print('gpu_metrics keys', sorted(g)[:200])
        print({k: g[k] for k in g if 'resid' in k or 'accum' in k or 'throttle' in k or 'energy' in k or k in ('curr_socket_power', 'average_socket_power')})
    except Exception as e:
        print('gpu_metrics failed:', type(e).__name__, e)
    a = c.read()
    # one second of load
    import numpy as np
    from terran_amd import ArcFace, weights
    arc = ArcFace(device=0, state=weights.make_arcface_state())
    crops = np.random.default_rng(0).integers(0, 256, (256, 3, 112, 112), dtype=np.uint8)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        arc.embed_crops(crops)
    b = c.read()
    print('reading a', a)
    print('reading b', b)
    print('delta', telemetry.SmiCounters.delta(a, b))
