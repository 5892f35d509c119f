"""Register / LDS / spill figures of every kernel of a hipcc object file (the gfx950 code object inside its .hip_fatbin section):
    python tools/kernel_resources.py terran_amd/csrc/conv_igemm.o [name filter]"""
import re
import struct
import subprocess
import sys
import tempfile

obj = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
data = open(obj, 'rb').read()
i = data.find(b'__CLANG_OFFLOAD_BUNDLE__')
assert i >= 0, 'no offload bundle'
n = struct.unpack_from('<Q', data, i + 24)[0]
o = i + 32
co = None
for _ in range(n):
    off, size, tl = struct.unpack_from('<QQQ', data, o)
    triple = data[o + 24:o + 24 + tl].decode()
    o += 24 + tl
    if 'gfx950' in triple:
        co = data[i + off:i + off + size]
assert co, 'no gfx950 code object'
with tempfile.NamedTemporaryFile(suffix='.co') as f:
    f.write(co)
    f.flush()
    txt = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', f.name], stdout=subprocess.PIPE).stdout.decode()
for blk in txt.split('- .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    name = g('name')
    if flt in name:
        print('%-90s vgpr %s agpr %s sgpr %s spill_v %s lds %s' % (name[:90], g('vgpr_count'), blk.split()[0], g('sgpr_count'), g('vgpr_spill_count'), g('group_segment_fixed_size')))
