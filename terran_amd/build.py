"""Build libterran_amd.so (hipcc, gfx950 only) in-tree.

`python -m terran_amd.build` or `terran_amd.build.build()`.  hipcc cross-compiles without a
GPU; the resulting .so is git-ignored but travels with the working tree to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libterran_amd.so')
SOURCES = ['runtime.hip', 'conv_igemm.hip', 'layers.hip', 'net.hip', 'retinaface_post.hip', 'arcface_post.hip',
           'openpose_post.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


STAMP = LIB + '.stamp'


def source_hash():
    """sha256 over every kernel source, header and the compile flags: what the built .so must correspond to."""
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.h'))]
    files.append(os.path.join(os.path.dirname(HERE), 'include', 'terran_amd.h'))
    for f in files:
        with open(f, 'rb') as fh:
            h.update(os.path.basename(f).encode() + b'\0' + fh.read())
    return h.hexdigest()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'terran_amd.h'))
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            extra = ['-ffp-contract=off'] if src.endswith('_post.hip') else []   # bit-exact float steps
            cmd = [hipcc] + FLAGS + extra + os.environ.get('TA_EXTRA_FLAGS', '').split() + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, out.decode(errors='replace')))
        if verbose and out:
            print(out.decode(errors='replace'))
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stdout.decode(errors='replace'))
    with open(STAMP, 'w') as fh:                 # lib.load() refuses a binary that does not match the sources beside it
        fh.write(source_hash() + '\n')
    return LIB


def build_probes(force=False):
    """The stand-alone micro-benchmarks under tools/probe (measurement aids, not part of the library)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    pdir = os.path.join(os.path.dirname(HERE), 'tools', 'probe')
    built = []
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not f.endswith('.hip'):
            continue
        src, exe = os.path.join(pdir, f), os.path.join(pdir, f[:-4])
        if force or _stale(exe, [src]):
            r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-Wno-unused-value', '-o', exe, src],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            if r.returncode != 0:
                raise RuntimeError('hipcc failed on %s:\n%s' % (f, r.stdout.decode(errors='replace')))
        built.append(exe)
    return built


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
    print(build_probes(force='--force' in sys.argv))
