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


def _unit_hash(src, headers, flags):
    """What one object file must correspond to: its source, every header and the exact flags."""
    import hashlib
    h = hashlib.sha256(' '.join(flags).encode())
    for f in [src] + sorted(headers):
        with open(f, 'rb') as fh:
            h.update(os.path.basename(f).encode() + b'\0' + fh.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Rebuilds are driven by CONTENT hashes, not mtimes: every object carries a `.o.stamp` with the hash of its source,
    all headers and its flags (a `git checkout` of older sources, or changed TA_EXTRA_FLAGS, recompiles it), and the
    library's stamp is written only for a link of objects that all match the sources beside them."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'terran_amd.h'))
    extra_all = os.environ.get('TA_EXTRA_FLAGS', '').split()
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        flags = FLAGS + (['-ffp-contract=off'] if src.endswith('_post.hip') else []) + extra_all   # _post: bit-exact float steps
        want = _unit_hash(s, headers, flags)
        try:
            with open(o + '.stamp') as fh:
                have = fh.read().strip()
        except OSError:
            have = None
        if force or have != want or not os.path.exists(o):
            cmd = [hipcc] + flags + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            if os.path.exists(o + '.stamp'):
                os.remove(o + '.stamp')
            procs.append((src, o, want, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, o, want, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, out.decode(errors='replace')))
        with open(o + '.stamp', 'w') as fh:
            fh.write(want + '\n')
        if verbose and out:
            print(out.decode(errors='replace'))
    try:
        with open(STAMP) as fh:
            lib_ok = fh.read().strip() == source_hash() and not extra_all
    except OSError:
        lib_ok = False
    if force or procs or not lib_ok or _stale(LIB, objs):
        if os.path.exists(STAMP):
            os.remove(STAMP)
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stdout.decode(errors='replace'))
        with open(STAMP, 'w') as fh:             # lib.load() refuses a binary that does not match the sources beside it
            fh.write(source_hash() + '\n')
    build_pyresults(force, verbose)
    return LIB


def build_pyresults(force=False, verbose=False):
    """The small CPython module that builds the reference's list-of-dicts results (csrc/pyresults.c); gcc, no GPU code.
    Optional: results.py falls back to a Python comprehension when it cannot be built."""
    import sysconfig
    src = os.path.join(CSRC, 'pyresults.c')
    out = os.path.join(HERE, '_pyresults' + (sysconfig.get_config_var('EXT_SUFFIX') or '.so'))
    try:
        import numpy
        inc = [sysconfig.get_paths()['include'], numpy.get_include()]
        if not os.path.exists(os.path.join(inc[0], 'Python.h')):
            return None
        want = _unit_hash(src, [], ['gcc', '-O2'] + inc)
        try:
            with open(out + '.stamp') as fh:
                have = fh.read().strip()
        except OSError:
            have = None
        if force or have != want or not os.path.exists(out):
            cmd = ['gcc', '-O2', '-shared', '-fPIC'] + ['-I' + i for i in inc] + [src, '-o', out]
            if verbose:
                print(' '.join(cmd))
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            if r.returncode != 0:
                if verbose:
                    print(r.stdout.decode(errors='replace'))
                return None
            with open(out + '.stamp', 'w') as fh:
                fh.write(want + '\n')
        return out
    except Exception:                                 # noqa: BLE001  (optional component)
        return None


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
