"""Build lib3dinfomax_hip.so (hipcc, gfx950 only) in-tree: 3dinfomax_amd/lib/lib3dinfomax_hip.so.

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(HERE, 'build')
LIB = os.path.join(LIBDIR, 'lib3dinfomax_hip.so')
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'infomax3d_hip.h')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result']


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stamp():
    h = hashlib.sha256()
    for p in _sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + [HEADER, os.path.abspath(__file__)]:
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def is_current():
    stamp = os.path.join(LIBDIR, 'build.stamp')
    return os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == _stamp()


def build(force=False, verbose=True):
    """Compile every .hip in csrc/ for gfx950 and link the C-ABI shared library."""
    if not force and is_current():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + '.o')
        # per-object stamp: the source, every header it can include and the flags (an unchanged unit is not recompiled)
        h = hashlib.sha256(' '.join(FLAGS).encode())
        for p in [src, HEADER] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')):
            with open(p, 'rb') as f:
                h.update(f.read())
        tag, tagfile = h.hexdigest(), obj + '.stamp'
        if not force and os.path.exists(obj) and os.path.exists(tagfile) and open(tagfile).read().strip() == tag:
            return obj
        cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(tagfile, 'w') as f:
            f.write(tag)
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, _sources()))
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs, capture_output=True,
                       text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    with open(os.path.join(LIBDIR, 'build.stamp'), 'w') as f:
        f.write(_stamp())
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
