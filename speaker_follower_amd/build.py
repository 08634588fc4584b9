"""Build libsf_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m speaker_follower_amd.build          # incremental
    python -m speaker_follower_amd.build --force  # rebuild everything

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU
box with the working tree.

Staleness is decided by CONTENT, not by mtime (a checkout leaves arbitrary mtimes): every object
carries a side file with the sha256 of its source, of every header and of the compiler flags, and
the library itself carries `build_id()` -- the same hash over ALL sources -- as the string
`sf_build_id()` returns.  `_lib` compares that string with the sources on disk at import and refuses a
library that was built from different ones.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(PKG, 'libsf_hip.so')
ARCH = 'gfx950'
SOURCES = ['sf_gemm.hip', 'sf_attention.hip', 'sf_pointwise.hip', 'sf_persist.hip', 'sf_nav.hip', 'sf_precise.hip', 'sf_api.hip']
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-I' + CSRC]


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: cannot build libsf_hip.so')
    return exe


def _headers():
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h'))
    return hs + [os.path.join(os.path.dirname(PKG), 'include', 'sf_hip.h')]


def _digest(paths, extra=''):
    h = hashlib.sha256(extra.encode())
    for path in paths:
        h.update(os.path.basename(path).encode() + b'\0')
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_id():
    """Hash of every source and header of the library and of the compiler flags (what sf_build_id() must return)."""
    flags = ' '.join(f for f in FLAGS if not f.startswith('-I'))
    return _digest([os.path.join(CSRC, s) for s in SOURCES] + _headers(), flags)


def _compile(src, force, bid):
    obj = os.path.join(OBJ, os.path.basename(src).replace('.hip', '.o'))
    path = os.path.join(CSRC, src)
    flags = ' '.join(f for f in FLAGS if not f.startswith('-I'))
    # sf_api.hip holds sf_build_id(): its object depends on the id of the whole library
    want = _digest([path] + _headers(), flags + (bid if src == 'sf_api.hip' else ''))
    side = obj + '.sha'
    if not force and os.path.exists(obj) and os.path.exists(side) and open(side).read().strip() == want:
        return obj, False
    cmd = [_hipcc()] + FLAGS + ['-DSF_BUILD_ID="%s"' % bid, '-c', path, '-o', obj]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, res.stdout, res.stderr))
    if res.stderr.strip():
        sys.stderr.write(res.stderr)
    with open(side, 'w') as f:
        f.write(want)
    return obj, True


def lib_build_id(path=LIB):
    """The id compiled into an existing library, read without loading it (the bytes after the marker)."""
    if not os.path.exists(path):
        return None
    with open(path, 'rb') as f:
        blob = f.read()
    i = blob.find(b'SF_BUILD_ID=')
    return blob[i + 12:i + 28].decode('ascii', 'replace') if i >= 0 else None


def build_lib(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    bid = build_id()
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda s: _compile(s, force, bid), SOURCES))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or lib_build_id() != bid:
        cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (res.stdout, res.stderr))
        rebuilt = True
    if verbose:
        print('libsf_hip.so %s (%d bytes, build id %s)' % ('rebuilt' if rebuilt else 'up to date',
                                                          os.path.getsize(LIB), bid))
    return LIB


def build_sim(force=False, verbose=True):
    """The navigation-only MatterSim pybind11 module (g++, no OpenCV / GL / jsoncpp) and, next to it, the
    batched panorama sweep `sf_sweep` (sweep_py.cpp) over the same simulator sources and the
    state-factored search's bookkeeping `sf_frontier` (frontier_core.cpp)."""
    import sysconfig
    import pybind11
    sim = os.path.join(PKG, 'sim')
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    outs = []
    for name, files in (('MatterSim', ('mattersim_nav.cpp', 'mattersim_py.cpp')),
                        ('sf_sweep', ('mattersim_nav.cpp', 'sweep_py.cpp')),
                        ('sf_frontier', ('frontier_core.cpp',))):
        out = os.path.join(sim, name + ext)
        srcs = [os.path.join(sim, f) for f in files]
        deps = srcs + [os.path.join(sim, 'mattersim_nav.hpp')]
        gxx = ['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-fvisibility=hidden']
        want = _digest(deps, ' '.join(gxx) + ext)                  # staleness by content, like the HIP library
        side = out + '.sha'
        if force or not os.path.exists(out) or not os.path.exists(side) or open(side).read().strip() != want:
            cmd = gxx + ['-I' + pybind11.get_include(), '-I' + sysconfig.get_paths()['include']] + srcs + ['-o', out]
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError('%s build failed:\n%s\n%s' % (name, res.stdout, res.stderr))
            with open(side, 'w') as f:
                f.write(want)
            if verbose:
                print('%s module rebuilt (%d bytes)' % (name, os.path.getsize(out)))
        outs.append(out)
    return outs[0]


if __name__ == '__main__':
    build_lib(force='--force' in sys.argv)
    build_sim(force='--force' in sys.argv)
