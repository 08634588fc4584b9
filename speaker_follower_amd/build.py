"""Build libsf_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m speaker_follower_amd.build          # incremental
    python -m speaker_follower_amd.build --force  # rebuild everything

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU
box with the working tree.
"""
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
SOURCES = ['sf_gemm.hip', 'sf_attention.hip', 'sf_pointwise.hip', 'sf_persist.hip', 'sf_nav.hip', 'sf_api.hip']
# libsf_experimental.so (on demand, --experimental): the product's kernel objects + the persistent decode loop;
# experimental/sf_mega_api.hip textually includes sf_api.hip, so sf_api.o is NOT linked into it
EXP_LIB = os.path.join(PKG, 'libsf_experimental.so')
EXP_SOURCES = ['experimental/sf_mega.hip', 'experimental/sf_mega_api.hip']
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-I' + CSRC]


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: cannot build libsf_hip.so')
    return exe


def _deps_mtime(experimental=False):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(PKG), 'include', 'sf_hip.h'))
    if experimental:
        headers += [os.path.join(CSRC, 'experimental', 'sf_mega.h'), os.path.join(CSRC, 'sf_api.hip'),
                    os.path.join(os.path.dirname(PKG), 'include', 'sf_hip_experimental.h')]
    return max(os.path.getmtime(h) for h in headers)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src).replace('.hip', '.o'))
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(path), _deps_mtime('experimental' in src))):
        return obj, False
    cmd = [_hipcc()] + FLAGS + ['-c', path, '-o', obj]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, res.stdout, res.stderr))
    if res.stderr.strip():
        sys.stderr.write(res.stderr)
    return obj, True


def build_lib(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (res.stdout, res.stderr))
    if verbose:
        print('libsf_hip.so %s (%d bytes)' % ('rebuilt' if rebuilt else 'up to date',
                                              os.path.getsize(LIB)))
    return LIB


def build_experimental(force=False, verbose=True):
    """libsf_experimental.so: sf_follower_decode_persistent (include/sf_hip_experimental.h).  Not built by
    __graft_entry__.build(); tests/test_gpu_mega.py and FollowerEngine.persistent_decode need it."""
    os.makedirs(OBJ, exist_ok=True)
    shared = [s for s in SOURCES if s != 'sf_api.hip']
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda s: _compile(s, force), shared + EXP_SOURCES))
    objs = [o for o, _ in results]
    if any(r for _, r in results) or not os.path.exists(EXP_LIB):
        cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', EXP_LIB] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (res.stdout, res.stderr))
    if verbose:
        print('libsf_experimental.so (%d bytes)' % os.path.getsize(EXP_LIB))
    return EXP_LIB


def build_sim(force=False, verbose=True):
    """The navigation-only MatterSim pybind11 module (g++, no OpenCV / GL / jsoncpp) and, next to it, the
    batched panorama sweep `sf_sweep` (sweep_py.cpp) over the same simulator sources."""
    import sysconfig
    import pybind11
    sim = os.path.join(PKG, 'sim')
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    outs = []
    for name, files in (('MatterSim', ('mattersim_nav.cpp', 'mattersim_py.cpp')),
                        ('sf_sweep', ('mattersim_nav.cpp', 'sweep_py.cpp'))):
        out = os.path.join(sim, name + ext)
        srcs = [os.path.join(sim, f) for f in files]
        deps = srcs + [os.path.join(sim, 'mattersim_nav.hpp')]
        if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
            cmd = ['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-fvisibility=hidden',
                   '-I' + pybind11.get_include(), '-I' + sysconfig.get_paths()['include']] + srcs + ['-o', out]
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError('%s build failed:\n%s\n%s' % (name, res.stdout, res.stderr))
            if verbose:
                print('%s module rebuilt (%d bytes)' % (name, os.path.getsize(out)))
        outs.append(out)
    return outs[0]


if __name__ == '__main__':
    build_lib(force='--force' in sys.argv)
    build_sim(force='--force' in sys.argv)
    if '--experimental' in sys.argv:
        build_experimental(force='--force' in sys.argv)
