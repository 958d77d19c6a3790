"""Build ``csrc/libegonet_hip.so`` in-tree with hipcc for gfx950.

    python -m egonet_amd.build [--force]

hipcc cross-compiles without a GPU.  The shared object is git-ignored but
travels with the repo snapshot to the GPU box.  The library is linked to a
temporary name, dlopen'ed (catches undefined symbols such as a kernel stub the
compiler silently dropped) and only then moved into place, so a failed build
never leaves a stale library behind.
"""
import ctypes
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(CSRC, 'libegonet_hip.so')
ARCH = 'gfx950'


def sources():
    """HIP kernels + host-only C++ (the KITTI evaluator) of the one shared library."""
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))


def needs_build():
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        [os.path.join(os.path.dirname(HERE), 'include', 'egonet_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_objects(hipcc, flags, probes, force, verbose):
    """One object per source under csrc/_obj/ (git-ignored), rebuilt when the source, any header or the flags are
    newer -- the sources are independent translation units, so a one-file edit costs one compile; up to 8 at once."""
    from concurrent.futures import ThreadPoolExecutor
    odir = os.path.join(CSRC, '_obj', 'probes' if probes else 'product')
    os.makedirs(odir, exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(os.path.dirname(HERE), 'include', 'egonet_hip.h')]
    th = max(os.path.getmtime(h) for h in hdrs)
    stamp = os.path.join(odir, 'flags.txt')
    flag_text = ' '.join([hipcc] + flags)
    if not os.path.isfile(stamp) or open(stamp).read() != flag_text:
        force = True
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(odir, os.path.basename(src) + '.o')
        objs.append(obj)
        if force or not os.path.isfile(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), th):
            jobs.append([hipcc] + flags + (['-x', 'hip'] if src.endswith('.hip') else []) + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    workers = int(os.environ.get('EGONET_AMD_BUILD_JOBS', '8'))
    with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
        list(ex.map(run, jobs))
    with open(stamp, 'w') as f:
        f.write(flag_text)
    return objs


PROBES_OUT = os.path.join(os.path.dirname(HERE), 'tools', '_build', 'libegonet_hip_probes.so')


def build(force=False, verbose=True, probes=False):
    """``probes``: the -DEGN_PROBES build for tools/ (timing-ablation / s_memtime-stamp builds and the kernel
    families that were measured and retired) -> tools/_build/libegonet_hip_probes.so; select it with
    EGONET_AMD_LIB=<that path>.  The product library never contains those kernels."""
    out = PROBES_OUT if probes else OUT
    if probes:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        if not force and os.path.isfile(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in
                                                     sources() + glob.glob(os.path.join(CSRC, '*.h'))):
            return out
    elif not force and not needs_build():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    tmp = out + '.tmp'
    flags = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-Wno-unused-value',
             '-Wno-inline-asm'] + (['-DEGN_PROBES'] if probes else [])
    objs = _compile_objects(hipcc, flags, probes, force, verbose)
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', tmp] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
        import torch  # noqa: F401  (its libamdhip64 must be the process' HIP runtime)
        ctypes.CDLL(tmp)
    except Exception:
        if os.path.exists(tmp):
            os.remove(tmp)
        if os.path.exists(out):
            os.remove(out)
        raise
    os.replace(tmp, out)
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, probes='--probes' in sys.argv))
