"""Builds ``lib/libopenpifpaf_amd.so`` with hipcc for gfx950 (in-tree, so that the
shared object travels with the repository snapshot to the GPU box)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'lib', 'libopenpifpaf_amd.so')
# -ffp-contract=off: the decode must reproduce the reference's float/double
# operation sequence; the reference's x86-64 build has no FMA contraction.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
         '-Wall', '-Wno-unused-result']
OBJ_DIR = os.path.join(HERE, 'lib', 'obj')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.hpp')) + \
        [os.path.join(HERE, '..', 'include', 'openpifpaf_amd.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=True):
    """One object per .hip file (compiled in parallel, rebuilt only when the file or a shared header changed),
    then one link."""
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get('HIPCC', 'hipcc')
    headers = glob.glob(os.path.join(CSRC, '*.hpp')) + [os.path.join(HERE, '..', 'include', 'openpifpaf_amd.h')]
    t_hdr = max(os.path.getmtime(h) for h in headers)
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), t_hdr):
            jobs.append([hipcc] + FLAGS + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
            list(pool.map(run, jobs))
    run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs)
    return OUT


def build_diagnostic(define, suffix, verbose=True, source='cifcaf.hip'):
    """A second library with one diagnostic switch of the association kernel compiled in (``-D<define>``), e.g.
    ``build_diagnostic('OPA_ASSOC_PHASE_TIMING', 'ph')`` -> ``lib/libopenpifpaf_amd_ph.so``; load it with
    ``OPA_LIB_PATH``.  Only cifcaf.hip is recompiled, the other objects are the production ones."""
    build_native(verbose=verbose)
    hipcc = os.environ.get('HIPCC', 'hipcc')
    obj = os.path.join(OBJ_DIR, '%s_%s.o' % (source[:-4], suffix))
    out = os.path.join(HERE, 'lib', 'libopenpifpaf_amd_%s.so' % suffix)
    defines = [define] if isinstance(define, str) else list(define)
    subprocess.check_call([hipcc] + FLAGS + ['-D' + d for d in defines] + ['-c', os.path.join(CSRC, source), '-o', obj])
    objs = [os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + '.o') for src in sources() if not src.endswith(os.sep + source)]
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out, obj] + objs)
    return out


def build_variants(verbose=True):
    """The measured-and-compiled-out designs that the GPU tests still decode through (each a second library, loaded with
    ``OPA_LIB_PATH`` in a process of its own): the self-serve hand-out of the association kernel
    (``-DOPA_ASSOC_SELFSERVE=1`` -> ``lib/libopenpifpaf_amd_selfserve.so``, ``tests/test_gpu_selfserve.py``).  Rebuilt only
    when a source is newer."""
    out = os.path.join(HERE, 'lib', 'libopenpifpaf_amd_selfserve.so')
    deps = sources() + glob.glob(os.path.join(CSRC, '*.hpp')) + [os.path.join(HERE, '..', 'include', 'openpifpaf_amd.h')]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return [out]
    return [build_diagnostic('OPA_ASSOC_SELFSERVE=1', 'selfserve', verbose=verbose)]


TORCH_SRC = os.path.join(CSRC, 'torch_binding.cpp')
TORCH_OUT = os.path.join(HERE, 'lib', 'libopenpifpaf_amd_torch.so')


def build_torch_binding(force=False, verbose=True):
    """TorchScript custom-class binding (csrc/torch_binding.cpp): host C++ only, g++ against the
    installed libtorch, linked to libopenpifpaf_amd.so next to it (rpath $ORIGIN)."""
    build_native(force=False, verbose=verbose)
    deps = [TORCH_SRC, OUT, os.path.join(HERE, '..', 'include', 'openpifpaf_amd.h')]
    if not force and os.path.exists(TORCH_OUT) and \
            all(os.path.getmtime(d) <= os.path.getmtime(TORCH_OUT) for d in deps):
        return TORCH_OUT
    import torch
    from torch.utils import cpp_extension
    inc = cpp_extension.include_paths() + ['/opt/rocm/include']
    libdirs = cpp_extension.library_paths()
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = (['g++', '-std=c++17', '-O2', '-fPIC', '-shared', '-w', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
            '-D_GLIBCXX_USE_CXX11_ABI=%d' % abi, '-DTORCH_API_INCLUDE_EXTENSION_H']
           + ['-I' + i for i in inc] + [TORCH_SRC, '-o', TORCH_OUT]
           + ['-L' + os.path.dirname(OUT), '-lopenpifpaf_amd', '-Wl,-rpath,$ORIGIN']
           + ['-L' + d for d in libdirs] + ['-Wl,-rpath,' + d for d in libdirs]
           + ['-lc10', '-lc10_hip', '-ltorch_cpu', '-ltorch'])
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return TORCH_OUT


if __name__ == '__main__':
    build_native(force=True)
    build_torch_binding(force=True)
