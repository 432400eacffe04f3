"""Build recipe for ``oracle/_ref/openpifpaf_ref.so`` -- the REAL reference decoder.

TEST INFRASTRUCTURE ONLY.  Nothing under ``openpifpaf_amd/`` may import this.

Compiles the reference's own, unmodified C++ sources *where they lie* under
``/root/reference/src/openpifpaf/csrc/src/*.cpp`` (headers from
``csrc/include``) with plain ``g++`` against the libtorch that ships with the
image's PyTorch.  No reference source is copied into this repository; the only
output is the shared object in ``oracle/_ref/`` (git-ignored, travels to the GPU
box with the snapshot).

Flags mirror the reference's ``setup.py:19-56`` (``-std=c++17``, optimisation
level from the Python sysconfig, i.e. -O2/-O3, **no** ``-march=native`` so that
the oracle - like the published wheels - has no FMA contraction).

Usage:  python oracle/build_ref.py            (no-op when /root/reference is absent)
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get('OPENPIFPAF_REFERENCE', '/root/reference')
REF_CSRC = os.path.join(REF_ROOT, 'src', 'openpifpaf', 'csrc')
OUT_DIR = os.path.join(HERE, '_ref')
OUT_SO = os.path.join(OUT_DIR, 'openpifpaf_ref.so')


def build(force=False, verbose=True):
    sources = sorted(glob.glob(os.path.join(REF_CSRC, 'src', '*.cpp')))
    if not sources:
        if verbose:
            print('oracle/_ref: reference sources not present at %s; '
                  'using prebuilt %s' % (REF_CSRC, OUT_SO))
        return OUT_SO if os.path.exists(OUT_SO) else None
    if os.path.exists(OUT_SO) and not force:
        newest = max(os.path.getmtime(s) for s in sources)
        if os.path.getmtime(OUT_SO) > newest:
            return OUT_SO

    import torch
    from torch.utils import cpp_extension
    os.makedirs(OUT_DIR, exist_ok=True)
    inc = [os.path.join(REF_CSRC, 'include')] + cpp_extension.include_paths()
    libdirs = cpp_extension.library_paths()
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    objs = []
    procs = []
    for src in sources:
        obj = os.path.join(OUT_DIR, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        cmd = (['g++', '-std=c++17', '-O3', '-fPIC', '-w',
                '-D_GLIBCXX_USE_CXX11_ABI=%d' % abi, '-DTORCH_API_INCLUDE_EXTENSION_H']
               + ['-I' + i for i in inc] + ['-c', src, '-o', obj])
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('failed to compile reference source %s' % src)
    cmd = (['g++', '-shared', '-o', OUT_SO] + objs
           + ['-L' + d for d in libdirs] + ['-Wl,-rpath,' + d for d in libdirs]
           + ['-lc10', '-ltorch_cpu', '-ltorch'])
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    if verbose:
        print('oracle/_ref: built', OUT_SO)
    return OUT_SO


if __name__ == '__main__':
    build(force='--force' in sys.argv)
