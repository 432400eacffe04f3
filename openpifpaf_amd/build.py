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
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
         '-Wall', '-Wno-unused-result']


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
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get('HIPCC', 'hipcc')
    cmd = [hipcc] + FLAGS + ['-o', OUT] + sources()
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build_native(force=True)
