"""Imports the reference's own PYTHON package (``/root/reference/src/openpifpaf``) in the build container.

TEST INFRASTRUCTURE ONLY -- used by ``tests/golden/make_golden_tracking.py`` to generate golden vectors for
the host-side callers of the decode path (tracking decoders); nothing under ``openpifpaf_amd/`` imports it, and
it cannot run on the GPU box (no /root/reference there).

The reference package does not import as it lies: its compiled extension is not next to the read-only sources
(``cpp_extension.py:22-26``) and ``torchvision`` / ``pysparkling`` / ``pycocotools`` are not installed
(SURVEY.md section 8c).  The extension is replaced by ``oracle/_ref/openpifpaf_ref.so`` -- the same C++ sources,
built by ``oracle/build_ref.py`` -- and the absent third-party modules, which the decoders never touch, by mocks.
"""
import sys
import types
from unittest import mock

REF_SRC = '/root/reference/src'
_MOCKED = ('torchvision', 'torchvision.models', 'torchvision.transforms', 'torchvision.transforms.functional',
           'torchvision.ops', 'torchvision.models.detection', 'pysparkling', 'pycocotools', 'pycocotools.coco',
           'pycocotools.cocoeval', 'cv2', 'thop')


def load():
    """-> the reference's ``openpifpaf`` package, its native decoder being ``oracle/_ref``."""
    if 'openpifpaf' in sys.modules and getattr(sys.modules['openpifpaf'], '_opa_reference', False):
        return sys.modules['openpifpaf']
    from . import reference
    reference.load()                                   # torch.classes.openpifpaf_decoder* from oracle/_ref
    for name in _MOCKED:
        if name not in sys.modules:
            m = mock.MagicMock()
            m.__version__ = '0.0'
            sys.modules[name] = m
    stub = types.ModuleType('openpifpaf.cpp_extension')
    stub.register_ops = lambda: None
    sys.modules['openpifpaf.cpp_extension'] = stub
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import openpifpaf
    openpifpaf._opa_reference = True
    return openpifpaf
