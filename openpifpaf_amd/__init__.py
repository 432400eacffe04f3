"""MI355X-native CifCaf decode path behind the OpenPifPaf decoder plugin API.

See DESIGN.md (what is built and why) and INTEGRATION.md (how it binds to the
reference).  The compute lives in ``lib/libopenpifpaf_amd.so`` (hand-written
HIP for gfx950, C ABI in ``include/openpifpaf_amd.h``); this package is the
host-side mirror of the reference's decoder interface.  There is no CPU
fallback: using the decode path without the built library / without a GPU raises.
"""
from . import constants, headmeta, synth                      # noqa: F401  (light, no torch)

__version__ = '0.1.0'

_LAZY = {
    'Annotation': ('annotation', 'Annotation'),
    'Predictor': ('predictor', 'Predictor'),
    'DECODERS': ('decoder', 'DECODERS'),
    'Decoder': ('decoder', 'Decoder'),
    'CifCaf': ('decoder', 'CifCaf'),
    'TrackingPose': ('tracking', 'TrackingPose'),
}


def __getattr__(name):
    import importlib
    if name in _LAZY:
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module('.' + mod, __name__), attr)
    if name in ('decoder', 'native', 'network', 'predictor', 'annotation', '_lib', 'build', 'tracking',
                'torchscript', 'fused', 'distributed'):
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(name)


def register():
    """Plugin entry point (reference ``plugin.py:17-40`` discovers importable modules whose
    name starts with ``openpifpaf_`` and calls ``register()``): make the HIP-backed
    ``CifCaf`` the decoder a host ``openpifpaf`` installation selects."""
    from . import decoder
    try:
        import openpifpaf
    except ImportError:
        return
    from . import tracking
    # the tracking decoders of a host that has the reference package are ITS classes with the HIP pose generator
    # (tracking.host_classes); this package's own restatement of their bookkeeping is for hosts without it
    try:
        host_tracking = set(tracking.host_classes(openpifpaf))
    except (ImportError, AttributeError):
        host_tracking = {tracking.TrackingPose, tracking.PoseSimilarity}
    ours = {decoder.CifCaf, decoder.CifCafDense, decoder.CifDet} | host_tracking
    names = {d.__name__ for d in ours}
    # mutate the set in place: the reference's factory iterates the very same object
    # (decoder/factory.py:17, re-exported by openpifpaf/__init__.py:28)
    for d in [d for d in openpifpaf.DECODERS if d.__name__ in names]:
        openpifpaf.DECODERS.discard(d)
    openpifpaf.DECODERS.update(ours)
