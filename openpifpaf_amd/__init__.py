"""MI355X-native CifCaf decode path behind the OpenPifPaf decoder plugin API.

See DESIGN.md (what is built and why) and INTEGRATION.md (how it binds to the
reference).  The compute lives in ``lib/libopenpifpaf_amd.so`` (hand-written
HIP for gfx950, C ABI in ``include/openpifpaf_amd.h``); this package is the
host-side mirror of the reference's decoder interface.  There is no CPU
fallback: using the decode path without the built library / without a GPU raises.
"""
import os

# Decode lanes (native.DecodeLanes, decoder.CifCaf.decoder_workers, one HIP stream each) only run side by side while every
# stream has a hardware queue of its own; the HIP runtime maps all streams of a process onto FOUR by default, and four
# lanes then share queues with the network's stream (measured in round 4: 54 k images/s with four lanes against 66 k with
# two; with eight queues 86 k).  The runtime reads the variable when it initialises, i.e. at the first GPU call of the
# process: import this package (or set the variable yourself) before that.  An explicit setting wins.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

from . import constants, headmeta, synth                      # noqa: F401,E402  (light, no torch)

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
