"""TorchScript face of the HIP decode path.

``load()`` registers the custom classes/ops of ``csrc/torch_binding.cpp`` -- the same
names the reference registers in ``csrc/src/module.cpp:19-118``, under the namespaces
``openpifpaf_amd`` / ``openpifpaf_amd_decoder`` / ``openpifpaf_amd_decoder_utils`` -- and
``DecoderModule`` / ``EncoderDecoder`` mirror the reference's export wrappers
(``export_torchscript.py:15-43``): a scriptable module that owns the decoder object and maps
head outputs to ``(annotations[n,K,4], ids[n])``.  A C++ host loads the scripted file after
``dlopen``-ing ``lib/libopenpifpaf_amd_torch.so`` exactly like ``cpp/cli_video.cpp:48-64`` does
with the reference's extension.
"""
import os

import torch

from . import build as _build

_LOADED = False


def library_path():
    return os.environ.get('OPA_TORCH_LIB_PATH', _build.TORCH_OUT)


def load():
    """Registers ``torch.classes.openpifpaf_amd_decoder.CifCaf`` & co.  Raises if the binding
    has not been built (``python -m openpifpaf_amd.build``); there is no Python fallback."""
    global _LOADED
    if not _LOADED:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError('%s is missing: run `python -m openpifpaf_amd.build`' % path)
        torch.ops.load_library(path)
        _LOADED = True
    return torch.classes.openpifpaf_amd_decoder


class DecoderModule(torch.nn.Module):
    """ref: export_torchscript.py:15-29"""

    def __init__(self, cif_meta, caf_meta):
        super().__init__()
        load()
        self.cif_stride = int(cif_meta.stride)
        self.caf_stride = int(caf_meta.stride)
        self.cpp_decoder = torch.classes.openpifpaf_amd_decoder.CifCaf(
            len(cif_meta.keypoints), torch.LongTensor(caf_meta.skeleton) - 1)

    def forward(self, cif_field, caf_field):
        return self.cpp_decoder.call(cif_field, self.cif_stride, caf_field, self.caf_stride)


class EncoderDecoder(torch.nn.Module):
    """ref: export_torchscript.py:32-43 -- traced network + scripted decoder, one image."""

    def __init__(self, traced_encoder, decoder):
        super().__init__()
        self.traced_encoder = traced_encoder
        self.decoder = decoder

    def forward(self, x):
        cif_head_batch, caf_head_batch = self.traced_encoder(x)
        return self.decoder(cif_head_batch[0], caf_head_batch[0])
