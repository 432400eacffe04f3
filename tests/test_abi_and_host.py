"""CPU-side checks: the C-ABI library loads and exports every symbol the header
declares; the host-side mirror of the reference interface behaves like the reference
(registry, factory, CLI -> global tunables, Annotation, preprocessing, field shapes).
No compute call is made here (there is no GPU in this tier and no CPU fallback)."""
import argparse
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'openpifpaf_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(opa_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from openpifpaf_amd import _lib
    assert _lib.available(), 'build the HIP library first: python -c "import __graft_entry__ as g; g.build()"'
    handle = ctypes.CDLL(_lib.LIB_PATH)
    declared = header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(handle, name), 'library does not export %s' % name
    assert sorted(_lib.SYMBOLS) == declared, 'ctypes table and header disagree'
    assert _lib.lib().opa_version().decode().startswith('openpifpaf_amd')


def test_params_struct_matches_reference_defaults():
    from openpifpaf_amd import _lib
    from oracle import port
    p, q = _lib.default_params(), port.default_params()
    assert ctypes.sizeof(_lib.Params) == ctypes.sizeof(port.Params) == 13 * 8 + 8 * 4
    for name, _ in _lib.Params._fields_:
        assert getattr(p, name) == getattr(q, name), name
    assert (p.cif_threshold, p.seed_threshold, p.caf_threshold) == (0.3, 0.2, 0.3)
    assert (p.keypoint_threshold, p.keypoint_threshold_rel, p.reverse_match, p.greedy) == (0.15, 0.5, 1, 0)


def test_static_getset_is_process_global_like_the_reference():
    from openpifpaf_amd import native
    assert native.CifCaf.get_keypoint_threshold() == 0.15
    try:
        native.CifCaf.set_keypoint_threshold(0.4)
        native.CifSeeds.set_threshold(0.25)
        assert native.CifCaf.get_keypoint_threshold() == 0.4 and native.CifSeeds.get_threshold() == 0.25
        assert native.CifCaf.get_reverse_match() is True
    finally:
        native.CifCaf.set_keypoint_threshold(0.15)
        native.CifSeeds.set_threshold(0.2)


def test_invalid_arguments_fail_loudly_without_gpu():
    from openpifpaf_amd import _lib
    L = _lib.lib()
    shape = _lib.Shape(1, 17, 19, 0, 81, 81, 81, 8, 8, 64)            # zero height
    assert L.opa_cifcaf_workspace_bytes(ctypes.byref(shape)) == 0
    assert b'positive' in L.opa_last_error()
    shape = _lib.Shape(32, 17, 19, 81, 81, 81, 81, 8, 8, 128)
    nbytes = L.opa_cifcaf_workspace_bytes(ctypes.byref(shape))
    assert 0.7e9 < nbytes < 1.1e9                                      # ~29 MB per image: the CIF map is a pool of tiles (r4; the
    #                                                                    dense map of rounds 1-3: ~51 MB per image)
    dense = _lib.Shape(32, 17, 19, 81, 81, 81, 81, 8, 8, 128, 0, -1)   # cifhr_pool_tiles = -1: a slot for every tile
    nbytes_dense = L.opa_cifcaf_workspace_bytes(ctypes.byref(dense))
    per_image = (nbytes_dense - nbytes) / 32
    assert 21e6 < per_image < 24e6                                     # 3927 - 1024 tiles of 8 KB
    # (the automatic size is 1024 tiles per image + ONE spill region for the batch that holds the rest of a whole map;
    # an explicit capacity has no spill region)
    small = _lib.Shape(32, 17, 19, 81, 81, 81, 81, 8, 8, 128, 0, 512)
    assert (nbytes - (3927 - 1024) * 8192 - L.opa_cifcaf_workspace_bytes(ctypes.byref(small))) / 32 == 512 * 8192
    # a map of at most 32 MB per image is kept whole (nothing can run out): the 47 x 16 fields of the round-4 sweeps
    whole = _lib.Shape(16, 133, 160, 16, 47, 16, 47, 8, 8, 128, 0, 0)
    whole_full = _lib.Shape(16, 133, 160, 16, 47, 16, 47, 8, 8, 128, 0, -1)
    assert L.opa_cifcaf_workspace_bytes(ctypes.byref(whole)) == L.opa_cifcaf_workspace_bytes(ctypes.byref(whole_full))
    # the structs are passed by pointer: the library says which header it was built from (checked at load, _lib.lib())
    assert L.opa_abi_version() == _lib.ABI_VERSION and L.opa_shape_bytes() == ctypes.sizeof(_lib.Shape)
    assert L.opa_params_bytes() == ctypes.sizeof(_lib.Params)
    # without force complete the second CAF list set is left out ("weak" 15): ~7 MB per image less
    plain = L.opa_cifcaf_workspace_bytes_for(ctypes.byref(shape), ctypes.byref(_lib.default_params()))
    full = L.opa_cifcaf_workspace_bytes_for(ctypes.byref(shape), ctypes.byref(_lib.default_params(force_complete=1)))
    assert full == nbytes and 32 * 6.5e6 < nbytes - plain < 32 * 8e6
    h = ctypes.c_void_p()
    skel = (ctypes.c_int64 * 4)(0, 1, 1, 99)
    assert L.opa_cifcaf_create(ctypes.byref(h), 17, skel, 2) == 1      # index out of range


def test_decoder_without_gpu_raises_instead_of_falling_back():
    import torch
    from openpifpaf_amd import _lib, native
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.NativeError):
        native.CifCaf(17, torch.zeros((19, 2), dtype=torch.int64))


def test_registry_factory_and_cli():
    import torch
    if torch.cuda.is_available():
        pytest.skip('constructing decoders on the GPU box is covered by the gpu tests')
    from openpifpaf_amd import decoder, headmeta, native
    assert decoder.CifCaf in decoder.DECODERS
    parser = argparse.ArgumentParser()
    decoder.cli(parser)
    args = parser.parse_args(['--force-complete-pose', '--seed-threshold', '0.3', '--caf-th', '0.25',
                              '--decoder', 'cifcaf:0'])
    try:
        decoder.configure(args)
        assert native.CifCaf.get_force_complete() is True
        assert native.CifCaf.get_keypoint_threshold() == 0.0          # reference cifcaf.py:181-185
        assert native.CifCaf.get_keypoint_threshold_rel() == 0.0
        assert native.NMSKeypoints.get_instance_threshold() == 0.0    # reference factory.py:53-57
        assert native.CifSeeds.get_threshold() == 0.3
        assert native.CafScored.get_default_score_th() == 0.25
        assert decoder.Factory.decoder_request == {'cifcaf': [0]}
    finally:
        from openpifpaf_amd import _lib
        _lib.set_params(_lib.default_params())
        decoder.Factory.decoder_request = None
    # the factory pairs consecutive (Cif, Caf) metas (reference cifcaf.py:213-222); building needs a GPU
    cif, caf = headmeta.cocokp_metas()
    assert cif.stride == 8 and caf.n_fields == 19 and cif.n_fields == 17
    # CifCafDense (reference cifcaf.py:17-78) is offered only with --dense-connections and three heads
    cif, caf, dcaf = headmeta.cocokp_dense_metas()
    both = headmeta.Caf.concatenate([caf, dcaf])
    assert both.n_fields == 19 + 25 and both.skeleton[:19] == list(caf.skeleton) and both.head_index == 1
    assert both.decoder_confidence_scales == [1.0] * 44
    assert decoder.CifCafDense in decoder.DECODERS and decoder.CifCafDense.factory([cif, caf, dcaf]) == []


def test_annotation_matches_reference_semantics():
    from openpifpaf_amd import constants
    from openpifpaf_amd.annotation import Annotation
    ann = Annotation(constants.COCO_KEYPOINTS, constants.COCO_PERSON_SKELETON,
                     score_weights=constants.COCO_PERSON_SCORE_WEIGHTS)
    ann.data[:, 0] = np.arange(17) * 10.0
    ann.data[:, 1] = 50.0
    ann.data[:, 2] = np.linspace(0.2, 1.0, 17)
    ann.joint_scales[:] = 4.0
    w = np.asarray(constants.COCO_PERSON_SCORE_WEIGHTS) / np.sum(constants.COCO_PERSON_SCORE_WEIGHTS)
    assert abs(ann.score - float(np.sum(w * np.sort(ann.data[:, 2])[::-1]))) < 1e-7   # reference annotation.py:98-110
    j = ann.json_data()
    assert len(j['keypoints']) == 51 and j['bbox'] == [-4.0, 46.0, 168.0, 8.0] and j['category_id'] == 1
    meta = {'offset': np.array((-10.0, -20.0)), 'scale': np.array((2.0, 2.0)), 'hflip': False,
            'rotation': {'angle': 0.0, 'width': None, 'height': None}, 'width_height': np.array((100, 100))}
    inv = ann.inverse_transform(meta)
    assert np.allclose(inv.data[1, :2], [(10.0 - 10.0) / 2.0, (50.0 - 20.0) / 2.0])
    assert np.allclose(inv.joint_scales, 2.0) and ann.data[1, 0] == 10.0


def test_preprocess_pads_like_the_reference():
    from openpifpaf_amd.predictor import preprocess_image
    img = (np.random.default_rng(0).random((300, 500, 3)) * 255).astype(np.uint8)
    t, meta = preprocess_image(img, long_edge=641, batch_mode=True)
    assert tuple(t.shape) == (3, 641, 641)                              # CenterPad(long_edge)
    assert meta['offset'][0] == 0 and meta['offset'][1] < 0
    t, meta = preprocess_image(img, long_edge=None, batch_mode=False)
    assert tuple(t.shape) == (3, 305, 513)                              # CenterPadTight(16): ceil((w-1)/16)*16+1
    t, meta = preprocess_image(img, long_edge=321, batch_mode=False)
    assert t.shape[2] == 321 and (t.shape[1] - 1) % 16 == 0


@pytest.mark.parametrize('name,features', [('resnet18', 512), ('shufflenetv2k16', 1392)])
def test_forward_field_shapes(name, features):
    """reference tests/test_forward.py:8-32: (1,17,5,H,W) / (1,19,8,H,W); 241x321 -> 31x41 at upsample 2."""
    import torch
    from openpifpaf_amd import network
    net = network.factory(name)
    assert net.base_net.out_features == features and net.base_net.stride == 16
    with torch.no_grad():
        cif, caf = net(torch.zeros((1, 3, 241, 321)))
    assert tuple(cif.shape) == (1, 17, 5, 31, 41) and tuple(caf.shape) == (1, 19, 8, 31, 41)
    assert float(cif[0, 0, 1].min()) >= 0.0 and float(cif[0, 0, 1].max()) <= 1.0   # sigmoid on confidences
    assert float(cif[0, :, 4].min()) >= 0.0                                             # softplus on scales


def test_conv_bn_folding_is_exact_enough():
    import torch
    from openpifpaf_amd import network
    net = network.factory('resnet18')
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    x = torch.randn((1, 3, 97, 129))
    with torch.no_grad():
        a = net(x)
        network.fuse_conv_bn_(net)
        b = net(x)
    assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in net.modules())
    assert max(float((u - v).abs().max()) for u, v in zip(a, b)) < 1e-3


def test_synth_is_deterministic_and_shaped():
    from openpifpaf_amd import synth
    a = synth.synth_fields(5, 3, height=21, width=33)
    b = synth.synth_fields(5, 3, height=21, width=33)
    assert a[0].shape == (17, 5, 21, 33) and a[1].shape == (19, 8, 21, 33)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    cifs, cafs = synth.synth_batch(3, seed0=9, height=21, width=21)
    assert cifs.shape == (3, 17, 5, 21, 21) and cafs.dtype == np.float32


def test_register_replaces_cifcaf_in_a_host_openpifpaf(monkeypatch):
    import sys
    import types
    import openpifpaf_amd
    from openpifpaf_amd import decoder

    class OldCifCaf:
        pass
    OldCifCaf.__name__ = 'CifCaf'

    class CifDetHost:
        pass
    host = types.ModuleType('openpifpaf')
    host.DECODERS = {OldCifCaf, CifDetHost}
    monkeypatch.setitem(sys.modules, 'openpifpaf', host)
    openpifpaf_amd.register()
    assert decoder.CifCaf in host.DECODERS and OldCifCaf not in host.DECODERS and CifDetHost in host.DECODERS


def test_batched_inverse_transform_equals_per_annotation():
    import torch
    from openpifpaf_amd import constants
    from openpifpaf_amd.annotation import Annotation, inverse_transform_batch
    rng = np.random.default_rng(3)
    B, M, K = 3, 4, 17
    t = torch.from_numpy(rng.uniform(0, 300, (B, M, K, 4)).astype(np.float32))
    metas = [
        {'offset': np.array((-12.0, 3.0)), 'scale': np.array((0.5, 0.52)), 'hflip': False,
         'rotation': {'angle': 0.0, 'width': None, 'height': None}, 'width_height': np.array((640, 480))},
        {'offset': np.array((0.0, 0.0)), 'scale': np.array((1.0, 1.0)), 'hflip': True,
         'rotation': {'angle': 0.0, 'width': None, 'height': None}, 'width_height': np.array((321, 200))},
        None,
    ]
    got = inverse_transform_batch(t, metas).numpy()
    for b in range(B):
        for m in range(M):
            ann = Annotation(constants.COCO_KEYPOINTS, constants.COCO_PERSON_SKELETON)
            ann.data[:, :2], ann.data[:, 2], ann.joint_scales[:] = t[b, m, :, 1:3].numpy(), t[b, m, :, 0].numpy(), t[b, m, :, 3].numpy()
            want = ann.inverse_transform(metas[b])
            assert np.allclose(got[b, m, :, 1:3], want.data[:, :2], rtol=1e-6, atol=1e-4)
            assert np.allclose(got[b, m, :, 3], want.joint_scales, rtol=1e-6) and np.array_equal(got[b, m, :, 0], want.data[:, 2])


def _build_c_example(tmp_path):
    import subprocess
    exe = str(tmp_path / 'decode_c_abi')
    lib_dir = os.path.join(ROOT, 'openpifpaf_amd', 'lib')
    subprocess.check_call(['gcc', '-std=c99', '-w', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include',
                           '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'decode_c_abi.c'),
                           '-L' + lib_dir, '-lopenpifpaf_amd', '-L/opt/rocm/lib', '-lamdhip64',
                           '-Wl,-rpath,' + lib_dir, '-Wl,-rpath,/opt/rocm/lib', '-o', exe])
    return exe


def test_header_is_plain_c_and_a_c_host_links(tmp_path):
    """include/openpifpaf_amd.h must be usable from C (no torch, no C++): strict C99 compile of the header, and
    the C example links against the library and fails loudly without a device."""
    import subprocess
    import torch
    src = tmp_path / 'hdr.c'
    src.write_text('#include "openpifpaf_amd.h"\nint main(void) { return sizeof(opa_shape) + sizeof(opa_params) > 0 ? 0 : 1; }\n')
    subprocess.check_call(['gcc', '-std=c99', '-pedantic', '-Wall', '-Wextra', '-Werror',
                           '-I' + os.path.join(ROOT, 'include'), '-c', str(src), '-o', str(tmp_path / 'hdr.o')])
    exe = _build_c_example(tmp_path)
    if not torch.cuda.is_available():
        proc = subprocess.run([exe], capture_output=True, text=True)
        assert proc.returncode == 1 and 'no HIP device' in proc.stderr


@pytest.mark.gpu
def test_c_host_example_runs_on_the_gpu(tmp_path):
    import subprocess
    proc = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0, proc.stderr
    assert 'annotations per image: 0 0' in proc.stdout
    assert '6 batches over 2 lanes: 0 annotations' in proc.stdout        # the two-handle / two-stream recipe


def test_rescale_restatement_equals_scipy_zoom_pixel_for_pixel():
    """The reference rescales with ``scipy.ndimage.zoom(im, (th/h, tw/w, 1), order=1)`` (transforms/scale.py:57-63).
    ``predictor.zoom_linear_u8`` restates it with torch ops so that it can run on the device: every pixel equal."""
    import scipy.ndimage
    import torch
    from openpifpaf_amd import predictor
    rng = np.random.default_rng(3)
    sizes = [((60, 80), (72, 97)), ((150, 90), (193, 115)), ((48, 64), (24, 33)), ((37, 53), (37, 53)),
             ((480, 640), (481, 641)), ((5, 7), (97, 97)), ((42, 17), (103, 101)), ((20, 78), (80, 45))]
    sizes += [((int(rng.integers(4, 80)), int(rng.integers(4, 80))), (int(rng.integers(2, 120)), int(rng.integers(2, 120))))
              for _ in range(60)]                       # incl. sizes whose last coordinate rounds past the edge
    for (h, w), (th, tw) in sizes:
        im = (rng.random((h, w, 3)) * 255).astype(np.uint8)
        want = scipy.ndimage.zoom(im, (th / h, tw / w, 1), order=1)
        got = predictor.zoom_linear_u8(torch.from_numpy(im), th, tw).numpy()
        assert want.shape == got.shape == (th, tw, 3)
        assert np.array_equal(got, want), '%s -> %s: %d pixels differ' % ((h, w), (th, tw), int((got != want).sum()))


def test_fast_rescale_restatement_equals_pillow_pixel_for_pixel():
    """The reference Predictor's DEFAULT rescale (``fast_rescaling = True``, reference predictor.py:17,88) without
    OpenCV is Pillow's antialiased ``image.resize(size, BILINEAR)`` (transforms/scale.py:55-58).
    ``predictor.resize_bilinear_u8`` restates Pillow's 8-bit resampler with torch integer ops so that it can run on
    the device: every pixel equal, up- and downscaling, degenerate sizes included."""
    import PIL.Image
    import torch
    from openpifpaf_amd import predictor
    rng = np.random.default_rng(4)
    sizes = [((480, 640), (481, 641)), ((427, 640), (427, 641)), ((1080, 1920), (361, 641)), ((100, 130), (321, 417)),
             ((333, 500), (641, 962)), ((64, 64), (64, 64)), ((375, 500), (241, 321)), ((17, 23), (5, 9)),
             ((50, 60), (50, 120)), ((719, 1279), (360, 640)), ((9, 300), (3, 641)), ((2, 2), (7, 5))]
    sizes += [((int(rng.integers(3, 400)), int(rng.integers(3, 400))), (int(rng.integers(2, 400)), int(rng.integers(2, 400))))
              for _ in range(50)]
    resample = getattr(PIL.Image, 'Resampling', PIL.Image).BILINEAR
    for (h, w), (th, tw) in sizes:
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.asarray(PIL.Image.fromarray(im).resize((tw, th), resample))
        got = predictor.resize_bilinear_u8(torch.from_numpy(im), th, tw).numpy()
        assert want.shape == got.shape == (th, tw, 3)
        assert np.array_equal(got, want), '%s -> %s: %d pixels differ' % ((h, w), (th, tw), int((got != want).sum()))


def test_predictor_rescaling_default_and_flag_follow_the_reference():
    """reference predictor.py:17 (``fast_rescaling = True``) and :72-74,81 (``--precise-rescaling`` clears it)."""
    import argparse
    from openpifpaf_amd import Predictor
    assert Predictor.fast_rescaling is True
    parser = argparse.ArgumentParser()
    Predictor.cli(parser)
    try:
        args = parser.parse_args(['--long-edge', '97', '--precise-rescaling'])
        assert args.fast_rescaling is False
        Predictor.configure(args)
        assert Predictor.fast_rescaling is False and Predictor.long_edge == 97
        Predictor.configure(parser.parse_args([]))
        assert Predictor.fast_rescaling is True
    finally:
        Predictor.fast_rescaling, Predictor.long_edge, Predictor.batch_size = True, None, 1


def test_device_preprocess_equals_the_host_path():
    """preprocess_batch_device (rescale + pad + normalise with torch ops, runs on any device) against the host
    preprocess_image in batch mode: same geometry, same meta, the SAME pixels -- with the
    reference's default rescale (Pillow) and with --precise-rescaling (scipy)."""
    import torch
    from openpifpaf_amd import predictor
    rng = np.random.default_rng(5)
    images = [(rng.random((60, 80, 3)) * 255).astype(np.uint8), (rng.random((40, 30, 3)) * 255).astype(np.uint8),
              (rng.random((300, 180, 3)) * 255).astype(np.uint8)]
    batches = {}
    for fast in (True, False):
        batch, metas = predictor.preprocess_batch_device(images, long_edge=97, device=torch.device('cpu'), fast=fast)
        assert batch.shape == (3, 3, 97, 97)
        for b, image in enumerate(images):
            want, wmeta = predictor.preprocess_image(image, long_edge=97, batch_mode=True, fast=fast)
            assert np.allclose(metas[b]['offset'], wmeta['offset']) and np.allclose(metas[b]['scale'], wmeta['scale'])
            assert np.allclose(metas[b]['valid_area'], wmeta['valid_area'])
            assert torch.equal(batch[b], want), (fast, b)
        batches[fast] = batch
    # the two are different filters: close, not equal
    assert not torch.equal(batches[True], batches[False])
    assert float((batches[True][0] - batches[False][0]).abs().mean()) < 0.6


def test_seed_tie_order_setting_and_workspace_view():
    """opa_set_seed_tie_order / opa_get_seed_tie_order (cif_seeds.cpp:94: the order std::sort leaves equal scores in is
    the default), the 'seed_ties' workspace view, and: the pass's scratch lies in regions that are dead while it runs --
    the workspace did not grow by more than the per-image state words."""
    from openpifpaf_amd import _lib, native
    L = _lib.lib()
    assert native.get_seed_tie_order() == 'libstdcxx'
    try:
        native.set_seed_tie_order('index')
        assert L.opa_get_seed_tie_order() == 0 and native.get_seed_tie_order() == 'index'
    finally:
        native.set_seed_tie_order('libstdcxx')
    assert L.opa_get_seed_tie_order() == 1
    with pytest.raises(KeyError):
        native.set_seed_tie_order('stable')
    for shape in (_lib.Shape(32, 17, 19, 81, 81, 81, 81, 8, 8, 128),           # COCO @641
                  _lib.Shape(16, 133, 160, 81, 81, 81, 81, 8, 8, 128),         # wholebody
                  _lib.Shape(2, 17, 19, 9, 11, 9, 11, 8, 8, 16)):
        off, size = ctypes.c_size_t(), ctypes.c_size_t()
        assert L.opa_cifcaf_workspace_view(ctypes.byref(shape), b'seed_ties', ctypes.byref(off), ctypes.byref(size)) == 0
        assert size.value == 4 * shape.batch
        total = L.opa_cifcaf_workspace_bytes_for(ctypes.byref(shape), ctypes.byref(_lib.default_params()))
        assert off.value + size.value <= total
        end_trace = ctypes.c_size_t()
        assert L.opa_cifcaf_workspace_view(ctypes.byref(shape), b'assoc_trace', ctypes.byref(end_trace), ctypes.byref(size)) == 0
        queue, qsize = ctypes.c_size_t(), ctypes.c_size_t()                    # [r6] the association kernel's image queue: B + 1 words
        assert L.opa_cifcaf_workspace_view(ctypes.byref(shape), b'assoc_queue', ctypes.byref(queue), ctypes.byref(qsize)) == 0
        assert queue.value == end_trace.value + size.value                     # right behind the trace ...
        assert 4 * (shape.batch + 1) <= qsize.value < 4 * (shape.batch + 1) + 256
        assert off.value == queue.value + qsize.value                          # ... and the per-image state words right behind the queue
        assert total - (off.value + 4 * shape.batch) < 512                      # (alignment only)
    # the stage-level entry point asks for its scratch: keys + the tie pass's arrays
    cells = 17 * 81 * 81
    assert L.opa_cifseeds_scratch_bytes(1, 17, 81, 81) >= 8 * cells + 16 * cells


def test_split_weight_is_exact_on_the_cpu():
    """``fused.split_weight``: every float32 weight is, bit for bit, the sum of its three bfloat16 pieces (the operand of the
    split-operand GEMM, csrc/gemm_f32x3.hip) -- including denormal-free extremes, zeros and negative numbers."""
    torch = pytest.importorskip('torch')
    from openpifpaf_amd import fused
    torch.manual_seed(0)
    w = torch.randn(64, 128) * torch.logspace(-6, 6, 128)
    w[0, :4] = torch.tensor([0.0, -0.0, 1.0, -1.0])
    w[1, :3] = torch.tensor([3.4028234e38, -1.1754944e-38 * 70000, 1.0 + 2.0 ** -23])
    w3 = fused.split_weight(w)
    assert w3.dtype == torch.bfloat16 and tuple(w3.shape) == (3, 64, 128)
    assert torch.equal((w3[0].float() + w3[1].float()) + w3[2].float(), w)
    assert float(w3[1].float().abs().max() / w.abs().max()) < 2.0 ** -7 and float((w3[2].float().abs() / w.abs().clamp_min(1e-30)).max()) < 2.0 ** -15


def test_pick_takes_the_pinned_choice_without_timing(monkeypatch):
    """``fused.pick`` (the pair / strided-3x3 / stem paths of the float32 trunk): the shipped table decides without a measurement,
    an unknown shape is decided by size where timing is not possible (several ranks), and ``FORCE_PICK`` overrides both."""
    pytest.importorskip('torch')
    from openpifpaf_amd import fused
    table = fused.choices()
    kinds = {k[0] for k in table if k[0].startswith('torch.float32/')}
    assert kinds >= {'torch.float32/pair', 'torch.float32/conv3', 'torch.float32/stem'}, kinds
    assert table[('torch.float32/stem', 3297312, 256, 64, True, False)] == 'x3'            # the bench's batch of 32 at 641 px
    assert table[('torch.float32/conv3', 1681, 4608, 512, True, False)] == 'conv'          # one image, layer 4: MIOpen

    def no_timing(fn, reps=3):
        raise AssertionError('timed')
    monkeypatch.setattr(fused, '_time_ms', no_timing)
    monkeypatch.setattr(fused.torch.cuda, 'is_current_stream_capturing', lambda: False)
    assert fused.pick('stem', 3297312, 256, 64, True, False, lambda: 'x3', lambda: 'other') == 'x3'
    assert fused.pick('conv3', 1681, 4608, 512, True, False, lambda: 'x3', lambda: 'other') == 'other'
    monkeypatch.setattr(fused, '_in_multi_rank_job', lambda: True)
    saved = fused.choices()
    try:
        assert fused.pick('pair', 999999, 192, 256, False, False, lambda: 'x3', lambda: 'other') == 'x3'       # by size
        assert fused.pick('pair', 999, 192, 256, False, False, lambda: 'x3', lambda: 'other') == 'other'
        monkeypatch.setattr(fused, 'FORCE_PICK', 'conv')
        assert fused.pick('stem', 3297312, 256, 64, True, False, lambda: 'x3', lambda: 'other') == 'other'
    finally:
        fused.set_choices(saved, replace=True)
