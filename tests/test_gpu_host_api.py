"""Host-side API on the GPU: Predictor / Decoder.batch keep the fields on the device and
return the same annotations as decoding the same head outputs image by image with the oracle."""
import sys
import types

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def test_decoder_batch_equals_oracle_on_model_fields(coco_skeleton0):
    """decoder.CifCaf.batch(model, images): fields never leave the device; results must equal the
    oracle's decode of the very same head outputs."""
    from openpifpaf_amd import decoder, headmeta, synth
    from oracle import port
    cifs, cafs = synth.synth_batch(4, seed0=700, height=41, width=41, people=(2, 4), size_range=(0.6, 0.95))

    class FieldModel:                       # stands in for a trained Shell: emits known fields
        def __call__(self, images):
            return (torch.from_numpy(cifs).cuda(), torch.from_numpy(cafs).cuda())

    metas = headmeta.cocokp_metas()
    dec = decoder.factory(list(metas))
    assert isinstance(dec, decoder.Multi) and isinstance(dec.decoders[0], decoder.CifCaf)
    images = torch.zeros((4, 3, 321, 321))
    result = dec.batch(FieldModel(), images, device=torch.device('cuda'))
    assert len(result) == 4 and dec.last_decoder_time > 0 and dec.last_nn_time >= 0
    for b in range(4):
        want, _ = port.decode(cifs[b], 8, cafs[b], 8, coco_skeleton0)
        assert len(result[b]) == len(want) >= 1
        for ann, w in zip(result[b], want):
            assert np.allclose(ann.data[:, 0], w[:, 1], atol=1e-4) and np.allclose(ann.data[:, 1], w[:, 2], atol=1e-4)
            assert np.allclose(ann.data[:, 2], w[:, 0], atol=1e-4) and np.allclose(ann.joint_scales, w[:, 3], atol=1e-4)
    # single-image entry point, fields as a per-image list like the reference's Decoder.__call__
    anns = dec([torch.from_numpy(cifs[1]).cuda(), torch.from_numpy(cafs[1]).cuda()])
    assert len(anns) == len(result[1])


def test_predictor_end_to_end_smoke():
    """Random-init network: nothing to compare against, but the whole chain must run on the device
    (preprocess -> backbone -> heads -> HIP decode -> Annotation objects / JSON)."""
    from openpifpaf_amd import Predictor
    Predictor.long_edge = 321
    Predictor.batch_size = 2
    try:
        pred = Predictor('resnet18', json_data=True)
        rng = np.random.default_rng(0)
        images = [(rng.random((240, 320, 3)) * 255).astype(np.uint8) for _ in range(3)]
        out = list(pred.numpy_images(images))
        assert len(out) == 3
        for anns, _, meta in out:
            assert isinstance(anns, list) and 'offset' in meta
            for a in anns:
                assert set(a) >= {'keypoints', 'bbox', 'score', 'category_id'}
        assert pred.total_images == 3 and pred.total_decoder_time > 0
    finally:
        Predictor.long_edge = None
        Predictor.batch_size = 1


def test_cifcafdense_decodes_concatenated_heads_like_the_oracle():
    """CifCafDense (reference decoder/cifcaf.py:17-78): sparse + dense CAF heads go through the association
    kernel as ONE 44-bone head (2A = 88 directed bones: the LDS variant of the growth state)."""
    from openpifpaf_amd import constants, decoder, headmeta, synth
    from oracle import port
    skeleton = list(constants.COCO_PERSON_SKELETON) + list(constants.DENSER_COCO_PERSON_CONNECTIONS)
    skeleton0 = np.asarray(skeleton, dtype=np.int64) - 1
    metas = headmeta.cocokp_dense_metas()
    decoder.CifCafDense.dense_coupling = 1.0
    try:
        decs = decoder.CifCafDense.factory(list(metas))
        assert len(decs) == 1 and decs[0].priority > decoder.CifCaf.factory(list(metas))[0].priority
        dec = decs[0]
        cifs, cafs = [], []
        for seed, people in ((81, 3), (82, 7)):
            cif, caf = synth.synth_fields(seed, people, height=49, width=49, skeleton=skeleton)
            assert caf.shape[0] == 44
            cifs.append(cif); cafs.append(caf)
            fields = [torch.from_numpy(cif).cuda(), torch.from_numpy(caf[:19]).cuda(), torch.from_numpy(caf[19:]).cuda()]
            anns = dec(fields)
            want, _ = port.decode(cif, 8, caf, 8, skeleton0)
            assert len(anns) == len(want) >= 1
            for ann, w in zip(anns, want):
                assert np.allclose(ann.data[:, 2], w[:, 0], atol=1e-4) and np.allclose(ann.data[:, :2], w[:, 1:3], atol=1e-4)
                assert len(ann.skeleton) == 44

        class Heads:                          # a model that emits the three heads for a batch
            def __call__(self, images):
                c, a = torch.from_numpy(np.stack(cifs)).cuda(), torch.from_numpy(np.stack(cafs)).cuda()
                return (c, a[:, :19].contiguous(), a[:, 19:].contiguous())

        result = dec.batch(Heads(), torch.zeros((2, 3, 385, 385)), device=torch.device('cuda'))
        assert [len(r) for r in result] == [len(port.decode(c, 8, a, 8, skeleton0)[0]) for c, a in zip(cifs, cafs)]
    finally:
        decoder.CifCafDense.dense_coupling = 0.0


def test_captured_decode_graph_replays_equal_eager_results(coco_skeleton0):
    """native.CifCaf.capture: the decode as one HIP graph; refilling the static input tensors and replaying
    must give what an eager call gives (workspace state -- the lazily cleared CifHr map -- included)."""
    from openpifpaf_amd import native, synth
    eager = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    a = synth.synth_batch(4, seed0=300, height=49, width=49, people=(6, 1, 3, 9))
    b = synth.synth_batch(4, seed0=400, height=49, width=49, people=(1, 8, 2, 0))
    cif_t, caf_t = torch.from_numpy(a[0]).cuda(), torch.from_numpy(a[1]).cuda()
    graph, (out, ids, counts) = dec.capture(cif_t, 8, caf_t, 8)
    for fields in (b, a, b):
        cif_t.copy_(torch.from_numpy(fields[0]))
        caf_t.copy_(torch.from_numpy(fields[1]))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        want = eager.call_batch(torch.from_numpy(fields[0]).cuda(), 8, torch.from_numpy(fields[1]).cuda(), 8)
        assert torch.equal(counts, want[2])
        for i in range(4):
            n = int(counts[i])
            assert torch.equal(out[i, :n], want[0][i, :n])


def test_predictor_with_device_side_preprocessing():
    """Predictor.device_preprocess: uint8 frames go to the GPU as they are and are rescaled there with the reference's
    own arithmetic: the network input equals the host path's pixel for pixel, and so do the annotations."""
    from openpifpaf_amd import Predictor, predictor
    rng = np.random.default_rng(1)
    images = [(rng.random((120, 160, 3)) * 255).astype(np.uint8), (rng.random((150, 90, 3)) * 255).astype(np.uint8)]
    results = {}
    for on_device in (True, False):
        Predictor.long_edge, Predictor.batch_size, Predictor.device_preprocess = 193, 2, on_device
        try:
            pred = Predictor('resnet18')
            results[on_device] = list(pred.numpy_images(images))
            assert len(results[on_device]) == 2 and pred.total_images == 2
        finally:
            Predictor.long_edge, Predictor.batch_size, Predictor.device_preprocess = None, 1, False
    for fast in (True, False):       # the reference's default (Pillow bilinear) and --precise-rescaling (scipy zoom)
        batch, metas = predictor.preprocess_batch_device(images, long_edge=193, device=torch.device('cuda'), fast=fast)
        assert batch.is_cuda and batch.shape == (2, 3, 193, 193)
        for b, image in enumerate(images):
            want, wmeta = predictor.preprocess_image(image, long_edge=193, batch_mode=True, fast=fast)
            assert np.allclose(metas[b]['offset'], wmeta['offset']) and np.allclose(metas[b]['scale'], wmeta['scale'])
            assert torch.equal(batch[b].cpu(), want), 'device and host preprocessing (fast=%s) differ in %d values' % (
                fast, int((batch[b].cpu() != want).sum()))
    for (pred_d, _, meta_d), (pred_h, _, meta_h) in zip(results[True], results[False]):
        assert len(pred_d) == len(pred_h)
        for a_d, a_h in zip(pred_d, pred_h):
            assert np.array_equal(a_d.data, a_h.data)
    # the inverse transform ran on the device tensor (decoder.batch(meta_batch=...)); on the host, annotation by
    # annotation like the reference (annotation.py:162-200), it gives the same coordinates
    Predictor.long_edge, Predictor.batch_size = 193, 2
    try:
        pred = Predictor('resnet18')
        pred.processor.supports_device_inverse = False
        host_inverse = list(pred.numpy_images(images))
    finally:
        Predictor.long_edge, Predictor.batch_size = None, 1
    for (pred_d, _, _), (pred_h, _, _) in zip(results[False], host_inverse):
        assert len(pred_d) == len(pred_h)
        for a_d, a_h in zip(pred_d, pred_h):
            assert np.allclose(a_d.data, a_h.data, rtol=0, atol=1e-4) and np.allclose(a_d.joint_scales, a_h.joint_scales, atol=1e-4)


def test_pipelined_batches_equal_synchronous_ones_and_the_oracle(coco_skeleton0):
    """decoder.CifCaf.batch_async / --decoder-workers: several batches in flight over decode lanes
    (stream + workspace each), annotations through pinned memory.  Every batch must come back as the synchronous
    ``batch`` returns it and as the oracle decodes the same fields -- whatever the order in which the results are
    collected, with 1, 2 and 3 lanes, with and without the device-side inverse transform."""
    from openpifpaf_amd import decoder, headmeta, synth
    from oracle import port
    batches = [synth.synth_batch(3, seed0=900 + 10 * i, height=41, width=41, people=(1 + i, 5, 2), size_range=(0.6, 0.95))
               for i in range(5)]

    class FieldModel:                       # emits the field batch the images name (their first value)
        def __call__(self, images):
            i = int(images[0, 0, 0, 0].item())
            return (torch.from_numpy(batches[i][0]).cuda(), torch.from_numpy(batches[i][1]).cuda())

    def images(i):
        return torch.full((3, 3, 321, 321), float(i))

    def as_rows(result):
        return [[(a.data.copy(), a.joint_scales.copy()) for a in anns] for anns in result]

    metas = [{'offset': np.array([3.0, -2.0]), 'scale': np.array([0.5, 0.5]), 'hflip': False, 'width_height': np.array([640, 640])}
             for _ in range(3)]
    old = decoder.CifCaf.decoder_workers
    try:
        for workers in (1, 2, 3):
            decoder.CifCaf.decoder_workers = workers
            dec = decoder.factory(list(headmeta.cocokp_metas()))
            assert dec.pipeline_depth == workers
            sync = [as_rows(dec.batch(FieldModel(), images(i), device=torch.device('cuda'))) for i in range(5)]
            for i in range(5):
                for b in range(3):
                    want, _ = port.decode(batches[i][0][b], 8, batches[i][1][b], 8, coco_skeleton0)
                    assert len(sync[i][b]) == len(want) >= 1
                    for (data, scales), w in zip(sync[i][b], want):
                        assert np.allclose(data[:, :2], w[:, 1:3], atol=1e-4) and np.allclose(data[:, 2], w[:, 0], atol=1e-4)
            # in flight: as many as there are lanes, collected oldest first
            pend, got = [], {}
            for i in range(5):
                if len(pend) >= workers:
                    j, p = pend.pop(0)
                    got[j] = as_rows(p.result())
                pend.append((i, dec.batch_async(FieldModel(), images(i), device=torch.device('cuda'))))
            for j, p in reversed(pend):      # the rest, newest first
                got[j] = as_rows(p.result())
            for i in range(5):
                assert len(got[i]) == 3
                for b in range(3):
                    assert len(got[i][b]) == len(sync[i][b])
                    for (d0, s0), (d1, s1) in zip(got[i][b], sync[i][b]):
                        assert np.array_equal(d0, d1) and np.array_equal(s0, s1), (workers, i, b)
            # a lane that is submitted to again before its batch was collected keeps that batch's result
            first = dec.batch_async(FieldModel(), images(0), device=torch.device('cuda'))
            later = [dec.batch_async(FieldModel(), images(1 + k), device=torch.device('cuda')) for k in range(workers)]
            assert [len(r) for r in first.result()] == [len(r) for r in sync[0]]
            for k, p in enumerate(later):
                assert [len(r) for r in p.result()] == [len(r) for r in sync[1 + k]]
            # with the metas of the preprocessing: the inverse transform runs on the lane as well
            a = as_rows(dec.batch(FieldModel(), images(2), device=torch.device('cuda'), meta_batch=metas))
            b_ = as_rows(dec.batch_async(FieldModel(), images(2), device=torch.device('cuda'), meta_batch=metas).result())
            assert len(a) == len(b_) == 3
            for x, y in zip(a, b_):
                for (d0, s0), (d1, s1) in zip(x, y):
                    assert np.array_equal(d0, d1) and np.array_equal(s0, s1)
    finally:
        decoder.CifCaf.decoder_workers = old


def test_pipelined_predictor_equals_the_synchronous_one():
    """Predictor._images pipelines its batches through the decoder's lanes by default; the predictions are the ones the
    synchronous loop gives (same network, same kernels, other streams)."""
    from openpifpaf_amd import Predictor
    Predictor.long_edge, Predictor.batch_size = 193, 2
    try:
        pred = Predictor('resnet18', json_data=True)
        rng = np.random.default_rng(5)
        images = [(rng.random((150 + 10 * k, 200, 3)) * 255).astype(np.uint8) for k in range(7)]
        pred.pipelined = False
        want = list(pred.numpy_images(images))
        pred.pipelined = True
        got = list(pred.numpy_images(images))
        assert len(got) == len(want) == 7 and pred.total_images == 14
        for (g, _, gm), (w, _, wm) in zip(got, want):
            assert g == w and np.array_equal(gm['offset'], wm['offset'])
    finally:
        Predictor.long_edge, Predictor.batch_size = None, 1
