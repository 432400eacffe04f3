"""The N > 1 path on real hardware without an 8-GPU node: two processes share the one GPU (gloo for the
collective, every rank on cuda:0), each decodes its shard of the global batch with the HIP path, the packed
annotation blocks are gathered with ONE collective, and rank 0's result must equal a single-process decode
of all 64 images."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_two_ranks(extra, dump):
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--share-device',
             '--no-cpu-baseline', '--profile-steps', '2', '--dump-annotations', dump] + extra,
            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out[-3000:]
    import json
    line = [ln for ln in outs[0].splitlines() if ln.startswith('{')][-1]
    return json.loads(line), np.load(dump)


def test_two_ranks_on_one_gpu_equal_a_single_process_decode(tmp_path, coco_skeleton0):
    import torch
    from openpifpaf_amd import native, synth
    res, got = _run_two_ranks(['--steps', '2', '--warmup', '1', '--decode-only'], str(tmp_path / 'gathered.npz'))
    assert res['n_gpus'] == 2 and res['config']['global_batch'] == 64 and res['scaling'] == 'weak'
    assert res['per_rank_ms_per_step']['min'] <= res['per_rank_ms_per_step']['max'] == res['ms_per_step']
    assert got['annotations'].shape[0] == 64 and got['counts'].shape == (64,)
    # single process, all 64 images (rank r of the bench decodes the images seeded variant * 100000 + r * 32 ...)
    cifs, cafs = synth.synth_batch(64, seed0=int(got['variant']) * int(got['variant_seed']))
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    out, ids, counts = dec.call_batch(torch.from_numpy(cifs).cuda(), 8, torch.from_numpy(cafs).cuda(), 8)
    out, counts = out.cpu().numpy(), counts.cpu().numpy()
    assert np.array_equal(got['counts'], counts)
    for b in range(64):
        n = int(counts[b])
        assert np.array_equal(got['annotations'][b, :n], out[b, :n]), 'image %d' % b


def test_two_ranks_full_step_gather_equals_the_oracle(tmp_path, coco_skeleton0):
    """One FULL step per rank -- network (a small backbone keeps the test short) + decode + the one collective -- with
    rank 0's table of 1x1-convolution kernel choices broadcast after the warm-up; what rank 0 gathers is compared with
    the ORACLE's decode of the global batch, image by image."""
    from openpifpaf_amd import native, synth
    from oracle import port
    from common import compare_annotations
    res, got = _run_two_ranks(['--steps', '2', '--warmup', '1', '--config', '2', '--backbone', 'resnet18', '--batch', '6',
                               '--no-bf16-leg'], str(tmp_path / 'gathered_full.npz'))
    assert res['n_gpus'] == 2 and res['config']['global_batch'] == 12 and res['config']['backbone'] == 'resnet18'
    assert 'DECODE ONLY' not in res['metric'] and res['value'] > 0
    assert got['counts'].shape == (12,)
    variant, vseed = int(got['variant']), int(got['variant_seed'])
    poses = 0
    for r in range(2):
        cifs, cafs = synth.synth_batch(6, seed0=variant * vseed + r * 6)       # rank r's shard
        for i in range(6):
            want, _ = port.decode(cifs[i], 8, cafs[i], 8, coco_skeleton0)
            b = r * 6 + i
            n = int(got['counts'][b]) & native.COUNT_ROWS_MASK
            ok, msg = compare_annotations(got['annotations'][b, :n], want)
            assert ok, 'global image %d: %s' % (b, msg)
            poses += len(want)
    assert poses > 20
