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


def test_two_ranks_on_one_gpu_equal_a_single_process_decode(tmp_path, coco_skeleton0):
    import torch
    from openpifpaf_amd import native, synth
    dump = str(tmp_path / 'gathered.npz')
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--share-device',
             '--steps', '2', '--warmup', '1', '--decode-only', '--no-cpu-baseline', '--profile-steps', '1',
             '--dump-annotations', dump],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out[-3000:]
    import json
    line = [ln for ln in outs[0].splitlines() if ln.startswith('{')][-1]
    res = json.loads(line)
    assert res['n_gpus'] == 2 and res['config']['global_batch'] == 64 and res['scaling'] == 'weak'
    got = np.load(dump)
    assert got['annotations'].shape[0] == 64 and got['counts'].shape == (64,)
    # single process, all 64 images (rank r of the bench decodes the images seeded r*32 ...)
    cifs, cafs = synth.synth_batch(64, seed0=0)
    dec = native.CifCaf(17, torch.from_numpy(coco_skeleton0))
    out, ids, counts = dec.call_batch(torch.from_numpy(cifs).cuda(), 8, torch.from_numpy(cafs).cuda(), 8)
    out, counts = out.cpu().numpy(), counts.cpu().numpy()
    assert np.array_equal(got['counts'], counts)
    for b in range(64):
        n = int(counts[b])
        assert np.array_equal(got['annotations'][b, :n], out[b, :n]), 'image %d' % b
