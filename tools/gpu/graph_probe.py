"""Decode as a captured HIP graph: one replay per batch instead of ~10 launches + allocations from Python."""
import time
import numpy as np, torch
from openpifpaf_amd import constants, native, synth
sk = torch.from_numpy(np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1)
cifs, cafs = synth.synth_batch(32, seed0=0)
ct, ft = torch.from_numpy(cifs).cuda(), torch.from_numpy(cafs).cuda()
ref_dec = native.CifCaf(17, sk)
want = ref_dec.call_batch(ct, 8, ft, 8)
torch.cuda.synchronize()
for n_streams in (1, 2, 3, 4):
    lanes = []
    for s in range(n_streams):
        dec = native.CifCaf(17, sk)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(2):
                dec.call_batch(ct, 8, ft, 8)            # warm-up: workspace allocated, header valid
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            out = dec.call_batch(ct, 8, ft, 8)
        lanes.append((dec, st, g, out))
    torch.cuda.synchronize()
    for _, st, g, _ in lanes:
        with torch.cuda.stream(st):
            g.replay()
    torch.cuda.synchronize()
    ok = all(torch.equal(o[2], want[2]) and torch.equal(o[0], want[0]) for _, _, _, o in lanes)
    steps = 200
    t0 = time.perf_counter()
    for i in range(steps):
        _, st, g, _ = lanes[i % n_streams]
        with torch.cuda.stream(st):
            g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%d stream(s), graph replay: %.3f ms/batch -> %.0f images/s  (results equal eager: %s)' % (
        n_streams, dt / steps * 1e3, 32 * steps / dt, ok))
