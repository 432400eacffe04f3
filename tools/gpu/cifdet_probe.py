"""CifDet decode (box detection fields) throughput: batch 32, 80 categories, 81x81 fields, vs the oracle on CPU."""
import time
import numpy as np, torch
from openpifpaf_amd import native, synth
from oracle import port
B, F = 32, 80
fields = np.stack([synth.synth_det_field(100 + b, 3 + b % 12, n_categories=F, height=81, width=81) for b in range(B)])
ft = torch.from_numpy(fields).cuda()
dec = native.CifDet()
for _ in range(3):
    out = dec.call_batch(ft, 8)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    cat, sc, bx, cnt = dec.call_batch(ft, 8)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
t1 = time.perf_counter()
for b in range(8):
    port.cifdet_decode(fields[b], 8)
cpu = (time.perf_counter() - t1) / 8
print('CifDet decode: %.3f ms per batch of %d (%d categories) -> %.0f images/s; oracle on 1 CPU thread %.1f ms/image -> %.0f images/s; detections %s' % (
    dt * 1e3, B, F, B / dt, cpu * 1e3, 1 / cpu, cnt.tolist()[:8]))
