"""Does the float32 trunk gain from running two HALF batches on two streams (the HBM-bound layer-1 kernels of one half beside the
MFMA- / power-bound kernels of the other)?  One batch of 32 on one stream against 2 x 16 and 4 x 8 on streams of their own."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from openpifpaf_amd import headmeta, network  # noqa: E402

torch.backends.cudnn.benchmark = True
model = network.factory('resnet50', list(headmeta.cocokp_metas())).cuda()
network.optimize_for_inference_(model)
model = model.to(memory_format=torch.channels_last)
B = 32
x = torch.randn((B, 3, 641, 641), device='cuda').contiguous(memory_format=torch.channels_last)


def run(parts, reps=8, offset=False):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    chunks = list(x.chunk(parts))

    def step():
        cur = torch.cuda.current_stream()
        outs = []
        for s, c in zip(streams, chunks):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(model(c))
        for s in streams:
            cur.wait_stream(s)
        return outs
    with torch.no_grad():
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('%d stream(s) x batch %d: %.2f ms per 32 images = %.1f images/s' % (parts, B // parts, dt * 1e3, B / dt), flush=True)


with torch.no_grad():
    for _ in range(2):
        model(x)
        for c in x.chunk(2):
            model(c)
        for c in x.chunk(4):
            model(c)
run(1)
run(2)
run(4)
run(1)
