"""Times backbone variants (dtype x memory format x MIOpen find mode) on the GPU box."""
import itertools
import sys
import time

import torch

sys.path.insert(0, '.')
from openpifpaf_amd import network

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda')


def run(dtype, cl, bench, fuse=True, name='resnet50', graph=False):
    torch.backends.cudnn.benchmark = bench
    m = network.factory(name).to(dev)
    if fuse:
        network.fuse_conv_bn_(m)
    if cl:
        m = m.to(memory_format=torch.channels_last)
    m = m.to(dtype)
    x = torch.randn(B, 3, 641, 641, device=dev, dtype=dtype)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        if graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                y = m(x)
            fn = g.replay
        else:
            fn = lambda: m(x)
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print('%-16s dtype=%-8s channels_last=%-5s find=%-5s fuse=%-5s graph=%-5s  %.1f ms/batch  %.1f img/s' % (
        name, str(dtype).split('.')[-1], cl, bench, fuse, graph, dt * 1e3, B / dt), flush=True)


for dtype, cl, bench in itertools.product((torch.bfloat16, torch.float16), (True, False), (False, True)):
    try:
        run(dtype, cl, bench)
    except Exception as e:
        print('failed', dtype, cl, bench, repr(e)[:200])
run(torch.float32, True, True)
run(torch.bfloat16, True, True, fuse=False)
try:
    run(torch.bfloat16, True, True, graph=True)
except Exception as e:
    print('graph failed', repr(e)[:300])
run(torch.bfloat16, True, True, name='shufflenetv2k16')
