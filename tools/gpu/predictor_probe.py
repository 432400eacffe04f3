"""Where a pipelined Predictor batch's time goes: host seconds per phase (collect / preprocess / submit) and the network's
GPU time per batch (events on the main stream), next to the wall time per batch.

    python tools/gpu/predictor_probe.py [--batches 16]
"""
import argparse
import os
import sys
import time

import numpy as np

os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpifpaf_amd import Predictor, headmeta, network, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batches', type=int, default=16)
ap.add_argument('--batch', type=int, default=32)
args = ap.parse_args()
torch.backends.cudnn.benchmark = True
device = torch.device('cuda', 0)
B = args.batch
metas = headmeta.cocokp_metas()
net = network.factory('resnet50', list(metas)).to(device)
network.optimize_for_inference_(net)
net = net.to(memory_format=torch.channels_last)
variants = []
for v in range(2):
    cifs, cafs = synth.synth_batch(B, seed0=v * 100000)
    variants.append((torch.from_numpy(cifs).to(device), torch.from_numpy(cafs).to(device)))
net_events = []


class Injected(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net, self.head_metas, self.n = net, list(metas), 0

    def forward(self, x):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.net(x)
        e1.record()
        net_events.append((e0, e1))
        v = variants[self.n % 2]
        self.n += 1
        return v


Predictor.batch_size, Predictor.long_edge, Predictor.device_preprocess, Predictor.device = B, 641, True, device
pred = Predictor(model=Injected())
rng = np.random.default_rng(3)
frames = [rng.integers(0, 255, (641, 641, 3), dtype=np.uint8) for _ in range(B)]
list(pred.numpy_images(frames * 3))
torch.cuda.synchronize()
net_events.clear()

# the pipelined loop of Predictor._images, instrumented
depth = pred.processor.pipeline_depth
t_pre = t_sub = t_col = 0.0
in_flight = []
t0 = time.perf_counter()
for i in range(args.batches):
    if len(in_flight) >= depth:
        t = time.perf_counter()
        in_flight.pop(0)[0]()
        t_col += time.perf_counter() - t
    t = time.perf_counter()
    batch, m = pred._preprocess(frames)
    t_pre += time.perf_counter() - t
    t = time.perf_counter()
    in_flight.append((pred.tensor_batch_async(batch, m), m))
    t_sub += time.perf_counter() - t
for c, _ in in_flight:
    t = time.perf_counter()
    c()
    t_col += time.perf_counter() - t
torch.cuda.synchronize()
wall = time.perf_counter() - t0
n = args.batches
gpu_net = [a.elapsed_time(b) for a, b in net_events]
gaps = [net_events[k][1].elapsed_time(net_events[k + 1][0]) for k in range(len(net_events) - 1)]
print('lanes %d; wall %.2f ms per batch (%.1f images/s)' % (depth, wall / n * 1e3, B * n / wall))
print('host per batch: preprocess %.2f ms, submit (network + decode queued) %.2f ms, collect %.2f ms' % (
    t_pre / n * 1e3, t_sub / n * 1e3, t_col / n * 1e3))
print('network on the GPU: %.2f ms per batch (min %.2f, max %.2f); idle gaps between networks on the main stream: mean %.2f ms, max %.2f ms' % (
    float(np.mean(gpu_net)), min(gpu_net), max(gpu_net), float(np.mean(gaps)), max(gaps)))
