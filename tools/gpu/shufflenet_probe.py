"""Where does a ShuffleNetV2K forward spend its time? (top kernels by the torch profiler)"""
import os, sys, time
os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
import torch
sys.path.insert(0, '.')
from openpifpaf_amd import headmeta, network
torch.backends.cudnn.benchmark = True
name = sys.argv[1] if len(sys.argv) > 1 else 'shufflenetv2k16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
for dt in (torch.float32, torch.bfloat16):
    m = network.factory(name, list(headmeta.cocokp_metas())).cuda()
    network.optimize_for_inference_(m)
    m = m.to(memory_format=torch.channels_last).to(dt)
    x = torch.randn(B, 3, 641, 641, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        print(name, dt, 'ms/batch %.1f' % ((time.perf_counter() - t0) / 3 * 1e3))
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            m(x)
            torch.cuda.synchronize()
        rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:8]
        for e in rows:
            print('   %-90s n=%3d  %.2f ms' % (e.key[:90], e.count, e.device_time_total / 1e3))
