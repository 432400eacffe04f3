#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
{
echo "=== probe (default waves)"; timeout 300 python tools/gpu/assoc_probe.py 2>&1 | grep -v amdgpu.ids
for w in 8 12; do echo "=== probe waves=$w"; OPA_ASSOC_WAVES=$w timeout 120 python tools/gpu/assoc_probe.py 2>&1 | grep -v amdgpu.ids | tail -12; done
echo "=== head output statistics of random-init networks (bf16)"
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, sys
sys.path.insert(0, '.')
from openpifpaf_amd import headmeta, network
cm, fm = headmeta.cocokp_metas()
for name, size in (('resnet18', (129, 161)), ('resnet50', (257, 321))):
    torch.manual_seed(7)
    m = network.factory(name, [cm, fm]).cuda().eval().to(torch.bfloat16)
    x = torch.randn((2, 3) + size, generator=torch.Generator().manual_seed(3)).cuda().to(torch.bfloat16)
    with torch.no_grad():
        h = m(x)
    c = h[0].float()
    print(name, tuple(c.shape), 'conf min/mean/max %.3f %.3f %.3f' % (c[:, :, 1].min(), c[:, :, 1].mean(), c[:, :, 1].max()),
          'scale mean %.3f' % c[:, :, 4].mean(), 'distinct conf values', c[:, :, 1].unique().numel())
PY
} > gpurun_out/call2_probe.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity_r2.py -q -m gpu -s > gpurun_out/call2_r2tests.log 2>&1
echo "r2 tests rc=$?" >> gpurun_out/call2_probe.log
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/call2_gputests.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/call2_probe.log
tail -n 8 gpurun_out/call2_r2tests.log; tail -n 8 gpurun_out/call2_gputests.log
cat gpurun_out/call2_probe.log
