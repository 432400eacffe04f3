"""Round 6: measures the 1x1-convolution kernel choice (fused.conv_bias_act: MFMA GEMM / epilogue pass + GEMM / MIOpen) for the shapes
of the BASELINE configurations on this GPU and writes the table the package ships (openpifpaf_amd/conv1x1_pinned.json), so that the
ranks of a multi-GPU job run the same kernels without a collective.

    python tools/gpu/dump_conv_choices.py [out.json]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpifpaf_amd import fused, headmeta, network  # noqa: E402

torch.backends.cudnn.benchmark = True
out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/conv1x1_pinned.json'
fused.set_choices({}, replace=True)
cases = [('resnet50', headmeta.cocokp_metas, 32, 641), ('resnet50', headmeta.cocokp_metas, 1, 641), ('resnet50', headmeta.cocokp_metas, 16, 641),
         ('resnet50', headmeta.cocokp_metas, 8, 641), ('resnet50', headmeta.cocokp_metas, 4, 641), ('resnet50', headmeta.cocokp_metas, 2, 641),
         ('resnet18', headmeta.cocokp_metas, 1, 321), ('shufflenetv2k16', headmeta.cocokp_metas, 32, 641),
         ('shufflenetv2k30', headmeta.wholebody_metas, 16, 641)]
for name, metas, B, edge in cases:
    for dtype in (torch.float32, torch.bfloat16):
        model = network.factory(name, list(metas())).cuda()
        network.optimize_for_inference_(model)
        model = model.to(memory_format=torch.channels_last)
        if dtype != torch.float32:
            model = model.to(dtype)
        x = torch.randn((B, 3, edge, edge), device='cuda', dtype=dtype).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            for _ in range(2):
                model(x)
        torch.cuda.synchronize()
        print(name, B, edge, dtype, len(fused.choices()), flush=True)
        del model, x
        torch.cuda.empty_cache()
table = [list(k) + [v] for k, v in sorted(fused.choices().items(), key=lambda kv: str(kv[0]))]
os.makedirs(os.path.dirname(out) or '.', exist_ok=True)
json.dump({'device': torch.cuda.get_device_name(0), 'columns': ['dtype', 'M', 'K', 'N', 'residual', 'a_bias', 'choice'], 'table': table},
          open(out, 'w'), indent=0)
print('wrote', out, len(table), 'entries')
