"""How expensive is decoding the REAL (random-init, all-active) head outputs?"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from openpifpaf_amd import _lib, constants, headmeta, native, network
torch.backends.cudnn.benchmark = True
B = 32
metas = headmeta.cocokp_metas()
m = network.factory('resnet50', list(metas)).cuda()
network.optimize_for_inference_(m)
m = m.to(memory_format=torch.channels_last).to(torch.bfloat16)
x = torch.randn(B, 3, 641, 641, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    cif, caf = m(x)
print(cif.shape, caf.shape, cif.dtype, cif.is_contiguous(), 'conf mean %.3f' % float(cif[:, :, 1].mean()))
skel = torch.from_numpy(np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1)
dec = native.CifCaf(17, skel)
for _ in range(2):
    out, ids, cnt = dec.call_batch(cif, 8, caf, 8)
torch.cuda.synchronize()
_lib.profile_begin(native._stream())
out, ids, cnt = dec.call_batch(cif, 8, caf, 8)
for name, ms in _lib.profile_end():
    print('%-24s %.3f ms' % (name, ms))
print('poses per image', cnt.cpu().tolist()[:8], 'seeds', dec.workspace_view('seed_count', torch.int32)[:8].cpu().tolist())
