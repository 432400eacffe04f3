"""cifhr tile kernel time per call: first call on a workspace (all tiles) vs. later calls (touched tiles only)."""
import numpy as np, torch
from openpifpaf_amd import _lib, constants, native, synth
sk = torch.from_numpy(np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1)
cifs, cafs = synth.synth_batch(32, seed0=0)
ct, ft = torch.from_numpy(cifs).cuda(), torch.from_numpy(cafs).cuda()
dec = native.CifCaf(17, sk)
for call in range(5):
    _lib.profile_begin(native._stream())
    dec.call_batch(ct, 8, ft, 8)
    t = dict(_lib.profile_end())
    hdr = dec._last[1][:32].view(torch.int64).cpu().tolist()
    print('call %d: cif_active %.3f ms, cifhr_tile %.3f ms, header %s' % (call, t['cif_active_kernel'], t['cifhr_tile_kernel'], [hex(h & (2**64 - 1)) for h in hdr[:3]]))
for people in (1, 5, 20):
    cifs, cafs = synth.synth_batch(32, seed0=0, people=(people,))
    ct, ft = torch.from_numpy(cifs).cuda(), torch.from_numpy(cafs).cuda()
    for call in range(3):
        _lib.profile_begin(native._stream())
        dec.call_batch(ct, 8, ft, 8)
        t = dict(_lib.profile_end())
    print('people %2d: cif_active %.3f ms, cifhr_tile %.3f ms, assoc %.3f' % (people, t['cif_active_kernel'], t['cifhr_tile_kernel'], t['cifcaf_assoc_kernel']))
