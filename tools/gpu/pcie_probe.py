"""Decode rate when the caller hands over HOST field tensors (pageable and pinned): H2D upload + decode."""
import time
import numpy as np, torch
from openpifpaf_amd import constants, native, synth
sk = torch.from_numpy(np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1)
cifs, cafs = synth.synth_batch(32, seed0=0)
dec = native.CifCaf(17, sk)
for name, pin in (('pageable', False), ('pinned', True)):
    ct, ft = torch.from_numpy(cifs), torch.from_numpy(cafs)
    if pin:
        ct, ft = ct.pin_memory(), ft.pin_memory()
    for _ in range(3):
        dec.call_batch(ct, 8, ft, 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        out, ids, counts = dec.call_batch(ct, 8, ft, 8)      # uploads, decodes, returns host tensors
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print('%s host fields: %.2f ms per batch of 32 (%.0f images/s, %.1f GB/s of fields over PCIe)' % (
        name, dt * 1e3, 32 / dt, (cifs.nbytes + cafs.nbytes) / dt / 1e9))
