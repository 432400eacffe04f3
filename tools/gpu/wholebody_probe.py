"""BASELINE config 4 shapes: 133 keypoints / 160 bones, 81x81 fields, batch 16: HIP decode vs the
reference CPU decoder on the same fields."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from openpifpaf_amd import _lib, constants, native, synth
from oracle import reference
wb = constants.wholebody()
skel0 = np.asarray(wb['skeleton'], dtype=np.int64) - 1
B = 16
people = (1, 3, 6, 10)
cifs, cafs = synth.synth_batch(B, seed0=0, people=people, pose=wb['standing_pose'], skeleton=wb['skeleton'])
print('fields', cifs.shape, cafs.shape, '%.0f MB' % ((cifs.nbytes + cafs.nbytes) / 1e6))
dec = native.CifCaf(133, torch.from_numpy(skel0))
cif_d, caf_d = torch.from_numpy(cifs).cuda(), torch.from_numpy(cafs).cuda()
for _ in range(2):
    out, ids, cnt = dec.call_batch(cif_d, 8, caf_d, 8)
torch.cuda.synchronize()
acc = {}
for _ in range(5):
    _lib.profile_begin(native._stream())
    out, ids, cnt = dec.call_batch(cif_d, 8, caf_d, 8)
    for name, ms in _lib.profile_end():
        acc.setdefault(name, []).append(ms)
tot = 0.0
for k, v in acc.items():
    print('%-24s %.3f ms' % (k, np.mean(v))); tot += np.mean(v)
print('decode %.3f ms per batch of %d -> %.0f images/s; poses %s; workspace %.2f GB' % (
    tot, B, B / tot * 1e3, cnt.cpu().tolist(), dec._last[1].numel() / 1e9))
torch_ = reference.load(); torch_.set_num_threads(1); reference.reset_statics()
t0 = time.perf_counter(); n = 0
for b in range(B):
    r, _, _ = reference.decode(cifs[b], 8, cafs[b], 8, skel0); n += 1
    g = out[b, :int(cnt[b])].cpu().numpy()
    assert g.shape == r.shape and (np.abs(g - r).max() if g.size else 0) <= 1e-4, b
    if time.perf_counter() - t0 > 40: break
dt = time.perf_counter() - t0
print('reference CPU decoder (1 thread): %.1f ms/image over %d images -> %.1f images/s; GPU/CPU = %.0fx; parity vs REFERENCE ok' % (
    dt / n * 1e3, n, n / dt, (B / tot * 1e3) / (n / dt)))
