"""Decode the self-serve test's cases through a diagnostic library and print what the watchdog dump says."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpifpaf_amd import _lib, constants, native, synth
print(_lib.LIB_PATH)
skel0 = np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1
rng = np.random.default_rng(11)
cases = [synth.synth_fields(7000 + i, int(rng.integers(2, 12)), height=57, width=65,
                            cif_noise=float(rng.choice([0.2, 0.4, 0.7])), size_range=(0.3, 0.8)) for i in range(8)]
cases += [synth.synth_fields(7100 + i, int(rng.integers(8, 22)), height=57, width=65) for i in range(8)]
cifs = np.stack([c for c, _ in cases]); cafs = np.stack([f for _, f in cases])
for growers in ('1', None, '2', '1', '3', '1'):
  if growers: os.environ['OPA_ASSOC_GROWERS'] = growers
  else: os.environ.pop('OPA_ASSOC_GROWERS', None)
  print('growers', growers)
  dec = native.CifCaf(17, torch.from_numpy(skel0))
  for rep in range(6):
      out, ids, counts = dec.call_batch(torch.from_numpy(cifs).cuda(), 8, torch.from_numpy(cafs).cuda(), 8)
      counts = counts.cpu().numpy()
      bad = [b for b in range(len(cases)) if counts[b] < 0 or counts[b] >= (1 << 29)]
      print('rep', rep, 'counts', counts.tolist())
      if bad:
          stats = dec.assoc_stats().cpu().numpy()
          tr = dec.workspace_view('assoc_trace', torch.int32)[:len(cases) * 64 * 4].view(len(cases), 64, 4).cpu().numpy()
          for b in bad:
              print('image', b, 'stats', stats[b].tolist())
              print('  tasks (state seed cancel npub|ack<<16):', tr[b, 40:52].tolist())
              print('  hd scan_pos n_live epoch:', tr[b, 60].tolist(), ' head slot (idx own ep shadow-by mask):', tr[b, 61].tolist(), ' sh_ctl12 sh_ctl9 iter commits:', tr[b, 62].tolist())
          break
