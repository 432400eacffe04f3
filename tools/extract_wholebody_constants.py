"""Extracts the wholebody dataset definition (133 keypoint names, 160-bone skeleton,
standing pose, sigmas) as DATA from the reference's
``plugins/wholebody/constants.py:38,71,210,358`` into ``openpifpaf_amd/data/wholebody.json``.
Run in the build container (needs /root/reference).  Values only -- no code is copied."""
import json
import os
import sys
import types

REF = '/root/reference/src/openpifpaf/plugins/wholebody/constants.py'
src = open(REF).read()
# the module imports openpifpaf only for its plotting helpers; stub it
sys.modules['openpifpaf'] = types.ModuleType('openpifpaf')
ns = {'__name__': 'wholebody_constants'}
exec(compile(src, REF, 'exec'), ns)
out = {
    'keypoints': list(ns['WHOLEBODY_KEYPOINTS']),
    'skeleton': [[int(a), int(b)] for a, b in ns['WHOLEBODY_SKELETON']],
    'standing_pose': [[float(v) for v in row[:2]] for row in ns['WHOLEBODY_STANDING_POSE']],
    'sigmas': [float(v) for v in ns['WHOLEBODY_SIGMAS']],
    'score_weights': [float(v) for v in ns['WHOLEBODY_SCORE_WEIGHTS']],
}
assert len(out['keypoints']) == 133 and len(out['skeleton']) == 160 and len(out['standing_pose']) == 133
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'openpifpaf_amd', 'data', 'wholebody.json')
json.dump(out, open(path, 'w'))
print('wrote', path, os.path.getsize(path))
