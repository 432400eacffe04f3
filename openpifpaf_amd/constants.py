"""COCO person keypoint constants (values only).

The numbers are the dataset definition the reference ships in
``src/openpifpaf/plugins/coco/constants.py:4-8,23-41,44-62``; they are data, not
code, and have to be identical for a drop-in decoder.
"""
import numpy as np

COCO_KEYPOINTS = [
    'nose', 'left_eye', 'right_eye', 'left_ear', 'right_ear',
    'left_shoulder', 'right_shoulder', 'left_elbow', 'right_elbow',
    'left_wrist', 'right_wrist', 'left_hip', 'right_hip',
    'left_knee', 'right_knee', 'left_ankle', 'right_ankle',
]

# 1-based joint pairs, in CAF field order
COCO_PERSON_SKELETON = [
    (16, 14), (14, 12), (17, 15), (15, 13), (12, 13), (6, 12), (7, 13),
    (6, 7), (6, 8), (7, 9), (8, 10), (9, 11), (2, 3), (1, 2), (1, 3),
    (2, 4), (3, 5), (4, 6), (5, 7),
]

# every plausible joint pair (reference plugins/coco/constants.py:106-126); the "dense" CAF head of
# CifCafDense carries the pairs that are not in the sparse skeleton
DENSER_COCO_PERSON_SKELETON = [
    (1, 2), (1, 3), (2, 3), (1, 4), (1, 5), (4, 5),
    (1, 6), (1, 7), (2, 6), (3, 7),
    (2, 4), (3, 5), (4, 6), (5, 7), (6, 7),
    (6, 12), (7, 13), (6, 13), (7, 12), (12, 13),
    (6, 8), (7, 9), (8, 10), (9, 11), (6, 10), (7, 11),
    (8, 9), (10, 11),
    (10, 12), (11, 13),
    (10, 14), (11, 15),
    (14, 12), (15, 13), (12, 15), (13, 14),
    (12, 16), (13, 17),
    (16, 14), (17, 15), (14, 17), (15, 16),
    (14, 15), (16, 17),
]
DENSER_COCO_PERSON_CONNECTIONS = [c for c in DENSER_COCO_PERSON_SKELETON if c not in
                                  [tuple(b) for b in COCO_PERSON_SKELETON]]

COCO_PERSON_SIGMAS = [
    0.026, 0.025, 0.025, 0.035, 0.035, 0.079, 0.079, 0.072, 0.072,
    0.062, 0.062, 0.107, 0.107, 0.087, 0.087, 0.089, 0.089,
]

COCO_PERSON_SCORE_WEIGHTS = [3.0] * 3 + [1.0] * (len(COCO_KEYPOINTS) - 3)

# x, y (y up) in "pose units"; a standing person is about 10 units tall
COCO_UPRIGHT_POSE = np.array([
    [0.0, 9.3], [-0.35, 9.7], [0.35, 9.7], [-0.7, 9.5], [0.7, 9.5],
    [-1.4, 8.0], [1.4, 8.0], [-1.75, 6.0], [1.75, 6.2], [-1.75, 4.0],
    [1.75, 4.2], [-1.26, 4.0], [1.26, 4.0], [-1.4, 2.0], [1.4, 2.1],
    [-1.4, 0.0], [1.4, 0.1],
], dtype=np.float64)


def wholebody():
    """The wholebody (133 keypoints, 160 bones) dataset definition, as data extracted from the
    reference's ``plugins/wholebody/constants.py:38,71,210,358`` by
    ``tools/extract_wholebody_constants.py``.  Returns a dict with ``keypoints``,
    ``skeleton`` (1-based), ``standing_pose`` [133,2], ``sigmas``, ``score_weights``."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'wholebody.json')
    with open(path) as f:
        d = json.load(f)
    d['standing_pose'] = np.asarray(d['standing_pose'], dtype=np.float64)
    return d
