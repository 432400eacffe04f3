"""Seeded "COCO-shaped" composite-field synthesiser.

Random-initialised networks emit structureless fields (every cell active), and
there are no checkpoints offline, so decode parity and throughput are measured
on synthetic CIF/CAF tensors that have the statistics of a trained network's
output: a few people per image, a confidence blob per keypoint and per bone,
regressions that point at the true joints with a little noise.

Layout follows CompositeField4 (reference ``network/heads.py:290,360-378``):
CIF  ``[F, 5, H, W]``: 0 unused, 1 confidence, 2 x, 3 y (absolute, field units), 4 scale
CAF  ``[A, 8, H, W]``: 0 unused, 1 confidence, 2..3 x1 y1, 4..5 x2 y2, 6 s1, 7 s2
"""
import numpy as np

from . import constants


def _blob(conf_plane, targets, cx, cy, radius, peak, sigma, rng):
    """Paint one confidence blob; returns the (j, i) cells it touched."""
    H, W = conf_plane.shape
    i0, i1 = max(0, int(np.floor(cx - radius))), min(W - 1, int(np.ceil(cx + radius)))
    j0, j1 = max(0, int(np.floor(cy - radius))), min(H - 1, int(np.ceil(cy + radius)))
    if i1 < i0 or j1 < j0:
        return np.zeros((0,), np.int64), np.zeros((0,), np.int64)
    jj, ii = np.meshgrid(np.arange(j0, j1 + 1), np.arange(i0, i1 + 1), indexing='ij')
    d2 = (ii - cx) ** 2 + (jj - cy) ** 2
    inside = d2 <= radius * radius
    jj, ii, d2 = jj[inside], ii[inside], d2[inside]
    c = peak * np.exp(-d2 / (2.0 * sigma * sigma))
    c = c * (1.0 - 0.02 * rng.random(c.shape))  # break exact ties
    better = c > conf_plane[jj, ii]
    jj, ii, c = jj[better], ii[better], c[better]
    conf_plane[jj, ii] = c
    return jj, ii


def synth_fields(seed, n_people, *, height=81, width=81,
                 pose=None, skeleton=None, noise=0.05, size_range=(0.25, 0.75), cif_noise=None):
    """Generate one image's (cif, caf) float32 field tensors.

    :param seed: numpy ``default_rng`` seed, fully determines the output
    :param n_people: number of synthetic people
    :param pose: ``[K, 2]`` pose template (x right, y up), default COCO upright
    :param skeleton: 1-based bone list, default COCO person skeleton
    :param size_range: person height as a fraction of min(height, width)
    :param cif_noise: regression noise of the CIF blobs alone (default ``noise``): large values spread the
        seeds of one confidence blob over several occupancy boxes (wrong blob-mate predictions in the
        association kernel) while the CAF fields still connect the poses
    """
    if cif_noise is None:
        cif_noise = noise
    rng = np.random.default_rng(seed)
    if pose is None:
        pose = constants.COCO_UPRIGHT_POSE
    if skeleton is None:
        skeleton = constants.COCO_PERSON_SKELETON
    pose = np.asarray(pose, dtype=np.float64)[:, :2]
    K, A = len(pose), len(skeleton)
    H, W = height, width

    jj, ii = np.meshgrid(np.arange(H, dtype=np.float64),
                         np.arange(W, dtype=np.float64), indexing='ij')

    cif = np.empty((K, 5, H, W), dtype=np.float64)
    cif[:, 0] = rng.normal(0.0, 1.0, (K, H, W))
    cif[:, 1] = rng.uniform(0.0, 0.05, (K, H, W))
    cif[:, 2] = ii[None] + rng.normal(0.0, 1.0, (K, H, W))
    cif[:, 3] = jj[None] + rng.normal(0.0, 1.0, (K, H, W))
    cif[:, 4] = rng.uniform(0.1, 1.0, (K, H, W))

    caf = np.empty((A, 8, H, W), dtype=np.float64)
    caf[:, 0] = rng.normal(0.0, 1.0, (A, H, W))
    caf[:, 1] = rng.uniform(0.0, 0.05, (A, H, W))
    caf[:, 2] = ii[None] + rng.normal(0.0, 1.0, (A, H, W))
    caf[:, 3] = jj[None] + rng.normal(0.0, 1.0, (A, H, W))
    caf[:, 4] = ii[None] + rng.normal(0.0, 1.0, (A, H, W))
    caf[:, 5] = jj[None] + rng.normal(0.0, 1.0, (A, H, W))
    caf[:, 6] = rng.uniform(0.1, 1.0, (A, H, W))
    caf[:, 7] = rng.uniform(0.1, 1.0, (A, H, W))

    # normalise the pose template: origin at its centre, y pointing down
    p = pose.copy()
    p[:, 1] = -p[:, 1]
    p -= 0.5 * (p.min(axis=0) + p.max(axis=0))
    extent = (p.max(axis=0) - p.min(axis=0)).max()

    for _ in range(n_people):
        # person height between ~25% and ~75% of the field
        size = rng.uniform(*size_range) * min(H, W)
        unit = size / extent                      # field units per pose unit
        half = 0.5 * unit * (p.max(axis=0) - p.min(axis=0))
        cx = rng.uniform(half[0] + 1.0, max(half[0] + 1.5, W - 2.0 - half[0]))
        cy = rng.uniform(half[1] + 1.0, max(half[1] + 1.5, H - 2.0 - half[1]))
        joints = p * unit + np.array([cx, cy])
        joints += rng.normal(0.0, 0.02 * unit, joints.shape)   # per-person shape jitter
        s = max(0.5, 0.3 * unit)                  # joint scale in field units

        for k in range(K):
            x, y = joints[k]
            bj, bi = _blob(cif[k, 1], None, x, y, max(2.3, 1.5 + 0.5 * s), 0.9, max(1.3, 0.6 + 0.5 * s), rng)
            n = len(bj)
            cif[k, 2, bj, bi] = x + rng.normal(0.0, cif_noise, n)
            cif[k, 3, bj, bi] = y + rng.normal(0.0, cif_noise, n)
            cif[k, 4, bj, bi] = s * (1.0 + rng.normal(0.0, 0.03, n))

        for a, (j1, j2) in enumerate(skeleton):
            x1, y1 = joints[j1 - 1]
            x2, y2 = joints[j2 - 1]
            for t in (0.0, 0.5, 1.0):
                bx, by = x1 + t * (x2 - x1), y1 + t * (y2 - y1)
                bj, bi = _blob(caf[a, 1], None, bx, by, max(2.0, 1.5 + 0.3 * s), 0.85, max(1.2, 0.8 + 0.4 * s), rng)
                n = len(bj)
                caf[a, 2, bj, bi] = x1 + rng.normal(0.0, noise, n)
                caf[a, 3, bj, bi] = y1 + rng.normal(0.0, noise, n)
                caf[a, 4, bj, bi] = x2 + rng.normal(0.0, noise, n)
                caf[a, 5, bj, bi] = y2 + rng.normal(0.0, noise, n)
                caf[a, 6, bj, bi] = s * (1.0 + rng.normal(0.0, 0.03, n))
                caf[a, 7, bj, bi] = s * (1.0 + rng.normal(0.0, 0.03, n))

    return cif.astype(np.float32), caf.astype(np.float32)


def tracking_skeleton(skeleton=None, n_keypoints=17):
    """1-based bones of the reference's tracking pose (``decoder/tracking_pose.py:50-57``): the single-frame
    skeleton plus one temporal bone per joint, (k, k + n_keypoints) -- current frame to previous frame."""
    skeleton = list(skeleton if skeleton is not None else constants.COCO_PERSON_SKELETON)
    return skeleton + [(k + 1, k + 1 + n_keypoints) for k in range(n_keypoints)]


def synth_tracking_fields(seed, n_people, *, height=49, width=49, motion=(0.35, -0.2), size_range=(0.3, 0.7)):
    """Fields of a two-frame tracking problem the way ``TrackingPose.__call__`` hands them to the decoder
    (``decoder/tracking_pose.py:209-217``): -> ``(cif [17,5,H,W], caf [19+17,8,H,W], full_cif [34,5,H,W])``.
    ``cif`` holds the CURRENT frame's joints only; ``caf`` = single-frame CAF head + temporal CAF head; joints
    17..33 (the previous frame, displaced by ``motion`` pose units) exist only through the temporal bones.
    ``full_cif`` (with fields for the previous frame too) lets a test derive previous-frame poses."""
    pose = np.asarray(constants.COCO_UPRIGHT_POSE, dtype=np.float64)[:, :2]
    K = len(pose)
    double_pose = np.concatenate([pose, pose + np.asarray(motion, dtype=np.float64)[None]], axis=0)
    full_cif, caf = synth_fields(seed, n_people, height=height, width=width, pose=double_pose,
                                 skeleton=tracking_skeleton(n_keypoints=K), size_range=size_range)
    return full_cif[:K].copy(), caf, full_cif


def synth_tracking_sequence(seed, n_people, n_frames, *, height=49, width=65, appear=None):
    """A short synthetic video for the tracking decoder: people with the COCO upright pose walk with constant
    velocity; per frame -> ``(cif [17,5,H,W], caf [19,8,H,W], tcaf [17,8,H,W])``, the three heads a tracking
    network emits (reference ``headmeta.py:136-186``): ``tcaf`` field k points from joint k in this frame to
    where that joint was in the previous frame.  ``appear[p]`` = first frame person p is visible in."""
    rng = np.random.default_rng(seed)
    pose = np.asarray(constants.COCO_UPRIGHT_POSE, dtype=np.float64)[:, :2].copy()
    pose[:, 1] = -pose[:, 1]
    pose -= 0.5 * (pose.min(axis=0) + pose.max(axis=0))
    extent = (pose.max(axis=0) - pose.min(axis=0)).max()
    K, skeleton = len(pose), constants.COCO_PERSON_SKELETON
    H, W = height, width
    appear = list(appear) if appear is not None else [0] * n_people
    people = []
    for _ in range(n_people):
        unit = rng.uniform(0.35, 0.6) * min(H, W) / extent
        half = 0.5 * unit * (pose.max(axis=0) - pose.min(axis=0))
        start = np.array([rng.uniform(half[0] + 3.0, W - 4.0 - half[0]), rng.uniform(half[1] + 2.0, H - 3.0 - half[1])])
        velocity = rng.uniform(-0.6, 0.6, 2)
        people.append((unit, start, velocity))
    jj, ii = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')

    def background(n, comps):
        """Structureless field: low confidences, regressions pointing at the own cell, random scales."""
        n_vec = 1 if comps == 5 else 2                   # CIF: (x, y); CAF: (x1, y1, x2, y2)
        f = np.empty((n, comps, H, W), dtype=np.float64)
        f[:, 0] = rng.normal(0.0, 1.0, (n, H, W))
        f[:, 1] = rng.uniform(0.0, 0.05, (n, H, W))
        for v in range(n_vec):
            f[:, 2 + 2 * v] = ii[None] + rng.normal(0.0, 1.0, (n, H, W))
            f[:, 3 + 2 * v] = jj[None] + rng.normal(0.0, 1.0, (n, H, W))
        f[:, 2 + 2 * n_vec:] = rng.uniform(0.1, 1.0, (n, comps - 2 - 2 * n_vec, H, W))
        return f

    def bone_blobs(field, a, p1, p2, s):
        for t in (0.0, 0.5, 1.0):
            bx, by = p1[0] + t * (p2[0] - p1[0]), p1[1] + t * (p2[1] - p1[1])
            bj, bi = _blob(field[a, 1], None, bx, by, max(2.0, 1.5 + 0.3 * s), 0.85, max(1.2, 0.8 + 0.4 * s), rng)
            n = len(bj)
            field[a, 2, bj, bi] = p1[0] + rng.normal(0.0, 0.05, n)
            field[a, 3, bj, bi] = p1[1] + rng.normal(0.0, 0.05, n)
            field[a, 4, bj, bi] = p2[0] + rng.normal(0.0, 0.05, n)
            field[a, 5, bj, bi] = p2[1] + rng.normal(0.0, 0.05, n)
            field[a, 6, bj, bi] = s * (1.0 + rng.normal(0.0, 0.03, n))
            field[a, 7, bj, bi] = s * (1.0 + rng.normal(0.0, 0.03, n))

    frames = []
    for t in range(n_frames):
        cif, caf, tcaf = background(K, 5), background(len(skeleton), 8), background(K, 8)
        for p, (unit, start, velocity) in enumerate(people):
            if t < appear[p]:
                continue
            now = pose * unit + start + t * velocity
            before = pose * unit + start + (t - 1) * velocity
            s = max(0.5, 0.3 * unit)
            for k in range(K):
                x, y = now[k]
                bj, bi = _blob(cif[k, 1], None, x, y, max(2.3, 1.5 + 0.5 * s), 0.9, max(1.3, 0.6 + 0.5 * s), rng)
                n = len(bj)
                cif[k, 2, bj, bi] = x + rng.normal(0.0, 0.05, n)
                cif[k, 3, bj, bi] = y + rng.normal(0.0, 0.05, n)
                cif[k, 4, bj, bi] = s * (1.0 + rng.normal(0.0, 0.03, n))
                if t > appear[p]:
                    bone_blobs(tcaf, k, now[k], before[k], s)
            for a, (j1, j2) in enumerate(skeleton):
                bone_blobs(caf, a, now[j1 - 1], now[j2 - 1], s)
        frames.append((cif.astype(np.float32), caf.astype(np.float32), tcaf.astype(np.float32)))
    return frames


PEOPLE_CYCLE = (1, 5, 10, 20, 3, 8, 15, 2)


def synth_batch(batch, *, seed0=0, height=81, width=81, people=None,
                pose=None, skeleton=None, size_range=(0.25, 0.75)):
    """A batch of images: image ``b`` uses seed ``seed0 + b`` and
    ``people[b % len(people)]`` synthetic persons."""
    if people is None:
        people = PEOPLE_CYCLE
    if isinstance(people, int):
        people = (people,)
    cifs, cafs = [], []
    for b in range(batch):
        cif, caf = synth_fields(seed0 + b, people[b % len(people)],
                                height=height, width=width, pose=pose, skeleton=skeleton,
                                size_range=size_range)
        cifs.append(cif)
        cafs.append(caf)
    return np.stack(cifs), np.stack(cafs)


def adversarial_fields(seed, *, n_keypoints=17, n_bones=19, height=81, width=81, people=0):
    """The structureless all-active case a random-initialised head produces
    (sigmoid ~ 0.5 everywhere): every cell passes every threshold.  ``people`` > 0 plants that many
    :func:`synth_fields` people into the noise (their blobs replace the background cells they cover), so that
    the decode of an all-active field has poses to return (COCO-17 fields only)."""
    if people:
        cif, caf = adversarial_fields(seed, n_keypoints=n_keypoints, n_bones=n_bones, height=height, width=width)
        pcif, pcaf = synth_fields(seed + 7919, people, height=height, width=width)
        return (np.where(pcif[:, 1:2] > 0.3, pcif, cif).astype(np.float32),
                np.where(pcaf[:, 1:2] > 0.3, pcaf, caf).astype(np.float32))
    rng = np.random.default_rng(seed)
    H, W = height, width
    jj, ii = np.meshgrid(np.arange(H, dtype=np.float64),
                         np.arange(W, dtype=np.float64), indexing='ij')
    cif = np.empty((n_keypoints, 5, H, W))
    cif[:, 0] = rng.normal(0, 1, (n_keypoints, H, W))
    cif[:, 1] = 1.0 / (1.0 + np.exp(-rng.normal(0, 1, (n_keypoints, H, W))))
    cif[:, 2] = ii[None] + rng.normal(0, 1, (n_keypoints, H, W))
    cif[:, 3] = jj[None] + rng.normal(0, 1, (n_keypoints, H, W))
    cif[:, 4] = np.log1p(np.exp(rng.normal(0, 1, (n_keypoints, H, W))))
    caf = np.empty((n_bones, 8, H, W))
    caf[:, 0] = rng.normal(0, 1, (n_bones, H, W))
    caf[:, 1] = 1.0 / (1.0 + np.exp(-rng.normal(0, 1, (n_bones, H, W))))
    for c in (2, 4):
        caf[:, c] = ii[None] + rng.normal(0, 1, (n_bones, H, W))
        caf[:, c + 1] = jj[None] + rng.normal(0, 1, (n_bones, H, W))
    caf[:, 6] = np.log1p(np.exp(rng.normal(0, 1, (n_bones, H, W))))
    caf[:, 7] = np.log1p(np.exp(rng.normal(0, 1, (n_bones, H, W))))
    return cif.astype(np.float32), caf.astype(np.float32)


def synth_det_field(seed, n_objects, *, n_categories=8, height=81, width=81):
    """One image's CifDet field ``[n_categories, 6, H, W]`` (0 unused, 1 conf, 2 x, 3 y, 4 w, 5 h):
    a confidence blob at every object centre, regressions pointing at the centre, box size in
    field units (reference ``headmeta.py:116-133``)."""
    rng = np.random.default_rng(seed)
    H, W = height, width
    jj, ii = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
    f = np.empty((n_categories, 6, H, W), dtype=np.float64)
    f[:, 0] = rng.normal(0.0, 1.0, (n_categories, H, W))
    f[:, 1] = rng.uniform(0.0, 0.05, (n_categories, H, W))
    f[:, 2] = ii[None] + rng.normal(0.0, 1.0, (n_categories, H, W))
    f[:, 3] = jj[None] + rng.normal(0.0, 1.0, (n_categories, H, W))
    f[:, 4] = rng.uniform(0.5, 3.0, (n_categories, H, W))
    f[:, 5] = rng.uniform(0.5, 3.0, (n_categories, H, W))
    for _ in range(n_objects):
        c = int(rng.integers(n_categories))
        w, h = rng.uniform(4.0, 0.5 * W), rng.uniform(4.0, 0.5 * H)
        cx, cy = rng.uniform(0.5 * w, W - 1 - 0.5 * w), rng.uniform(0.5 * h, H - 1 - 0.5 * h)
        bj, bi = _blob(f[c, 1], None, cx, cy, 2.5 + 0.05 * min(w, h), 0.9, 1.3 + 0.05 * min(w, h), rng)
        n = len(bj)
        f[c, 2, bj, bi] = cx + rng.normal(0.0, 0.05, n)
        f[c, 3, bj, bi] = cy + rng.normal(0.0, 0.05, n)
        f[c, 4, bj, bi] = w * (1.0 + rng.normal(0.0, 0.03, n))
        f[c, 5, bj, bi] = h * (1.0 + rng.normal(0.0, 0.03, n))
    return f.astype(np.float32)
