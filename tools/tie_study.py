#!/usr/bin/env python
"""How often does the ORDER OF EQUAL-SCORE SEEDS change a decode?  (CPU only.)

The reference sorts seeds with an unstable ``std::sort`` (cif_seeds.cpp:93-99): the order of seeds with
exactly equal scores is unspecified by the language and is whatever libstdc++'s introsort leaves.  The HIP
path sorts a total order (score descending, then cell index).  On float32 synthetic fields exact ties do not
occur; a bf16 network's confidences are 8-bit-mantissa values and tie constantly.  This tool quantises the
field tensors to bfloat16 (all components, like a bf16 head's output cast back to float32), decodes with

  reference   the REAL reference decoder (oracle/_ref) -- libstdc++ std::sort
  rule 1      the restatement with ties ordered by cell index ascending  (= the HIP path's order)
  rule 2      the restatement with ties ordered by cell index descending

and reports, per rule: images with any discrete mismatch (annotation count or joint presence), images whose
largest |delta| exceeds 1e-4, and the largest |delta| among the images that agree discretely.

    python tools/tie_study.py [--images 200] [--size 81]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from openpifpaf_amd import constants, synth  # noqa: E402
from oracle import port, reference  # noqa: E402


def to_bf16(a):
    """float32 -> nearest-even bfloat16 -> float32 (numpy only)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def set_tie_rule(rule):
    port.set_seed_tie_rule(rule)


def classify(got, want, tol=1e-4):
    """-> ('equal' | 'within' | 'beyond' | 'discrete', max |delta| or None)"""
    if got.shape != want.shape:
        return 'discrete', None
    if got.size == 0:
        return 'equal', 0.0
    if not np.array_equal(got[..., 0] > 0, want[..., 0] > 0):
        return 'discrete', None
    d = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    return ('equal' if d == 0.0 else 'within' if d <= tol else 'beyond'), d


def n_tied_seeds(cif, stride=8):
    hr = port.cifhr_accumulate(cif, stride)
    _, v = port.cifseeds(cif, stride, hr)
    s = v[:, 0]
    return int(len(s) - len(np.unique(s))), len(s)


def study(n_images, size, quantise=True, people=synth.PEOPLE_CYCLE, log=print):
    skel0 = np.asarray(constants.COCO_PERSON_SKELETON, dtype=np.int64) - 1
    reference.reset_statics()
    stats = {1: dict(equal=0, within=0, beyond=0, discrete=0, max=0.0),
             2: dict(equal=0, within=0, beyond=0, discrete=0, max=0.0)}
    restatement_vs_ref = dict(equal=0, within=0, beyond=0, discrete=0)
    tied = total = 0
    for i in range(n_images):
        cif, caf = synth.synth_fields(50_000 + i, people[i % len(people)], height=size, width=size)
        if quantise:
            cif, caf = to_bf16(cif), to_bf16(caf)
        want, _, _ = reference.decode(cif, 8, caf, 8, skel0)
        t, n = n_tied_seeds(cif)
        tied += t
        total += n
        set_tie_rule(0)
        got0, _ = port.decode(cif, 8, caf, 8, skel0)
        restatement_vs_ref[classify(got0, want)[0]] += 1
        for rule in (1, 2):
            set_tie_rule(rule)
            got, _ = port.decode(cif, 8, caf, 8, skel0)
            kind, d = classify(got, want)
            stats[rule][kind] += 1
            if d is not None:
                stats[rule]['max'] = max(stats[rule]['max'], d)
        set_tie_rule(0)
    log('%d images %dx%d, fields %s; %d of %d seeds share their score with another seed (%.1f%%)'
        % (n_images, size, size, 'quantised to bf16' if quantise else 'float32', tied, total,
           100.0 * tied / max(1, total)))
    log('restatement with std::sort vs reference: %s' % restatement_vs_ref)
    for rule, name in ((1, 'cell index ascending (HIP path)'), (2, 'cell index descending')):
        s = stats[rule]
        log('ties by %-32s: bit-equal %d, <=1e-4 %d, >1e-4 %d, discrete mismatch %d (max |delta| among '
            'discretely equal images %.3g)' % (name, s['equal'], s['within'], s['beyond'], s['discrete'], s['max']))
    return stats, restatement_vs_ref, (tied, total)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=200)
    ap.add_argument('--size', type=int, default=81)
    ap.add_argument('--float32', action='store_true', help='control: unquantised fields (no ties expected)')
    a = ap.parse_args()
    study(a.images, a.size, quantise=not a.float32)
