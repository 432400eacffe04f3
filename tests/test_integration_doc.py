"""INTEGRATION.md section 3 shows the ctypes binding a maintainer would add inside the reference.  The
snippet is extracted from the document and checked against the header and the library, so that the
documentation cannot drift from ``include/openpifpaf_amd.h`` (its opa_shape lacked a field)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def snippet():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    code = [b for b in blocks if 'class HipCifCaf' in b]
    assert len(code) == 1, 'INTEGRATION.md section 3 no longer holds the binding snippet'
    from openpifpaf_amd import _lib
    return code[0].replace("ctypes.CDLL('libopenpifpaf_amd.so')", 'ctypes.CDLL(%r)' % _lib.LIB_PATH)


def header_struct_fields(name):
    text = open(os.path.join(ROOT, 'include', 'openpifpaf_amd.h')).read()
    body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (name, name), text, re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = decl.split(None, 1)
        fields += [(n.strip(), ctype) for n in names.split(',')]
    return fields


def test_documented_shape_struct_equals_the_header():
    ns = {}
    exec(compile(snippet(), 'INTEGRATION.md', 'exec'), ns)
    doc_fields = [(n, t) for n, t in ns['Shape']._fields_]
    hdr_fields = header_struct_fields('opa_shape')
    assert [n for n, _ in doc_fields] == [n for n, _ in hdr_fields]
    assert all(t is ctypes.c_int32 for _, t in doc_fields) and all(t == 'int32_t' for _, t in hdr_fields)
    assert ctypes.sizeof(ns['Shape']) == 4 * len(hdr_fields)
    from openpifpaf_amd import _lib
    assert ctypes.sizeof(ns['Shape']) == ctypes.sizeof(_lib.Shape)


@pytest.mark.gpu
def test_documented_binding_decodes_like_the_package(coco_skeleton0):
    import torch
    from openpifpaf_amd import native, synth
    ns = {}
    exec(compile(snippet(), 'INTEGRATION.md', 'exec'), ns)
    cifs, cafs = synth.synth_batch(3, seed0=321, height=41, width=49)
    cif, caf = torch.from_numpy(cifs).cuda(), torch.from_numpy(cafs).cuda()
    doc = ns['HipCifCaf'](17, torch.from_numpy(coco_skeleton0))
    out, ids, cnt = doc.call_batch(cif, 8, caf, 8, max_ann=64)
    torch.cuda.synchronize()
    want_out, want_ids, want_cnt = native.CifCaf(17, torch.from_numpy(coco_skeleton0), max_annotations=64).call_batch(cif, 8, caf, 8)
    assert torch.equal(cnt, want_cnt) and int(cnt.sum()) > 0
    for b in range(3):
        n = int(cnt[b])
        assert torch.equal(out[b, :n], want_out[b, :n]) and torch.equal(ids[b, :n], want_ids[b, :n])
