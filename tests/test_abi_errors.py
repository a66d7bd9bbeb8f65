"""Malformed tapes handed to the tape-level ABI are error codes (validated on the host before anything reaches a
device), never device faults.  Runs without a GPU: validation precedes device selection."""
import ctypes as C

import numpy as np
import pytest

import maray_amd as M
from maray_amd.api import Program

OP = dict(NEG=2, STEP=6, ADD=10, MUL=11, MIN=13, APP=14, OUT=16, SKIPZ=18)
NONE = 0xFFF


def ins(op, aux=0, dst=NONE, a=0, b=0):
    return (op & 0x7F) | ((aux & 0x1FFF) << 7) | ((dst & 0xFFF) << 20) | ((a & 0xFFFF) << 32) | ((b & 0xFFFF) << 48)


def ref(kind, idx):
    return (kind << 14) | idx


SLOT, CONST, YVAL, SPEC = 0, 1, 2, 3
X, Y, ACC = ref(SPEC, 0), ref(SPEC, 1), ref(SPEC, 2)


def create(pix_ops, n_slots=4, consts=(1.0,), n_yvals=0, n_app=0, row_ops=(), version=2):
    c = np.array(consts, np.float64)
    po = np.array(pix_ops, np.uint64)
    ro = np.array(row_ops, np.uint64)
    p = Program()
    p.version = version
    p.n_consts = len(c); p.consts = c.ctypes.data_as(C.POINTER(C.c_double))
    p.n_row_ops = len(ro); p.row_ops = ro.ctypes.data_as(C.POINTER(C.c_uint64)) if len(ro) else None
    p.n_row_slots = 4; p.n_yvals = n_yvals
    p.n_pix_ops = len(po); p.pix_ops = po.ctypes.data_as(C.POINTER(C.c_uint64))
    p.n_pix_slots = n_slots; p.n_app = n_app
    h = C.c_void_p()
    rc = M.lib().maray_hip_ctx_create(0, C.byref(p), None, 0, None, C.byref(h))
    if rc == 0:
        M.lib().maray_hip_ctx_free(h)
    return rc, M.lib().maray_last_error().decode()


GOOD = [ins(OP['ADD'], dst=0, a=X, b=ref(CONST, 0)), ins(OP['OUT'], 0, a=ACC), ins(OP['OUT'], 1, a=ref(SLOT, 0)), ins(OP['OUT'], 2, a=Y)]


def test_well_formed_tape_passes_validation():
    rc, msg = create(GOOD)
    assert rc in (0, -8), msg          # -8 only because this host has no gfx950 device


@pytest.mark.parametrize('name,ops,kw', [
    ('bad version', GOOD, dict(version=7)),
    ('invalid opcode', [ins(99, a=X)] + GOOD, {}),
    ('slot out of range', [ins(OP['NEG'], dst=9, a=X)] + GOOD, {}),
    ('slot read before write', [ins(OP['NEG'], dst=1, a=ref(SLOT, 2))] + GOOD, {}),
    ('ACC before any op', [ins(OP['OUT'], 0, a=ACC)], {}),
    ('constant out of range', [ins(OP['ADD'], dst=0, a=X, b=ref(CONST, 5))] + GOOD, {}),
    ('y value out of range', [ins(OP['ADD'], dst=0, a=X, b=ref(YVAL, 0))] + GOOD, {}),
    ('output index out of range', GOOD + [ins(OP['OUT'], 3, a=X)], {}),
    ('App id above n_app', [ins(OP['APP'], 7, dst=0, a=X, b=Y)] + GOOD, {}),
    ('skip past the end', [ins(OP['STEP'], dst=1, a=X), ins(OP['SKIPZ'], 50, dst=2, a=ACC)] + GOOD, {}),
    ('skip region not ending in an AND/OR', [ins(OP['STEP'], dst=1, a=X), ins(OP['SKIPZ'], 1, dst=2, a=ACC), ins(OP['NEG'], dst=2, a=Y)] + GOOD, {}),
    ('value defined only inside a skipped region read after it',
     [ins(OP['STEP'], dst=1, a=X), ins(OP['SKIPZ'], 2, dst=2, a=ACC), ins(OP['STEP'], dst=3, a=Y), ins(OP['MIN'], dst=2, a=ref(SLOT, 1), b=ACC),
      ins(OP['OUT'], 0, a=ref(SLOT, 3))], {}),
    ('span special (XMIN) in the PIXEL section', [ins(OP['ADD'], dst=0, a=X, b=ref(SPEC, 4))] + GOOD, {}),
    ('span special (YMAX) in the PIXEL section', [ins(OP['ADD'], dst=0, a=X, b=ref(SPEC, 5))] + GOOD, {}),
    ('special operand out of range', [ins(OP['ADD'], dst=0, a=X, b=ref(SPEC, 7))] + GOOD, {}),
    ('X in the ROW section', GOOD, dict(n_yvals=1, row_ops=[ins(OP['ADD'], dst=0, a=X, b=Y), ins(OP['OUT'], 0, a=ACC)])),
    ('OUT inside a skip region',
     [ins(OP['STEP'], dst=1, a=X), ins(OP['SKIPZ'], 2, dst=2, a=ACC), ins(OP['OUT'], 0, a=Y), ins(OP['MIN'], dst=2, a=ref(SLOT, 1), b=ref(SLOT, 1))], {}),
])
def test_malformed_tapes_are_rejected(name, ops, kw):
    rc, msg = create(ops, **kw)
    assert rc in (-1, -7), (name, rc, msg)
    assert msg


def test_span_specials_are_accepted_in_the_row_section():
    """XMIN / XMAX / YMIN / YMAX: the ends of the span of pixels a guard bounds its boolean over (tape v2)."""
    row = [ins(OP['ADD'], dst=0, a=ref(SPEC, 3), b=ref(SPEC, 4)), ins(OP['ADD'], dst=1, a=ref(SPEC, 5), b=ref(SPEC, 6)),
           ins(OP['MIN'], dst=2, a=ref(SLOT, 0), b=ref(SLOT, 1)), ins(OP['OUT'], 0, a=ACC)]
    pix = [ins(OP['ADD'], dst=0, a=X, b=ref(YVAL, 0))] + GOOD[1:]
    rc, msg = create(pix, n_yvals=1, row_ops=row)
    assert rc in (0, -8), msg


def test_null_arguments_are_errors_not_crashes():
    L = M.lib()
    assert L.maray_scene_from_bytes(None, 0, None) == -1
    assert L.maray_lower(None, None, None) == -1
    assert L.maray_hip_ctx_create(0, None, None, 0, None, None) == -1
    assert L.maray_hip_render_rows(None, 4, 4, 0, 4, None, None) == -1
    assert L.maray_png_write(None, None, 0, 0) == -1
    L.maray_scene_free(None); L.maray_tape_free(None); L.maray_hip_ctx_free(None); L.maray_free(None)
