"""Reference evaluator of the tape format (include/maray_tape.h) in numpy.

Test infrastructure for the host logic: it lets the CPU suite check that the
lowering (Expr -> tape) preserves the oracle's values without a GPU.  It is not
part of the product and is never used on a render path.  sin/exp/ln go through
Python's math module (= the platform libm the oracle calls).
"""
import math

import numpy as np

OP = dict(NOP=0, MOV=1, NEG=2, ABS=3, RECIP=4, SQRT=5, STEP=6, SIN=7, EXP=8, LN=9, ADD=10, MUL=11, MAX=12, MIN=13,
          APP=14, TEXDIM=15, OUT=16, STEPSIN=17, SKIPZ=18, SKIPNZ=19)
K_SLOT, K_CONST, K_YVAL, K_SPEC = 0, 1, 2, 3
DST_NONE = 0xFFF

_vsin = np.vectorize(math.sin, otypes=[np.float64])
_vexp = np.vectorize(lambda v: math.exp(v) if v < 709.782712893384 else (math.inf if v == v else v), otypes=[np.float64])


def _vlog(a):
    out = np.empty_like(a)
    for i, v in np.ndenumerate(a):
        if v != v: out[i] = v
        elif v > 0: out[i] = math.log(v)
        elif v == 0: out[i] = -math.inf
        else: out[i] = math.nan
    return out


def _sin(a):
    with np.errstate(all='ignore'):
        fin = np.isfinite(a)
        out = np.full_like(a, np.nan)
        out[fin] = _vsin(a[fin])
    return out


def _max(a, b):
    """f64::max as the oracle defines it: NaN-ignoring, and +0 > -0 (IEEE 754-2019 maximumNumber)."""
    r = np.where(a > b, a, b)
    r = np.where(a == b, np.where(np.signbit(a), b, a), r)
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, r))


def _min(a, b):
    r = np.where(a < b, a, b)
    r = np.where(a == b, np.where(np.signbit(a), a, b), r)
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, r))


def decode(ins):
    ins = int(ins)
    return (ins & 0x7F, (ins >> 7) & 0x1FFF, (ins >> 20) & 0xFFF, (ins >> 32) & 0xFFFF, (ins >> 48) & 0xFFFF)


def _cast_u32(v):
    v = np.where(v > 0, v, 0.0)          # NaN / negatives -> 0
    return np.minimum(v, 4294967295.0).astype(np.uint64)


def run_section(ops, consts, n_slots, X, Y, yvals, textures, n_out, honor_skips=False, w=None, span=None, yspan=None):
    """Evaluate one section for a vector of items.  X, Y: float64 arrays (same
    shape); yvals: array [..., n_yvals] broadcastable per item or None.
    honor_skips: take SKIPZ / SKIPNZ when the whole vector agrees (call per 64-item "wavefront")."""
    shape = np.shape(Y)
    slots = [None] * max(n_slots, 1)
    outs = [None] * n_out
    acc = None

    def fetch(ref):
        kind, idx = ref >> 14, ref & 0x3FFF
        if kind == K_SLOT: return slots[idx]
        if kind == K_CONST: return np.full(shape, consts[idx])
        if kind == K_YVAL: return yvals[..., idx]
        if idx == 3: return np.full(shape, float(span[1] if span else w - 1))      # XMAX
        if idx == 4: return np.full(shape, float(span[0] if span else 0))          # XMIN
        if idx == 5: return Y if yspan is None else np.broadcast_to(yspan[1], shape)     # YMAX (per row: = Y)
        if idx == 6: return Y if yspan is None else np.broadcast_to(yspan[0], shape)     # YMIN
        return X if idx == 0 else (Y if idx == 1 else acc)

    with np.errstate(all='ignore'):
        pc = -1
        n_ops = len(ops)
        while pc + 1 < n_ops:
            pc += 1
            op, aux, dst, ra, rb = decode(ops[pc])
            if op == OP['NOP']: continue
            if op in (OP['SKIPZ'], OP['SKIPNZ']):
                if not honor_skips: continue                                   # an evaluator may ignore the skips
                gv = fetch(ra)
                want = 0.0 if op == OP['SKIPZ'] else 1.0
                if np.all(gv == want):
                    acc = np.full(shape, want)
                    if dst != DST_NONE: slots[dst] = acc
                    pc += aux
                continue
            if op == OP['OUT']:
                outs[aux] = fetch(ra); continue
            if op == OP['MOV']: r = fetch(ra)
            elif op == OP['NEG']: r = -fetch(ra)
            elif op == OP['ABS']: r = np.abs(fetch(ra))
            elif op == OP['RECIP']: r = 1.0 / fetch(ra)
            elif op == OP['SQRT']: r = np.sqrt(fetch(ra))
            elif op == OP['STEP']: r = np.where(fetch(ra) >= 0.0, 1.0, 0.0)
            elif op == OP['SIN']: r = _sin(fetch(ra))
            elif op == OP['STEPSIN']: r = np.where(_sin(fetch(ra)) >= 0.0, 1.0, 0.0)
            elif op == OP['EXP']: r = _vexp(fetch(ra))
            elif op == OP['LN']: r = _vlog(fetch(ra))
            elif op == OP['ADD']: r = fetch(ra) + fetch(rb)
            elif op == OP['MUL']: r = fetch(ra) * fetch(rb)
            elif op == OP['MAX']: r = _max(fetch(ra), fetch(rb))
            elif op == OP['MIN']: r = _min(fetch(ra), fetch(rb))
            elif op == OP['TEXDIM']:
                t = textures[aux // 5]
                r = np.full(shape, float(t.shape[1] if aux % 5 == 3 else t.shape[0]))
            elif op == OP['APP']:
                t = textures[aux // 5]
                a, b = fetch(ra), fetch(rb)
                h, w = t.shape[0], t.shape[1]
                xi, yi = _cast_u32(a), _cast_u32(b)
                inb = ~((a < 0) | (b < 0)) & (xi < w) & (yi < h)
                r = np.zeros(shape)
                r[inb] = t[yi[inb].astype(np.int64), xi[inb].astype(np.int64), aux % 5].astype(np.float64)
            else:
                raise ValueError('bad opcode %d' % op)
            acc = r
            if dst != DST_NONE:
                slots[dst] = r
    return outs


def render_rows_waves(tape, w, y0, y1, textures=None, tile=None, yrows=None):
    """Like render_rows, but wavefront by wavefront (64 consecutive x of one row; the ROW section in
    groups of 64 rows) with SKIPZ / SKIPNZ honoured the way the device kernels do.  tile: evaluate
    the ROW section once per `tile` pixels of a row with XMIN / XMAX = that span (what the
    specialised kernels do for the guard values) instead of once per row.  yrows: additionally
    with YMIN / YMAX = the ends of the group of `yrows` rows a row belongs to (legal only for a
    tape none of whose guards reads Y; the caller checks)."""
    consts, row_ops, pix_ops = tape.arrays()
    info = tape.info
    rows = y1 - y0
    out = np.zeros((rows, w, 3))
    spans = [None] if not tile else [(x0, min(w, x0 + tile) - 1) for x0 in range(0, w, tile)]
    yv_span = []
    for span in spans:
        yv_all = None
        if info['n_yvals']:
            yv_all = np.zeros((rows, info['n_yvals']))
            for r0 in range(0, rows, 64):
                ys = np.arange(y0 + r0, min(y1, y0 + r0 + 64), dtype=np.float64)
                ysp = None
                if yrows:
                    lo = y0 + ((ys - y0) // yrows) * yrows
                    ysp = (lo, np.minimum(lo + yrows - 1, y1 - 1))
                outs = run_section(row_ops, consts, info['n_row_slots'], None, ys, None, textures, info['n_yvals'], True, w=w, span=span,
                                   yspan=ysp)
                yv_all[r0:r0 + len(ys)] = np.stack(outs, axis=-1)
        yv_span.append(yv_all)
    for r in range(rows):
        for x0 in range(0, w, 64):
            yv_all = yv_span[x0 // tile if tile else 0]
            X = np.arange(x0, x0 + 64, dtype=np.float64)        # lanes beyond w compute too, like on the device
            Y = np.full(64, float(y0 + r))
            yv = np.broadcast_to(yv_all[r][None, :], (64, info['n_yvals'])) if yv_all is not None else None
            o = run_section(pix_ops, consts, info['n_pix_slots'], X, Y, yv, textures, 3, True)
            n = min(64, w - x0)
            out[r, x0:x0 + n] = np.stack(o, axis=-1)[:n]
    return out


def render_rows(tape, w, y0, y1, textures=None):
    """Evaluate a maray_amd.Tape over rows [y0,y1) x [0,w) -> (rows, w, 3) float64."""
    consts, row_ops, pix_ops = tape.arrays()
    info = tape.info
    rows = y1 - y0
    ys = np.arange(y0, y1, dtype=np.float64)
    yv = None
    if info['n_yvals']:
        outs = run_section(row_ops, consts, info['n_row_slots'], None, ys, None, textures, info['n_yvals'], w=w)
        yv = np.stack(outs, axis=-1)                       # (rows, n_yvals)
        yv = np.broadcast_to(yv[:, None, :], (rows, w, info['n_yvals']))
    X = np.broadcast_to(np.arange(w, dtype=np.float64)[None, :], (rows, w)).copy()
    Y = np.broadcast_to(ys[:, None], (rows, w)).copy()
    o = run_section(pix_ops, consts, info['n_pix_slots'], X, Y, yv, textures, 3)
    return np.stack(o, axis=-1)


def cast_u8(v):
    """Rust `as u8`."""
    v = np.where(v > 0, v, 0.0)
    return np.minimum(v, 255.0).astype(np.uint8)


def guards_reading_y(tape):
    """How many guards (y values that only gate SKIP ops of the PIXEL section) have SPEC Y in their cone: those are
    exact for one row and must be evaluated per row; a tape with none may have its guards evaluated for groups of rows
    (YMIN / YMAX).  Returns (n_guards, n_reading_y)."""
    consts, row_ops, pix_ops = tape.arrays()
    used_as_operand, used_as_guard = set(), set()
    for ins in pix_ops:
        op, aux, dst, ra, rb = decode(ins)
        if op == OP['NOP']:
            continue
        if ra >> 14 == K_YVAL:
            (used_as_guard if op in (OP['SKIPZ'], OP['SKIPNZ']) else used_as_operand).add(ra & 0x3FFF)
        if OP['ADD'] <= op <= OP['APP'] and rb >> 14 == K_YVAL:
            used_as_operand.add(rb & 0x3FFF)
    guards = used_as_guard - used_as_operand
    reads_y, slot_writer, acc = {}, {}, None
    n_reading = 0

    def src(ref):
        kind, idx = ref >> 14, ref & 0x3FFF
        if kind == K_SLOT: return reads_y.get(slot_writer.get(idx), False)
        if kind == K_SPEC: return idx == 1 or (idx == 2 and reads_y.get(acc, False))
        return False
    for j, ins in enumerate(row_ops):
        op, aux, dst, ra, rb = decode(ins)
        if op in (OP['NOP'], OP['SKIPZ'], OP['SKIPNZ']):
            continue
        r = src(ra) if op != OP['TEXDIM'] else False
        if op == OP['OUT']:
            if aux in guards and r:
                n_reading += 1
            continue
        if OP['ADD'] <= op <= OP['APP']:
            r = r or src(rb)
        reads_y[j] = r
        acc = j
        if dst != DST_NONE:
            slot_writer[dst] = j
    return len(guards), n_reading
