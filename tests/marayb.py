"""Scene-authoring helpers for the tests: Expr builders + bincode writer.

Test infrastructure.  Expressions are nested tuples ``(tag_name, ...)``; the
builder functions restate the reference's free functions so the parity tests
read like the reference's own tests (``src/lib.rs:1241-1285``):

    builders            src/lib.rs:836-1150  (x, y, nat, add, sub, mul, div, neg,
                        recip, sqrt, abs, step*, pos, range*, clamp*, set_*, sin,
                        cos, exp, ln, max, min, lerp, chess, p2_*, to_barycentric,
                        inside_triangle, to_uv, quad_*, let_, app)
    textures ids        src/textures.rs:14-23
    Grid2::cell         src/grid.rs:10-32
    Expr::subst2/scale  src/lib.rs:709-735, 804-806
    save()              src/lib.rs:1216-1224  (bincode 1.3.3 default options)

``encode`` writes the *current* tag numbering (``src/lib.rs:101-149``);
``encode(..., legacy=True)`` writes the older numbering ``data/chess.maray``
uses (no ``Arc`` variant, every tag one lower).
"""
import struct

TAGS = ['Arc', 'X', 'Y', 'Tau', 'E', 'Var', 'Nat', 'Neg', 'Abs', 'Recip', 'Sqrt', 'Step',
        'Sin', 'Exp', 'Ln', 'Add', 'Mul', 'Max', 'Min', 'Let', 'Decor', 'App']
TAG = {n: i for i, n in enumerate(TAGS)}
UNARY = ('Arc', 'Neg', 'Abs', 'Recip', 'Sqrt', 'Step', 'Sin', 'Exp', 'Ln')
BINARY = ('Add', 'Mul', 'Max', 'Min')


# ---- constructors (src/lib.rs:836-966) ------------------------------------
def app(id_, a, b): return ('App', id_, a, b)
def x(): return ('X',)
def y(): return ('Y',)
def var_id(i): return ('Var', i)
def tau(): return ('Tau',)
def pi(): return div(tau(), nat(2))
def rad_45(): return div(tau(), nat(8))
def rad_90(): return div(tau(), nat(4))
def e(): return ('E',)
def nat(a): return ('Nat', a)
def half(): return div(nat(1), nat(2))
def neg(a): return ('Neg', a)
def abs_(a): return ('Abs', a)
def recip(a): return ('Recip', a)
def sqrt(a): return ('Sqrt', a)
def step(a): return ('Step', a)
def step_at(a, x_): return step(sub(x_, a))
def step_pos(a): return set_inv(step(neg(a)))
def step_pos_at(a, x_): return step_pos(sub(x_, a))
def pos(cond, a, b): return lerp(b, a, step_pos(cond))
def range_(a, b, x_): return mul(step_at(a, x_), set_inv(step_at(b, x_)))
def range_incl(a, b, x_): return mul(step_at(a, x_), set_inv(step_pos_at(b, x_)))
def clamp(a, b, x_): return pos(sub(x_, a), pos(sub(x_, b), b, x_), a)
def clamp_unit(x_): return clamp(nat(0), nat(1), x_)
def clamp_u8(x_): return clamp(nat(0), nat(255), x_)
def ge(a, b): return step(sub(a, b))
def gt(a, b): return step_pos(sub(a, b))
def le(a, b): return set_inv(gt(a, b))
def lt(a, b): return set_inv(ge(a, b))
def eq(a, b): return set_and(ge(a, b), le(a, b))
def set_inv(a): return sub(nat(1), a)
def set_and(a, b): return min_(a, b)
def set_or(a, b): return max_(a, b)
def set_xor(a, b): return set_or(set_and(a, set_inv(b)), set_and(b, set_inv(a)))
def sin(a): return ('Sin', a)
def cos(a): return sin(add(a, rad_90()))
def exp(a): return ('Exp', a)
def ln(a): return ('Ln', a)
def max_(a, b): return ('Max', a, b)
def min_(a, b): return ('Min', a, b)
def add(a, b): return ('Add', a, b)
def sub(a, b): return add(a, neg(b))
def mul(a, b): return ('Mul', a, b)
def div(a, b): return mul(a, recip(b))
def square(a): return mul(a, a)
def lerp(a, b, t): return add(a, mul(sub(b, a), t))
def unit_to_rad(a): return mul(a, tau())
def rad_to_unit(a): return div(a, tau())
def let_(vars_, body): return ('Let', tuple(vars_), body)
def arc(a): return ('Arc', a)
def decor(a, tokens): return ('Decor', a, tuple(tokens))


def chess(n):   # src/lib.rs:969-973
    sx = step(sin(mul(mul(div(nat(n), nat(2)), tau()), x())))
    sy = step(sin(mul(mul(div(nat(n), nat(2)), tau()), y())))
    return set_xor(sx, sy)


def set_unit_square(f):   # src/lib.rs:975-980
    return set_and(set_and(range_(nat(0), nat(1), x()), range_(nat(0), nat(1), y())), f)


# ---- 2-D points (src/lib.rs:983-1075) -------------------------------------
def p2_neg(a): return [neg(a[0]), neg(a[1])]
def p2_add(a, b): return [add(a[0], b[0]), add(a[1], b[1])]
def p2_sub(a, b): return [sub(a[0], b[0]), sub(a[1], b[1])]
def p2_mul(a, b): return [mul(a[0], b[0]), mul(a[1], b[1])]
def p2_div(a, b): return [div(a[0], b[0]), div(a[1], b[1])]
def p2_scale(a, b): return p2_mul(a, [b, b])
def p2_dot(a, b): return add(mul(a[0], b[0]), mul(a[1], b[1]))
def p2_len(a): return sqrt(p2_dot(a, a))
def p2_lerp(a, b, t): return [lerp(a[0], b[0], t), lerp(a[1], b[1], t)]
def p2_pos(cond, a, b): return p2_lerp(b, a, step_pos(cond))                             # src/lib.rs:983-985


def quad_to_tri(quad, uv):   # src/lib.rs:1078-1085
    q0, q1, q2, q3 = quad
    uv0, uv1, uv2, uv3 = uv
    return [([q0, q1, q2], [uv0, uv1, uv2]), ([q1, q2, q3], [uv1, uv2, uv3])]


def quad_pos(quad, uv):   # src/lib.rs:1088-1096
    q0, q1, q2, q3 = quad
    return p2_lerp(p2_lerp(q0, q1, uv[0]), p2_lerp(q2, q3, uv[0]), uv[1])


def to_barycentric(tri, p):   # src/lib.rs:1115-1131
    x_, y_ = p
    (x1, y1), (x2, y2), (x3, y3) = tri
    den = add(mul(sub(y2, y3), sub(x1, x3)), mul(sub(x3, x2), sub(y1, y3)))
    l1 = div(add(mul(sub(y2, y3), sub(x_, x3)), mul(sub(x3, x2), sub(y_, y3))), den)
    l2 = div(add(mul(sub(y3, y1), sub(x_, x3)), mul(sub(x1, x3), sub(y_, y3))), den)
    l3 = sub(sub(nat(1), l1), l2)
    return [l1, l2, l3]


def inside_triangle(tri, p):   # src/lib.rs:1099-1102
    b0, b1, b2 = to_barycentric(tri, p)
    return set_and(set_and(step(b0), step(b1)), step(b2))


def to_uv(tri, uv, p):   # src/lib.rs:1105-1112
    b0, b1, b2 = to_barycentric(tri, p)
    return p2_add(p2_add(p2_scale(uv[0], b0), p2_scale(uv[1], b1)), p2_scale(uv[2], b2))


def grid_cell(grid, pos_, quad):   # Grid2::cell, src/grid.rs:10-32
    w, h = nat(grid[0]), nat(grid[1])
    fx, fy = div(nat(pos_[0]), w), div(nat(pos_[1]), h)
    gx, gy = div(nat(pos_[0] + 1), w), div(nat(pos_[1] + 1), h)
    uv0, uv1, uv2, uv3 = [fx, fy], [gx, fy], [fx, gy], [gx, gy]
    return ([quad_pos(quad, uv0), quad_pos(quad, uv1), quad_pos(quad, uv2), quad_pos(quad, uv3)],
            [uv0, uv1, uv2, uv3])


def subst2(ex, p):   # Expr::subst2, src/lib.rs:709-735 (does NOT descend into Let)
    t = ex[0]
    if t == 'Arc': return subst2(ex[1], p)
    if t == 'X': return p[0]
    if t == 'Y': return p[1]
    if t in ('Tau', 'E', 'Var', 'Nat', 'Let'): return ex
    if t in UNARY: return (t, subst2(ex[1], p))
    if t in BINARY: return (t, subst2(ex[1], p), subst2(ex[2], p))
    if t == 'Decor': return ('Decor', subst2(ex[1], p), ex[2])
    if t == 'App': return ('App', ex[1], subst2(ex[2], p), subst2(ex[3], p))
    raise ValueError(t)


def subst_xy_deep(ex, p):
    """Substitute X and Y everywhere, *including* inside Let definitions and
    bodies (unlike Expr::subst2).  Used to rescale a decoded scene
    (X -> X*(1/4), Y -> Y*(1/4))."""
    memo = {}

    def go(n):
        k = id(n)
        if k in memo: return memo[k]
        t = n[0]
        if t == 'X': r = p[0]
        elif t == 'Y': r = p[1]
        elif t in ('Tau', 'E', 'Var', 'Nat'): r = n
        elif t in UNARY: r = (t, go(n[1]))
        elif t in BINARY: r = (t, go(n[1]), go(n[2]))
        elif t == 'Let': r = ('Let', tuple((i, go(d)) for i, d in n[1]), go(n[2]))
        elif t == 'Decor': r = ('Decor', go(n[1]), n[2])
        elif t == 'App': r = ('App', n[1], go(n[2]), go(n[3]))
        else: raise ValueError(t)
        memo[k] = r
        return r
    return go(ex)


# ---- the rest of the reference's authoring functions (each restated from the file:line it cites) -------------------
def var(name):   # src/lib.rs:845-850: fnv::FnvHasher over str::hash = the bytes and a 0xff terminator
    h = 0xcbf29ce484222325
    for c in name.encode() + b'\xff':
        h = ((h ^ c) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return var_id(h)


def p2_abs(a): return [abs_(a[0]), abs_(a[1])]
def p2_max(a, b): return [max_(a[0], b[0]), max_(a[1], b[1])]
def p2_circle(ang): return [cos(ang), sin(ang)]                                          # :1031-1033
def p2_spiral(ang): return p2_scale(p2_circle(ang), rad_to_unit(ang))                   # :1035-1037
def p2_qbez(a, b, c, t): return p2_lerp(p2_lerp(a, b, t), p2_lerp(b, c, t), t)          # :1055-1059
def p2_cbez(a, b, c, d, t): return p2_lerp(p2_qbez(a, b, c, t), p2_qbez(b, c, d, t), t)  # :1061-1065
def p2_subst(p, off): return [subst2(p[0], off), subst2(p[1], off)]                     # :1067-1070
def p4_same(v): return [v, v, v, v]                                                      # :1141-1151
def p4_xy(p): return [p[0], p[1]]
def p4_zw(p): return [p[2], p[3]]
def translate(ex, off): return subst2(ex, p2_sub([x(), y()], off))                       # :799-801
def scale(ex, s): return subst2(ex, p2_div([x(), y()], s))                               # :804-806
def scale_at(ex, off, s): return translate(scale(translate(ex, p2_neg(off)), s), off)   # :809-811


def rotate(ex, rad):   # :814-821
    sn, cs = sin(rad), cos(rad)
    ident = [x(), y()]
    return subst2(ex, [p2_dot([cs, neg(sn)], ident), p2_dot([sn, cs], ident)])


def rotate_at(ex, off, rad): return translate(rotate(translate(ex, p2_neg(off)), rad), off)   # :824-826


def from_barycentric(tri, lam):   # :1127-1138
    return [add(add(mul(lam[0], tri[0][0]), mul(lam[1], tri[1][0])), mul(lam[2], tri[2][0])),
            add(add(mul(lam[0], tri[0][1]), mul(lam[1], tri[1][1])), mul(lam[2], tri[2][1]))]


def sd_circle(r): return sub(p2_len([x(), y()]), r)                                      # src/sd.rs:20-22
def sd_box(b):                                                                           # src/sd.rs:23-29
    d = p2_sub(p2_abs([x(), y()]), b)
    return add(p2_len(p2_max(d, [nat(0), nat(0)])), min_(max_(d[0], d[1]), nat(0)))
def sd_inside(sd): return step(neg(sd))                                                  # src/sd.rs:51-53


def var_range_union(a, b):   # :830-834
    if a[1] - a[0] == 0: return b
    if b[1] - b[0] == 0: return a
    return [min(a[0], b[0]), max(a[1], b[1])]


def var_range(ex):   # Expr::var_range, :738-764
    t = ex[0]
    if t in ('Arc', 'Decor'): return var_range(ex[1])
    if t in ('X', 'Y', 'Tau', 'E', 'Nat'): return [0, 0]
    if t == 'Var': return [ex[1], ex[1] + 1]
    if t in UNARY: return var_range(ex[1])
    if t in BINARY: return var_range_union(var_range(ex[1]), var_range(ex[2]))
    if t == 'App': return var_range_union(var_range(ex[2]), var_range(ex[3]))
    if t == 'Let':
        r = var_range(ex[2])
        for i, d in ex[1]:
            r = var_range_union(r, [i, i + 1])
            r = var_range_union(r, var_range(d))
        return r
    raise ValueError(t)


def var_offset(ex, off):   # Expr::var_offset, :767-796
    t = ex[0]
    if t == 'Arc': return var_offset(ex[1], off)
    if t in ('X', 'Y', 'Tau', 'E', 'Nat'): return ex
    if t == 'Var': return ('Var', (ex[1] + off) & 0xFFFFFFFFFFFFFFFF)
    if t in UNARY: return (t, var_offset(ex[1], off))
    if t in BINARY: return (t, var_offset(ex[1], off), var_offset(ex[2], off))
    if t == 'Decor': return ('Decor', var_offset(ex[1], off), ex[2])
    if t == 'App': return ('App', ex[1], var_offset(ex[2], off), var_offset(ex[3], off))
    if t == 'Let': return ('Let', tuple(((i + off) & 0xFFFFFFFFFFFFFFFF, var_offset(d, off)) for i, d in ex[1]), var_offset(ex[2], off))
    raise ValueError(t)


# ---- textures ids (src/textures.rs:14-23) ---------------------------------
ALIGN = 5
def channel(img, ch): return img * ALIGN + ch
def image_width(img): return img * ALIGN + 3
def image_height(img): return img * ALIGN + 4


# ---- bincode (save, src/lib.rs:1216-1224) ---------------------------------
def _enc(ex, out, off):
    # iterative to survive deep trees
    stack = [ex]
    while stack:
        n = stack.pop()
        if isinstance(n, bytes):
            out.append(n)
            continue
        t = n[0]
        out.append(struct.pack('<I', TAG[t] - off))
        if t in ('X', 'Y', 'Tau', 'E'):
            pass
        elif t in ('Var', 'Nat'):
            out.append(struct.pack('<Q', n[1]))
        elif t in UNARY:
            stack.append(n[1])
        elif t in BINARY:
            stack.append(n[2]); stack.append(n[1])
        elif t == 'Let':
            out.append(struct.pack('<Q', len(n[1])))
            stack.append(n[2])
            for i, d in reversed(n[1]):
                stack.append(d)
                stack.append(struct.pack('<Q', i))
        elif t == 'Decor':
            toks = []
            for tok in n[2]:
                sub_out = []
                if isinstance(tok, tuple) and tok and tok[0] == 'TokenExpr':
                    sub_out.append(struct.pack('<I', 0)); _enc(tok[1], sub_out, off)
                elif isinstance(tok, str):
                    b = tok.encode()
                    sub_out.append(struct.pack('<IQ', 1, len(b)) + b)
                else:   # unit variant index 2..12
                    sub_out.append(struct.pack('<I', int(tok)))
                toks.append(b''.join(sub_out))
            stack.append(struct.pack('<Q', len(toks)) + b''.join(toks))
            stack.append(n[1])
        elif t == 'App':
            out.append(struct.pack('<I', n[1]))
            stack.append(n[3]); stack.append(n[2])
        else:
            raise ValueError(t)


def encode_expr(ex, legacy=False):
    if legacy and _has_arc(ex):
        raise ValueError('legacy numbering has no Arc variant')
    out = []
    _enc(ex, out, 1 if legacy else 0)
    return b''.join(out)


def _has_arc(ex):
    stack = [ex]
    seen = set()
    while stack:
        n = stack.pop()
        if id(n) in seen: continue
        seen.add(id(n))
        t = n[0]
        if t == 'Arc': return True
        if t in UNARY: stack.append(n[1])
        elif t in BINARY: stack += [n[1], n[2]]
        elif t == 'Let': stack += [d for _, d in n[1]] + [n[2]]
        elif t == 'Decor': stack.append(n[1])
        elif t == 'App': stack += [n[2], n[3]]
    return False


def encode(size, color, legacy=False):
    """`save(file, (size, color))` → bytes."""
    return struct.pack('<II', size[0], size[1]) + b''.join(encode_expr(c, legacy) for c in color)


def decode(data, legacy=None):
    """Inverse of encode → (size, color).  legacy=None: auto-detect."""
    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 200000))
    if legacy is None:
        for lg in (False, True):
            try:
                return decode(data, lg)
            except Exception:
                continue
        raise ValueError('cannot decode')
    pos_ = [0]
    off = 1 if legacy else 0

    def u32():
        v = struct.unpack_from('<I', data, pos_[0])[0]; pos_[0] += 4; return v

    def u64():
        v = struct.unpack_from('<Q', data, pos_[0])[0]; pos_[0] += 8; return v

    def ex():
        t = TAGS[u32() + off]
        if t in ('X', 'Y', 'Tau', 'E'): return (t,)
        if t in ('Var', 'Nat'): return (t, u64())
        if t in UNARY: return (t, ex())
        if t in BINARY:
            a = ex(); b = ex(); return (t, a, b)
        if t == 'Let':
            k = u64(); vs = []
            for _ in range(k):
                i = u64(); d = ex(); vs.append((i, d))
            return ('Let', tuple(vs), ex())
        if t == 'App':
            i = u32(); a = ex(); b = ex(); return ('App', i, a, b)
        if t == 'Decor':
            a = ex(); k = u64(); toks = []
            for _ in range(k):
                kind = u32()
                if kind == 0: toks.append(('TokenExpr', ex()))
                elif kind == 1:
                    n = u64(); toks.append(data[pos_[0]:pos_[0] + n].decode()); pos_[0] += n
                else: toks.append(kind)
            return ('Decor', a, tuple(toks))
        raise ValueError(t)
    w = u32(); h = u32()
    color = [ex() for _ in range(3)]
    if pos_[0] != len(data):
        raise ValueError('trailing bytes')
    return (w, h), color
