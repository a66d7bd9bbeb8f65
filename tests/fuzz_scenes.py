"""Random scene generator for the parity fuzz tests (deterministic per seed)."""
import random

from marayb import (abs_, add, app, arc, channel, decor, div, exp, image_height, image_width, let_, ln, max_, min_, mul,
                    nat, neg, recip, sin, sqrt, step, sub, var_id, x, y)

UN = [neg, abs_, recip, sqrt, step, sin, exp, ln]
BIN = [add, mul, max_, min_]


def _leaf(rng, vars_):
    r = rng.random()
    if vars_ and r < 0.25: return var_id(rng.choice(vars_))
    if r < 0.50: return x()
    if r < 0.70: return y()
    if r < 0.90: return nat(rng.choice([0, 1, 2, 3, 5, 7, 16, 100, 255, 1024, 10 ** 6, 10 ** 12]))
    return rng.choice([('Tau',), ('E',)])


def _expr(rng, depth, vars_, n_tex):
    if depth <= 0 or rng.random() < 0.12:
        return _leaf(rng, vars_)
    r = rng.random()
    if r < 0.08:   # scaled coordinate, keeps many arguments in a moderate range
        return mul(_leaf(rng, vars_), div(nat(1), nat(rng.choice([3, 7, 64, 1000]))))
    if r < 0.40:
        f = rng.choice(UN)
        return f(_expr(rng, depth - 1, vars_, n_tex))
    if r < 0.80:
        f = rng.choice(BIN)
        return f(_expr(rng, depth - 1, vars_, n_tex), _expr(rng, depth - 1, vars_, n_tex))
    if r < 0.86:   # boolean algebra, the shapes the skip regions look for
        a = step(_expr(rng, depth - 1, vars_, n_tex)); b = step(_expr(rng, depth - 1, vars_, n_tex))
        c = rng.choice([min_, max_, mul])(a, rng.choice([b, sub(nat(1), b)]))
        return c if rng.random() < 0.5 else mul(c, _expr(rng, depth - 2, vars_, n_tex))
    if r < 0.885:   # half-plane polygons gating a heavier cone: what the row bounds and SKIP regions target
        def half_plane():
            a, b, c = rng.choice([1, 2, 3, 7]), rng.choice([0, 1, 2, 5]), rng.randrange(0, 200)
            e_ = sub(add(mul(x(), nat(a)), mul(y(), nat(b))), nat(c))
            e_ = mul(e_, div(nat(1), nat(rng.choice([1, 3, 16]))))
            return step(e_ if rng.random() < 0.5 else neg(e_))
        poly = min_(min_(half_plane(), half_plane()), half_plane())
        heavy = _expr(rng, depth - 1, vars_, n_tex)
        for _ in range(3):
            heavy = add(mul(heavy, heavy), sin(add(heavy, x())))
        body = step(heavy) if rng.random() < 0.5 else heavy
        return rng.choice([mul, min_])(poly, body) if body[0] == 'Step' else mul(poly, body)
    if r < 0.90 and n_tex:
        t = rng.randrange(n_tex)
        sel = rng.choice([channel(t, 0), channel(t, 1), channel(t, 2), image_width(t), image_height(t)])
        return app(sel, _expr(rng, depth - 1, vars_, n_tex), _expr(rng, depth - 1, vars_, n_tex))
    if r < 0.93:
        return arc(_expr(rng, depth - 1, vars_, n_tex))
    if r < 0.95:
        return decor(_expr(rng, depth - 1, vars_, n_tex), ['note', 2, ('TokenExpr', x())])
    # Let with definitions that reference earlier ones (like the compressor's output, src/compressor.rs:226-232)
    base = rng.randrange(0, 1000) * 10
    ids, defs = [], []
    for k in range(rng.randint(1, 4)):
        defs.append((base + k, _expr(rng, depth - 2, ids[:], n_tex)))
        ids.append(base + k)
    return let_(defs, _expr(rng, depth - 1, ids, n_tex))


def scene(seed, depth=6, n_tex=0):
    """Three channel expressions.  Let ids are unique per Let (base chosen at random), so the reference's
    id-keyed Cache stays value-transparent; scenes that still alias are skipped by the caller."""
    rng = random.Random(seed)
    return [_expr(rng, depth, [], n_tex) for _ in range(3)]


def polygon_soup(seed, n, w, h, mixed=True):
    """n random textured triangles painted over one another (max chain), the kind of scene `examples/chess.rs` builds
    and the lowering's shape machinery targets: balanced OR trees, group and shape guards, private regions.  Vertices
    are integers inside (and a little outside) the w x h image; each triangle carries a chess pattern in its own
    barycentric frame; channels differ in a gradient term and, with `mixed`, in which triangles they paint (every shape
    is then shared by two OR trees)."""
    from marayb import chess, inside_triangle, to_uv
    rng = random.Random(0x50117 + seed)
    p = [x(), y()]
    tris = []
    for _ in range(n):
        cx, cy = rng.randrange(0, w), rng.randrange(0, h)
        r = rng.choice([6, 12, 25, 60])
        pts = [(nat(max(0, cx + rng.randrange(-r, r + 1))), nat(max(0, cy + rng.randrange(-r, r + 1)))) for _ in range(3)]
        if len({(a[1], b[1]) for a, b in pts}) < 3:
            continue
        inside = inside_triangle(pts, p)
        uv = to_uv(pts, [(nat(0), nat(0)), (nat(1), nat(0)), (nat(0), nat(1))], p)
        pattern = subst_xy(chess(rng.choice([2, 4, 6])), uv[0], uv[1])
        tris.append(min_(inside, pattern) if rng.random() < 0.7 else inside)
    if not tris:
        tris = [step(sub(x(), nat(w)))]

    def paint(items):
        acc = items[0]
        for t in items[1:]:
            acc = max_(acc, t)
        return acc
    grad = mul(add(x(), mul(y(), nat(3))), div(nat(1), nat(w + 3 * h)))
    if mixed == 'colours':      # every shape has its own colour: channel = max_i(shape_i * c_i), a max chain of non-booleans
        cols = [[div(nat(rng.randrange(1, 9)), nat(8)) for _ in range(3)] for _ in tris]
        return [mul(max_(paint([mul(t, c[k]) for t, c in zip(tris, cols)]), mul(grad, div(nat(1), nat(8)))), nat(255)) for k in range(3)]
    if not mixed:       # one mask for the three channels, like examples/chess.rs: the shapes belong to one tree
        m = paint(tris)
        return [mul(m, nat(255)), mul(max_(m, mul(grad, div(nat(1), nat(2)))), nat(255)), mul(add(mul(m, div(nat(3), nat(4))), mul(grad, div(nat(1), nat(4)))), nat(255))]
    return [mul(paint(tris), nat(255)), mul(max_(paint(tris[::2]), mul(grad, div(nat(1), nat(2)))), nat(255)),
            mul(add(mul(paint(tris[1::2] or tris), div(nat(3), nat(4))), mul(grad, div(nat(1), nat(4)))), nat(255))]


def subst_xy(e, ex, ey):
    """e with X -> ex and Y -> ey (tuple expression trees)."""
    if e[0] == 'X': return ex
    if e[0] == 'Y': return ey
    return tuple(subst_xy(c, ex, ey) if isinstance(c, tuple) and c and isinstance(c[0], str) else c for c in e)


def curved_soup(seed, n, w, h, mixed=False):
    """n random CURVED shapes painted over one another: the gating shapes are not half-planes but signed-distance
    circles, boxes and rounded boxes (Sd2, /root/reference/src/sd.rs:28-48: sqrt, abs, max / min of differences), cubic
    Bezier strokes (p2_cbez, src/lib.rs:1061-1065, the curve's y at a clamped parameter of x, a band of |y - c(t)|
    around it) and 1/(1 + d^2) falloffs cut at a level -- so that the sqrt / recip / abs / corner-product paths of the
    lowering's interval bounds (RowBounds::ival, lower.cpp) decide which rectangles of pixels skip a shape.  Each
    shape carries a pattern in its own frame (a cone worth guarding); `mixed`: False = one mask for the three channels,
    True = channels paint different subsets, 'colours' = channel = max_i(shape_i * falloff_i): max trees of non-booleans."""
    from marayb import (abs_, chess, clamp_unit, p2_cbez, p2_lerp, p2_pos, p4_xy, p4_zw, pos, range_, sd_box, sd_circle, sd_inside,
                        set_inv, subst2, translate)
    rng = random.Random(0xC0FFEE + seed)
    shapes, falloffs = [], []
    for _ in range(n):
        cx, cy = rng.randrange(0, w), rng.randrange(0, h)
        r = rng.choice([5, 9, 17, 33, 70])
        off = [nat(cx), nat(cy)]
        kind = rng.choice(['circle', 'box', 'rbox', 'stroke', 'falloff', 'ring'])
        dx, dy = sub(x(), nat(cx)), sub(y(), nat(cy))
        d2 = add(mul(dx, dx), mul(dy, dy))
        if kind == 'circle':
            inside = sd_inside(translate(sd_circle(nat(r)), off))
        elif kind == 'box':
            inside = sd_inside(translate(sd_box([nat(r), nat(max(2, r // rng.choice([1, 2, 3])))]), off))
        elif kind == 'rbox':                              # Sd2::RoundedBox, src/sd.rs:39-47
            b = [nat(r), nat(max(3, r // 2))]
            rr = [div(nat(rng.randrange(1, 5)), nat(2)) for _ in range(4)]
            rxy = p2_pos(x(), p4_xy(rr), p4_zw(rr))
            rx = pos(y(), rxy[0], rxy[1])
            q = [add(sub(abs_(x()), b[0]), rx), add(sub(abs_(y()), b[1]), rx)]
            sd = sub(add(min_(max_(q[0], q[1]), nat(0)), sqrt(add(mul(max_(q[0], nat(0)), max_(q[0], nat(0))), mul(max_(q[1], nat(0)), max_(q[1], nat(0)))))), rx)
            inside = sd_inside(translate(sd, off))
        elif kind == 'stroke':                            # the band |y - c_y(t(x))| <= th over x in [x0, x0 + L)
            L = rng.choice([40, 90, 200])
            x0 = max(0, cx - L // 2)
            t = clamp_unit(mul(sub(x(), nat(x0)), div(nat(1), nat(L))))
            pts = [[nat(x0 + L * k // 3), nat(max(0, cy + rng.randrange(-r, r + 1)))] for k in range(4)]
            c = p2_cbez(pts[0], pts[1], pts[2], pts[3], t)
            th = nat(rng.choice([2, 4, 7]))
            inside = min_(step(sub(th, abs_(sub(y(), c[1])))), range_(nat(x0), nat(x0 + L), x()))
        elif kind == 'falloff':                           # 1 / (1 + d^2 / s) >= 1/k  <=>  d^2 <= (k - 1) s
            s_, k = rng.choice([4, 16, 50]), rng.choice([2, 5, 17])
            inside = step(sub(recip(add(nat(1), mul(d2, div(nat(1), nat(s_))))), div(nat(1), nat(k))))
        else:                                             # ring: r/2 <= d <= r, through two sqrt tests
            d = sqrt(d2)
            inside = min_(step(sub(nat(r), d)), step(sub(d, div(nat(r), nat(2)))))
        u, v = mul(dx, div(nat(1), nat(max(2, r // 2)))), mul(dy, div(nat(1), nat(max(2, r // 2))))
        pat = rng.random()
        if pat < 0.45:
            pattern = subst_xy(chess(rng.choice([2, 4])), u, v)
        elif pat < 0.75:
            pattern = step(sin(add(mul(u, nat(3)), mul(v, v))))
        else:
            pattern = None
        shapes.append(inside if pattern is None else min_(inside, pattern))
        falloffs.append(recip(add(nat(1), mul(d2, div(nat(1), nat(4 * r * r))))))

    def paint(items):
        acc = items[0]
        for t in items[1:]:
            acc = max_(acc, t)
        return acc
    grad = mul(add(x(), mul(y(), nat(3))), div(nat(1), nat(w + 3 * h)))
    if mixed == 'colours':
        return [mul(max_(paint([mul(s_, mul(f, div(nat(k + 2), nat(4)))) for s_, f in zip(shapes, falloffs)]), mul(grad, div(nat(1), nat(8)))), nat(200))
                for k in range(3)]
    if not mixed:
        m = paint(shapes)
        return [mul(m, nat(255)), mul(max_(m, mul(grad, div(nat(1), nat(2)))), nat(255)), mul(add(mul(m, div(nat(3), nat(4))), mul(grad, div(nat(1), nat(4)))), nat(255))]
    return [mul(paint(shapes), nat(255)), mul(max_(paint(shapes[::2]), mul(grad, div(nat(1), nat(2)))), nat(255)),
            mul(add(mul(paint(shapes[1::2] or shapes), div(nat(3), nat(4))), mul(grad, div(nat(1), nat(4)))), nat(255))]


def product_soup(seed, n, w, h):
    """n shapes whose edge tests are Steps of PRODUCTS, SQUARES, ROOTS and RECIPROCALS of terms that are monotone in x and in y --
    what the lowering's monotonicity rule for two varying factors, its square and abs rules and the end-point bounds decide
    on (lower.cpp: monotonicity, RowBounds): (x + a)(x + b) <= k, (x + a)(y + b) >= k, sqrt(x + a)(y + b) <= k, products of
    two non-positive factors, a factor that changes sign inside the image (no monotone bound: interval corners), 1/(x + a)
    against y.  Each shape carries a Step(Sin) pattern, so that skipping it matters."""
    rng = random.Random(0xBEEF + seed)
    shapes = []
    for _ in range(n):
        a, b = rng.randrange(1, 40), rng.randrange(1, 40)
        k = rng.randrange(1, w * h // 4)
        kind = rng.randrange(8)
        X, Y = x(), y()
        if kind == 0:
            e_ = sub(nat(k), mul(add(X, nat(a)), add(X, nat(b))))                          # (x+a)(x+b) <= k: both factors >= 0, increasing
        elif kind == 1:
            e_ = sub(mul(add(X, nat(a)), add(Y, nat(b))), nat(k))                          # (x+a)(y+b) >= k
        elif kind == 2:
            e_ = sub(nat(k), mul(sqrt(add(X, nat(a))), add(Y, nat(b))))                     # sqrt(x+a)(y+b) <= k
        elif kind == 3:
            e_ = sub(nat(k), mul(neg(add(X, nat(a))), neg(add(Y, nat(b)))))                 # two non-positive factors
        elif kind == 4:
            e_ = sub(nat(k), mul(sub(X, nat(rng.randrange(0, w))), sub(X, nat(rng.randrange(0, w)))))      # factors that change sign inside the image
        elif kind == 5:
            c = rng.randrange(0, w)
            e_ = sub(nat(rng.randrange(1, 60) ** 2), add(mul(sub(X, nat(c)), sub(X, nat(c))), mul(sub(Y, nat(b)), sub(Y, nat(b)))))   # a disc, written with squares
        elif kind == 6:
            e_ = sub(mul(recip(add(X, nat(a))), nat(k)), add(Y, nat(b)))                    # k / (x+a) >= y + b
        else:
            e_ = sub(nat(k), mul(mul(add(X, nat(a)), add(X, nat(b))), add(Y, nat(1))))      # a product of three
        inside = step(e_)
        from marayb import chess
        pattern = subst_xy(chess(rng.choice([2, 3])), mul(add(X, Y), div(nat(1), nat(rng.randrange(5, 19)))), mul(sub(X, Y), div(nat(1), nat(rng.randrange(5, 19)))))
        shapes.append(min_(inside, pattern))
    acc = shapes[0]
    for t in shapes[1:]:
        acc = max_(acc, t)
    grad = mul(add(x(), y()), div(nat(1), nat(w + h)))
    return [mul(acc, nat(255)), mul(max_(acc, mul(grad, div(nat(1), nat(2)))), nat(255)), mul(add(mul(acc, div(nat(1), nat(2))), mul(grad, div(nat(1), nat(2)))), nat(255))]
