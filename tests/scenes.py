"""Synthetic scenes of SURVEY.md §8(d) (configs 2, 3b, 5) built with the test builders."""
import numpy as np

from marayb import (abs_, add, app, channel, clamp_u8, div, exp, image_width, ln, max_, mul, nat, neg, sin, sqrt, sub,
                    x, y)


def radial_gradient():
    """Config 2: c = Sqrt(Add(Mul(X,X), Mul(Y,Y))), color = [c,c,c]."""
    c = sqrt(add(mul(x(), x()), mul(y(), y())))
    return [c, c, c]


def all_ops(w, h):
    """Config 3b: every computing variant of Expr, different phase per channel:
    clamp_u8(255 * (1/2 + 1/2 * sin(ln(1 + sqrt(u^2+v^2)) * 8 + k) * exp(-abs(u*v)))), u = x/W - 1/2, v = y/H - 1/2."""
    u = sub(div(x(), nat(w)), div(nat(1), nat(2)))
    v = sub(div(y(), nat(h)), div(nat(1), nat(2)))
    r = sqrt(add(mul(u, u), mul(v, v)))
    out = []
    for k in (0, 1, 2):
        s = sin(add(mul(ln(add(nat(1), r)), nat(8)), nat(k)))
        e = exp(neg(abs_(mul(u, v))))
        val = mul(nat(255), add(div(nat(1), nat(2)), mul(mul(div(nat(1), nat(2)), s), e)))
        out.append(clamp_u8(val))
    return out


def splitmix64(seed, n):
    """n bytes of splitmix64 output (little-endian words), seed 0x6d61726179 = "maray"."""
    out = bytearray()
    s = seed & 0xFFFFFFFFFFFFFFFF
    while len(out) < n:
        s = (s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 31
        out += z.to_bytes(8, 'little')
    return bytes(out[:n])


def textures(scale=1):
    """T0 (1024/scale)^2, T1 (2048/scale) x (512/scale), RGB8 from splitmix64."""
    w0 = h0 = 1024 // scale
    w1, h1 = 2048 // scale, 512 // scale
    raw = splitmix64(0x6d61726179, (w0 * h0 + w1 * h1) * 3)
    t0 = np.frombuffer(raw[:w0 * h0 * 3], np.uint8).reshape(h0, w0, 3).copy()
    t1 = np.frombuffer(raw[w0 * h0 * 3:], np.uint8).reshape(h1, w1, 3).copy()
    return [t0, t1]


def textured(w):
    """Config 5 (pattern of examples/test6.rs:5-9): per channel
    Max(App(channel(0,c), X/4, Y/4), App(channel(1,c), (W - X)/2, Y/8)), plus one image_width(1) use."""
    out = []
    for c in (0, 1, 2):
        a = app(channel(0, c), mul(x(), div(nat(1), nat(4))), mul(y(), div(nat(1), nat(4))))
        b = app(channel(1, c), mul(add(nat(w), neg(x())), div(nat(1), nat(2))), mul(y(), div(nat(1), nat(8))))
        e = max_(a, b)
        if c == 2:   # selectors 3/4: scale blue by width(T1)/width(T1) == 1 through an App
            e = mul(e, div(app(image_width(1), x(), y()), app(image_width(1), nat(0), nat(0))))
        out.append(e)
    return out


def shapes_through_inf_and_nan():
    """Guarded shapes whose factors pass through inf and NaN inside a 192 x 24 image: 1/(x-40) (a pole), exp of a large
    argument (overflow from x = 118), 0 * inf and inf - inf (NaN; step(NaN) = 0), next to honest linear edges."""
    from marayb import add, exp, max_, min_, mul, nat, recip, sin, step, sub, x, y
    pole = recip(sub(x(), nat(40)))
    blow = exp(mul(sub(x(), nat(100)), nat(40)))
    f1 = step(add(mul(pole, sub(y(), nat(7))), nat(1)))
    f2 = step(sub(blow, mul(blow, nat(2))))
    f3 = step(sub(nat(150), add(x(), mul(y(), nat(2)))))
    f4 = step(sub(add(mul(x(), nat(3)), y()), nat(60)))
    heavy = step(sin(mul(add(mul(x(), x()), mul(y(), nat(3))), recip(nat(7)))))
    for k in range(5):
        heavy = min_(heavy, step(add(mul(x(), nat(k + 1)), sub(nat(400 * (k + 1)), mul(y(), nat(9))))))
    shape_a = mul(mul(mul(f3, f4), f1), heavy)
    shape_b = mul(mul(f3, max_(f2, f4)), mul(heavy, step(sub(y(), nat(3)))))
    return [mul(max_(shape_a, shape_b), nat(255)), mul(shape_a, add(x(), nat(1))), add(shape_b, mul(f1, nat(2)))]


def transforms(size):
    """The reference's transforming and curve builders in one small picture (src/lib.rs:799-826, 1031-1070, 1127-1151, 845-850,
    738-796): R = a box scaled and rotated about the picture's centre XOR a disk, G = distance between a cubic Bezier point
    (parameter x) and a spiral point (angle from y), B = a point sent to barycentric coordinates and back, through a Let whose
    variable is named by `var` and moved by `var_offset`."""
    from marayb import (clamp_unit, from_barycentric, half, let_, p2_cbez, p2_len, p2_spiral, p2_sub, p2_subst, p4_same, p4_xy,
                        p4_zw, rad_45, rotate_at, scale_at, sd_box, sd_circle, sd_inside, set_xor, subst2, to_barycentric,
                        translate, unit_to_rad, var, var_offset, var_range)
    p = [div(x(), nat(size)), div(y(), nat(size))]
    centre = p4_same(half())
    quarter, eighth = div(nat(1), nat(4)), div(nat(1), nat(8))
    box = sd_inside(sd_box([quarter, eighth]))
    box = rotate_at(scale_at(translate(box, p4_xy(centre)), p4_zw(centre), [div(nat(3), nat(2)), half()]), p4_xy(centre), rad_45())
    disk = translate(sd_inside(sd_circle(quarter)), p4_zw(centre))
    r = mul(subst2(set_xor(box, disk), p), nat(255))
    curve = p2_cbez([nat(0), nat(0)], [nat(1), nat(0)], [nat(0), nat(1)], [nat(1), nat(1)], x())
    spiral = p2_spiral(unit_to_rad(y()))
    g = mul(clamp_unit(subst2(p2_len(p2_sub(curve, spiral)), p)), nat(255))
    tri = [[nat(0), nat(0)], [nat(1), nat(0)], [nat(0), nat(1)]]
    back = p2_subst(from_barycentric(tri, to_barycentric(tri, [x(), y()])), p)
    u = var('u')
    body = var_offset(let_([(u[1], back[0])], mul(clamp_unit(u), nat(255))), 7)
    assert var_range(body) == [u[1] + 7, u[1] + 8]
    return [r, g, body]


def ops_on_a_guarded_mask(w, h):
    """Every kind of op fed with the VALUE of guarded shapes (textured triangles, each behind a rectangle guard), directly and
    through arithmetic with constants: in the specialised kernel's variant for tiles without a guard bit those values are the
    literal 0.0, so every op there meets operands that are numbers, not vector values — texture coordinates included (a
    random scene of round 4's second GPU sweep, seed 6462, was the first to do that to an App: its four-wide form did not
    compile).  Two textures (ids 0..4 and 5..9)."""
    from fuzz_scenes import subst_xy
    from marayb import chess, inside_triangle, min_, recip, step, to_uv
    p = [x(), y()]
    masks = []
    for pts in ([(w // 8, h // 8), (w // 2, h // 6), (w // 5, h - h // 8)], [(w // 2, h // 3), (w - w // 8, h // 8), (w - w // 6, h - h // 6)],
                [(w // 3, h - h // 3), (w // 2 + w // 8, h // 2), (w // 2, h - h // 10)]):
        pts = [(nat(a), nat(b)) for a, b in pts]
        uv = to_uv(pts, [(nat(0), nat(0)), (nat(1), nat(0)), (nat(0), nat(1))], p)
        masks.append(min_(inside_triangle(pts, p), subst_xy(chess(4), uv[0], uv[1])))
    m0, m1, m2 = masks
    c0 = add(add(abs_(sub(m0, div(nat(1), nat(2)))), recip(add(m1, nat(1)))), add(sqrt(m2), mul(sin(m0), exp(m1))))
    c0 = add(c0, add(ln(add(m2, nat(1))), add(step(sub(m0, div(nat(1), nat(2)))), step(sin(mul(m1, nat(3)))))))
    c1 = add(app(channel(0, 1), mul(m0, nat(20)), y()), add(app(channel(1, 0), x(), mul(m1, nat(9))), app(channel(0, 2), m2, mul(m0, nat(3)))))
    c1 = add(c1, add(app(channel(1, 2), mul(m2, nat(7)), mul(y(), div(nat(1), nat(2)))), neg(m1)))
    c2 = max_(mul(max_(max_(m0, m1), m2), nat(255)), add(mul(sqrt(add(m0, m1)), nat(40)), mul(x(), div(nat(1), nat(8)))))
    return [mul(c0, nat(30)), mul(c1, div(nat(1), nat(2))), c2]
