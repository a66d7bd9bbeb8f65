"""Pin the oracle against the reference's own known answers.

Each assert below is one `assert_eq!` of `it_works` (src/lib.rs:1241-1285),
in the same order, evaluated by the oracle's eval2 restatement.
"""
import math

from marayb import (add, clamp, cos, div, lerp, mul, nat, neg, p2_len, pi, range_, step, step_at, x, exp, ln, sqrt,
                    abs_, recip, max_, min_, sin, let_, var_id, y, arc, decor, encode)
from oracle_ffi import Scene, eval1, lib


def test_it_works_known_answers():
    a = mul(x(), x())
    assert eval1(a, 2.0) == 4.0

    a = neg(nat(1))
    assert eval1(a, 0.0) == -1.0

    a = div(nat(1), nat(2))
    assert eval1(a, 0.0) == 0.5

    a = pi()
    assert eval1(a, 0.0) == 3.141592653589793

    a = lerp(neg(nat(1)), nat(1), x())
    assert eval1(a, 0.0) == -1.0
    assert eval1(a, 1.0) == 1.0

    a = cos(x())
    assert eval1(a, 0.0) == 1.0

    a = step(x())
    assert eval1(a, -1.0) == 0.0
    assert eval1(a, 0.0) == 1.0
    assert eval1(a, 1.0) == 1.0

    a = step_at(nat(2), x())
    assert eval1(a, 1.0) == 0.0
    assert eval1(a, 2.0) == 1.0

    a = range_(nat(1), nat(2), x())
    assert eval1(a, 0.5) == 0.0
    assert eval1(a, 1.5) == 1.0
    assert eval1(a, 2.5) == 0.0

    a = p2_len([x(), x()])
    assert eval1(a, 0.0) == 0.0
    assert eval1(a, 1.0) == math.sqrt(2.0)

    a = clamp(nat(1), nat(5), x())
    assert eval1(a, 0.0) == 1.0
    assert eval1(a, 1.0) == 1.0
    assert eval1(a, 5.0) == 5.0
    assert eval1(a, 6.0) == 5.0


def test_scalar_semantics_corner_cases():
    """eval2 corner cases the reference defines by its Rust expressions
    (src/lib.rs:636-658, src/cache.rs:40)."""
    nan = float('nan')
    assert eval1(step(x()), -0.0) == 1.0                 # -0.0 >= 0.0
    assert eval1(step(x()), nan) == 0.0                  # NaN >= 0.0 is false
    assert eval1(max_(x(), nat(3)), nan) == 3.0          # f64::max ignores NaN
    assert eval1(min_(nat(3), x()), nan) == 3.0
    assert math.isnan(eval1(var_id(7), 0.0))             # unknown Var -> NaN
    assert eval1(recip(x()), 0.0) == math.inf
    assert eval1(recip(x()), -0.0) == -math.inf
    assert math.isnan(eval1(sqrt(x()), -1.0))
    assert eval1(abs_(x()), -2.5) == 2.5
    assert eval1(('Tau',), 0.0) == 6.283185307179586
    assert eval1(('E',), 0.0) == 2.718281828459045
    assert eval1(nat(107374182400), 0.0) == 107374182400.0
    assert eval1(exp(x()), 1.0) == math.exp(1.0)         # platform libm, like Rust std
    assert eval1(ln(x()), 10.0) == math.log(10.0)
    assert eval1(sin(x()), 1e22) == math.sin(1e22)
    # Let replaces the context (src/lib.rs:659-662); Arc and Decor are transparent.
    a = let_([(0, add(x(), nat(1))), (1, mul(var_id(0), var_id(0)))], add(var_id(1), var_id(0)))
    assert eval1(a, 2.0) == 12.0
    assert eval1(arc(a), 2.0) == 12.0
    assert eval1(decor(a, ['hello', 2, ('TokenExpr', y())]), 2.0) == 12.0
    inner = let_([(5, y())], add(var_id(5), var_id(0)))   # Var(0) not in the inner ctx, not cached -> NaN
    assert math.isnan(eval1(let_([(0, x())], inner), 2.0))


def test_cast_u8_is_rust_saturating_cast():
    L = lib()
    for v, want in [(float('nan'), 0), (-1.0, 0), (-0.0, 0), (0.0, 0), (0.999, 0), (1.0, 1), (254.999, 254),
                    (255.0, 255), (255.5, 255), (1e300, 255), (float('inf'), 255), (float('-inf'), 0), (127.5, 127)]:
        assert L.oracle_cast_u8(v) == want, v


def test_reader_rejects_garbage_and_autodetects():
    import pytest
    with pytest.raises(ValueError):
        Scene(b'\x01\x00\x00\x00')
    with pytest.raises(ValueError):
        Scene(encode((4, 4), [x(), x(), x()]) + b'\x00')
    s = Scene(encode((4, 5), [x(), y(), nat(3)]))
    assert s.size == (4, 5) and not s.legacy
    s = Scene(encode((4, 5), [x(), y(), nat(3)], legacy=True))
    assert s.size == (4, 5) and s.legacy
