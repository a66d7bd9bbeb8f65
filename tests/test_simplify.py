"""`Expr::simplify` (first half) against the reference's own known answers: every assertion of
`test_simplify_neg_neg`, `test_barycentric`, `test_simplify`, `test_simplify_step` and `test_constant_reduction`
(src/lib.rs:1288-1516, :1694-1720), re-typed with the test builders.  Comparison is on the product's `save` encoding
of the simplified channel, decoded back into a tree."""
import maray_amd as M
from marayb import add, decode, div, encode, mul, nat, neg, recip, step, sub, to_barycentric, x, y


def square(a):
    return mul(a, a)


def simplified(e, times=1):
    s = M.Scene(encode((1, 1), [e, e, e]))
    for _ in range(times):
        s.simplify()
    (_, _), color = decode(s.encode())
    assert color[0] == color[1] == color[2]
    return color[0]


def check(e, want, times=1):
    assert simplified(e, times) == want, (e, want)


def test_simplify_neg_neg():
    e1 = sub(nat(0), nat(1))
    check(e1, neg(nat(1)))
    check(mul(e1, e1), nat(1))


def test_barycentric():
    tri = [(nat(0), nat(0)), (nat(1), nat(0)), (nat(1), nat(1))]
    center = (div(nat(2), nat(3)), div(nat(1), nat(3)))
    for b in to_barycentric(tri, center):
        check(b, recip(nat(3)))


def test_simplify():
    a = neg(neg(nat(1)))
    check(mul(a, a), nat(1))
    # subtraction
    check(sub(div(nat(2), nat(5)), recip(nat(3))), recip(nat(15)))
    check(sub(recip(nat(3)), div(nat(2), nat(5))), neg(recip(nat(15))))
    check(div(x(), nat(1)), x())
    check(sub(div(nat(2), nat(5)), div(nat(1), nat(5))), recip(nat(5)))
    check(sub(div(nat(3), nat(5)), div(nat(1), nat(5))), div(nat(2), nat(5)))
    check(sub(div(nat(3), nat(5)), div(nat(1), nat(2))), recip(nat(10)))
    check(sub(div(nat(4), nat(5)), div(nat(1), nat(2))), div(nat(3), nat(10)))
    # addition
    check(add(div(nat(2), nat(5)), recip(nat(3))), div(nat(11), nat(15)))
    check(add(recip(nat(3)), div(nat(2), nat(5))), div(nat(11), nat(15)))
    check(add(div(nat(2), nat(5)), div(nat(1), nat(5))), div(nat(3), nat(5)))
    check(add(div(nat(3), nat(5)), div(nat(1), nat(5))), div(nat(4), nat(5)))
    check(add(div(nat(3), nat(5)), div(nat(1), nat(2))), div(nat(11), nat(10)))
    check(add(div(nat(4), nat(5)), div(nat(1), nat(2))), div(nat(13), nat(10)))
    check(add(nat(1), sub(div(x(), nat(100)), nat(1))), div(x(), nat(100)))
    check(sub(sub(x(), nat(1)), sub(y(), nat(1))), sub(x(), y()))
    # multiplication
    check(mul(div(nat(2), nat(5)), recip(nat(3))), div(nat(2), nat(15)))
    check(mul(recip(nat(3)), div(nat(2), nat(5))), div(nat(2), nat(15)))
    check(mul(div(nat(2), nat(5)), div(nat(1), nat(5))), div(nat(2), nat(25)))
    check(mul(div(nat(3), nat(5)), div(nat(1), nat(5))), div(nat(3), nat(25)))
    check(mul(div(nat(3), nat(5)), div(nat(1), nat(2))), div(nat(3), nat(10)))
    check(mul(div(nat(4), nat(5)), div(nat(1), nat(2))), div(nat(2), nat(5)))
    check(mul(nat(3), nat(0)), nat(0))
    check(neg(mul(nat(3), nat(0))), nat(0))
    check(add(nat(2), mul(nat(9), nat(1))), nat(11))
    check(mul(mul(nat(2), x()), nat(3)), mul(nat(6), x()))
    # division
    check(div(div(nat(2), nat(5)), recip(nat(3))), div(nat(6), nat(5)))
    check(div(recip(nat(3)), div(nat(2), nat(5))), div(nat(5), nat(6)))
    check(div(div(nat(2), nat(5)), div(nat(1), nat(5))), nat(2))
    check(div(div(nat(3), nat(5)), div(nat(1), nat(5))), nat(3))
    check(div(div(nat(3), nat(5)), div(nat(1), nat(2))), div(nat(6), nat(5)))
    check(div(div(nat(4), nat(5)), div(nat(1), nat(2))), div(nat(8), nat(5)))
    check(div(div(nat(2), nat(3)), nat(5)), div(nat(2), nat(15)))
    check(div(mul(div(x(), nat(2)), nat(2)), nat(3)), div(x(), nat(3)))
    # recip
    check(recip(div(nat(1), nat(3))), nat(3))
    # edge cases
    check(add(div(nat(4), nat(5)), div(nat(3), nat(20))), div(nat(19), nat(20)))
    check(sub(nat(6), div(nat(2), nat(3))), div(nat(16), nat(3)))
    check(sub(div(nat(2), nat(3)), nat(6)), neg(div(nat(16), nat(3))))
    check(add(nat(6), div(nat(2), nat(3))), div(nat(20), nat(3)))
    check(add(div(nat(2), nat(3)), nat(6)), div(nat(20), nat(3)))
    check(add(recip(nat(2)), recip(nat(3))), div(nat(5), nat(6)))
    check(sub(recip(nat(2)), recip(nat(2))), nat(0))
    check(mul(neg(nat(2)), neg(nat(3))), nat(6))
    check(mul(neg(recip(nat(2))), neg(nat(3))), div(nat(3), nat(2)))
    check(sub(neg(recip(nat(2))), neg(nat(3))), div(nat(5), nat(2)))
    check(mul(neg(x()), x()), neg(square(x())))
    check(mul(x(), neg(x())), neg(square(x())))
    check(mul(neg(x()), y()), neg(mul(x(), y())))
    check(mul(x(), neg(y())), neg(mul(x(), y())))
    check(add(neg(x()), y()), sub(y(), x()))
    check(add(x(), neg(y())), sub(x(), y()))
    check(mul(div(x(), nat(2)), div(y(), nat(2))), div(mul(x(), y()), nat(4)))
    check(mul(div(x(), nat(2)), y()), div(mul(x(), y()), nat(2)))
    check(mul(x(), div(y(), nat(2))), div(mul(x(), y()), nat(2)))


def test_simplify_step():
    check(step(nat(1)), nat(1))
    check(step(div(nat(2), nat(1))), nat(1))
    check(step(div(nat(1), nat(2))), nat(1))
    check(step(neg(nat(1))), nat(0))
    check(step(neg(div(nat(1), nat(2)))), nat(0))
    check(step(neg(nat(0))), nat(1))


def test_constant_reduction_through_simplify():
    """src/lib.rs:1694-1720.  The two bare `constant_reduction()` vectors are not reachable through the ABI (simplify
    runs the rewrite rules after it); the third one is five rounds of `simplify` and pins both passes together."""
    e3 = mul(nat(77), sub(div(x(), nat(512)), div(nat(179), nat(256))))
    e6 = mul(nat(3264), sub(div(y(), nat(512)), div(nat(205), nat(512))))
    e7 = sub(div(e3, nat(256)), div(e6, nat(32768)))
    a = div(mul(e7, nat(524288)), nat(47432))
    want = div(sub(mul(nat(77), sub(div(x(), nat(2)), nat(179))),
                   mul(nat(51), sub(div(y(), nat(4)), div(nat(205), nat(4))))), nat(5929))
    check(a, want, times=5)


def test_rules_that_do_not_terminate_are_an_error_not_a_crash():
    """`(x/2) * 1/y`: src/simplify.rs:276-283 rewrites it to `(x * 1/y) / 2`, whose numerator is a quotient again, and
    swaps the two divisors for ever (the reference overflows its stack).  The library reports it."""
    import pytest
    from marayb import recip as r
    with pytest.raises(M.MarayError) as e:
        simplified(mul(div(x(), nat(2)), r(y())))
    assert 'do not terminate' in str(e.value)


def test_the_reference_rules_do_not_terminate_on_examples_chess_and_merged_divisors_do():
    """The term that stops `/root/reference/examples/chess.rs:43`: `to_uv` of grid cell (0,0), triangle 2, arrives at the
    Mul rules (src/simplify.rs:247-327) as `1/8 * -((10400 * (y/64 - 43/5)) / 6240)`.  By hand: the Neg moves out (:265-270),
    `1/8 * B` with B = M/6240 becomes B / 8 (:271-276), its left operand is a quotient so :277-283 makes it
    `(M * 1/8) / 6240`, whose numerator is the quotient M/8 -- no rule merges the two divisors, :277-283 fires again with
    8 and 6240 swapped, for ever.  The restated rules report it (MARAY_E_LIMIT); with MARAY_SIMPLIFY_MERGE_DIVISORS the
    quotient of a quotient is one quotient, `M / 49920`, which constant_reduction (src/constant_reduction.rs:33-48) brings
    to `(5 * ...) / 24`.  Values before and after agree (the oracle evaluates both)."""
    import pytest
    from marayb import recip as r
    from oracle_ffi import eval1
    term = mul(r(nat(8)), neg(div(mul(nat(10400), sub(div(y(), nat(64)), div(nat(43), nat(5)))), nat(6240))))
    with pytest.raises(M.MarayError) as e:
        simplified(term)
    assert e.value.code == -7 and 'do not terminate' in str(e.value)
    s = M.Scene(encode((8, 8), [term, term, term]))
    s.simplify(merge_divisors=True)
    got = decode(s.encode())[1][0]
    inner = sub(div(y(), nat(64)), div(nat(43), nat(5)))
    assert got == neg(div(mul(nat(5), inner), nat(24))), got
    for yv in (0.0, 3.0, 550.4, 1023.0):
        a, b = eval1(term, 0.0, yv), eval1(got, 0.0, yv)
        assert abs(a - b) <= 4e-16 * max(1.0, abs(a)), (yv, a, b)
