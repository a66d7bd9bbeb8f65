"""Pin the oracle's fix_color against `test_var_fixer` (src/lib.rs:1517-1691).

The expected trees are the reference test's `b`/`b2` values, re-typed with
the test builders; comparison is on the bincode encoding of each channel.
"""
from marayb import add, encode, encode_expr, let_, nat, sub, var_id, x, y
from oracle_ffi import Scene


def fixed(color):
    s = Scene(encode((1, 1), color))
    s.fix_color()
    return [s.encode_channel(c) for c in range(3)]


def nested(id0, d0, id1, d1, body):
    return let_([(id0, d0)], let_([(id1, d1)], var_id(body)))


def test_same_channels_share_ids():
    a = nested(0, x(), 0, y(), 0)
    b = nested(0, x(), 1, y(), 1)
    assert fixed([a, a, a]) == [encode_expr(b)] * 3


def test_distinct_inner_definitions_get_fresh_ids():
    a2 = [nested(0, x(), 0, add(y(), nat(k)), 0) for k in (1, 2, 3)]
    b2 = [nested(0, x(), k, add(y(), nat(k)), k) for k in (1, 2, 3)]
    assert fixed(a2) == [encode_expr(b) for b in b2]


def test_distinct_outer_and_inner_definitions():
    a2 = [nested(0, sub(x(), nat(k)), 0, add(y(), nat(k)), 0) for k in (1, 2, 3)]
    b2 = [nested(2 * k - 2, sub(x(), nat(k)), 2 * k - 1, add(y(), nat(k)), 2 * k - 1) for k in (1, 2, 3)]
    assert fixed(a2) == [encode_expr(b) for b in b2]


def test_chess_fix_color_is_identity(chess_bytes):
    """ids already 0..858, channels equal."""
    from marayb import decode
    s = Scene(chess_bytes)
    s.fix_color()
    _, color = decode(chess_bytes)
    for c in range(3):
        assert s.encode_channel(c) == encode_expr(color[c])
