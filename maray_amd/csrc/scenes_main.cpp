// scenes_main.cpp — `maray_scenes OUTDIR`: writes the benchmark / test scenes of SURVEY.md §8(d)
// as `.maray` files with the C++ builders (include/maray_builders.hpp).  Counterpart of the
// reference's scene-generating examples (examples/chess.rs, test.rs, test6.rs, test7.rs).
#include <cstdio>
#include <string>

#include "maray_builders.hpp"
#include "maray_hip.h"

using namespace maray_build;

// `e.simplify(mem).compress(mem)` (src/lib.rs:601-614) through the library, which works on encoded scenes: e goes in as
// channel 0 of a scene and comes back as bytes.  merge: MARAY_SIMPLIFY_MERGE_DIVISORS (the reference's own rules do not
// terminate on examples/chess.rs, maray_hip.h).  Exits on failure: this is a command-line tool.
static Expr simplify_compress(const Expr &e, bool merge)
{
    const std::vector<uint8_t> in = encode(1, 1, {e, nat(0), nat(0)});
    maray_scene *sc = nullptr;
    int rc = maray_scene_from_bytes(in.data(), in.size(), &sc);
    if (!rc) rc = maray_scene_simplify_ex(sc, merge ? MARAY_SIMPLIFY_MERGE_DIVISORS : 0);
    if (!rc) rc = maray_scene_compress(sc, nullptr);
    size_t len = 0;
    std::vector<uint8_t> out;
    if (!rc) rc = maray_scene_encode(sc, nullptr, 0, &len);
    if (!rc) { out.resize(len); rc = maray_scene_encode(sc, out.data(), out.size(), &len); }
    maray_scene_free(sc);
    if (rc) { fprintf(stderr, "maray_scenes: simplify / compress: %s\n", maray_last_error()); exit(1); }
    // header (8 bytes), channel 0, then two `Nat(0)` (4 + 8 bytes each: simplify and compress leave a natural alone)
    return encoded(std::vector<uint8_t>(out.begin() + 8, out.end() - 24));
}

// Config 2: radial gradient, c = Sqrt(X*X + Y*Y)
static Color radial() { Expr c = sqrt(add(mul(x(), x()), mul(y(), y()))); return {c, c, c}; }

// Config 3b: every computing variant, one phase per channel
static Color all_ops(uint64_t w, uint64_t h)
{
    Expr u = sub(div(x(), nat(w)), div(nat(1), nat(2)));
    Expr v = sub(div(y(), nat(h)), div(nat(1), nat(2)));
    Expr r = sqrt(add(mul(u, u), mul(v, v)));
    Color out;
    for (uint64_t k = 0; k < 3; k++) {
        Expr s = sin(add(mul(ln(add(nat(1), r)), nat(8)), nat(k)));
        Expr e_ = exp(neg(abs(mul(u, v))));
        out[k] = clamp_u8(mul(nat(255), add(div(nat(1), nat(2)), mul(mul(div(nat(1), nat(2)), s), e_))));
    }
    return out;
}

// Config 5 (pattern of examples/test6.rs:5-9): two textures
static Color textured(uint64_t w)
{
    Color out;
    for (uint32_t c = 0; c < 3; c++) {
        Expr a = app(channel(0, c), mul(x(), div(nat(1), nat(4))), mul(y(), div(nat(1), nat(4))));
        Expr b = app(channel(1, c), mul(add(nat(w), neg(x())), div(nat(1), nat(2))), mul(y(), div(nat(1), nat(8))));
        Expr e_ = max(a, b);
        if (c == 2) e_ = mul(e_, div(app(image_width(1), x(), y()), app(image_width(1), nat(0), nat(0))));
        out[c] = e_;
    }
    return out;
}

// examples/chess.rs:5-50: an 8x8 grid of quads in perspective with a chess texture.  authored: with the example's
// `.simplify(mem).compress(mem)` (:43) -- the file a user of the reference would save at this size; else the bare tree.
static Color chess_board(uint64_t size, bool authored = false)
{
    Point2 p = {div(x(), nat(size)), div(y(), nat(size))};
    Expr texture = set_unit_square(chess(8));
    Grid2 grid{8, 8};
    std::array<Point2, 4> quad = {Point2{recip(nat(5)), recip(nat(2))}, Point2{sub(nat(1), recip(nat(5))), recip(nat(2))},
                                  Point2{nat(0), sub(nat(1), recip(nat(5)))}, Point2{nat(1), sub(nat(1), recip(nat(5)))}};
    Expr shape = nat(0);
    for (uint64_t i = 0; i < 8; i++)
        for (uint64_t j = 0; j < 8; j++) {
            auto cell = grid.cell(i, j, quad);
            TriPair t = quad_to_tri(cell.first, cell.second);
            Expr s1 = subst2(mul(inside_triangle(t.tri[0], {x(), y()}), subst2(texture, to_uv(t.tri[0], t.uv[0], {x(), y()}))), p);
            Expr s2 = subst2(mul(inside_triangle(t.tri[1], {x(), y()}), subst2(texture, to_uv(t.tri[1], t.uv[1], {x(), y()}))), p);
            shape = set_or(shape, set_or(s1, s2));
        }
    if (authored) shape = simplify_compress(shape, true);
    Expr c = mul(shape, nat(255));
    return {c, c, c};
}

// examples/test.rs:9-27: rounded box XOR circle; authored: with its `.simplify(mem).compress(mem)` (:21), which the
// reference's own rules get through
static Color sdf(uint64_t size, bool authored = false)
{
    Point2 p = {div(x(), nat(size)), div(y(), nat(size))};
    Point2 center = {half(), half()};
    Expr tenth = div(nat(1), nat(10));
    Expr a = translate(sd_inside(sd_rounded_box({div(half(), nat(2)), half()}, {tenth, tenth, tenth, tenth})), center);
    Expr b = translate(sd_inside(sd_circle(recip(nat(3)))), center);
    Expr shape = subst2(set_xor(a, b), p);
    if (authored) shape = simplify_compress(shape, false);
    Expr c = mul(shape, nat(255));
    return {c, c, c};
}

// The reference's transforming and curve builders in one small picture (tests/scenes.py transforms() is the same scene by
// the test builders): src/lib.rs:799-826, 1031-1070, 1127-1151, 845-850, 738-796
static Color transforms(uint64_t size)
{
    Point2 p = {div(x(), nat(size)), div(y(), nat(size))};
    Point4 centre = p4_same(half());
    Expr quarter = div(nat(1), nat(4)), eighth = div(nat(1), nat(8));
    Expr box = sd_inside(sd_box({quarter, eighth}));
    box = rotate_at(scale_at(translate(box, p4_xy(centre)), p4_zw(centre), {div(nat(3), nat(2)), half()}), p4_xy(centre), rad_45());
    Expr disk = translate(sd_inside(sd_circle(quarter)), p4_zw(centre));
    Expr r = mul(subst2(set_xor(box, disk), p), nat(255));
    Point2 curve = p2_cbez({nat(0), nat(0)}, {nat(1), nat(0)}, {nat(0), nat(1)}, {nat(1), nat(1)}, x());
    Point2 spiral = p2_spiral(unit_to_rad(y()));
    Expr g = mul(clamp_unit(subst2(p2_len(p2_sub(curve, spiral)), p)), nat(255));
    std::array<Point2, 3> tri = {Point2{nat(0), nat(0)}, Point2{nat(1), nat(0)}, Point2{nat(0), nat(1)}};
    Point2 back = p2_subst(from_barycentric(tri, to_barycentric(tri, {x(), y()})), p);
    Expr u = var("u");
    Expr body = var_offset(let_({{u->u, back[0]}}, mul(clamp_unit(u), nat(255))), 7);
    if (var_range(body) != std::array<uint64_t, 2>{u->u + 7, u->u + 8}) { fprintf(stderr, "maray_scenes: var_range\n"); exit(1); }
    return {r, g, body};
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: maray_scenes OUTDIR [SCENE ...]      (no SCENE: all of them)\n"); return 2; }
    const std::string d = std::string(argv[1]) + "/";
    auto want = [&](const char *name) {
        if (argc == 2) return true;
        for (int i = 2; i < argc; i++) if (name == std::string(argv[i])) return true;
        return false;
    };
    struct Item { const char *name; uint32_t w, h; Color (*make)(); };
    const Item items[] = {
        {"radial_1024", 1024, 1024, [] { return radial(); }},
        {"allops_4096", 4096, 4096, [] { return all_ops(4096, 4096); }},
        {"textured_4096", 4096, 4096, [] { return textured(4096); }},
        {"sdf_512", 512, 512, [] { return sdf(512); }},
        {"transforms_256", 256, 256, [] { return transforms(256); }},
        // examples/test7.rs:5-7: `nat(255) * x() / nat(size[0])`, the operators being mul and div (src/lib.rs: impl Mul / Div for Expr)
        {"test7_128", 128, 128, [] { Expr c = nat(255) * x() / nat(128); return Color{c, c, c}; }},
        {"sdf_512_authored", 512, 512, [] { return sdf(512, true); }},
        {"chess_board_1024", 1024, 1024, [] { return chess_board(1024); }},
        // examples/chess.rs as the example runs it, at its own size and natively at 4096 (what simplify
        // and compress are for -- a scene regenerated at a new size, not a stored one rescaled)
        {"chess_authored_1024", 1024, 1024, [] { return chess_board(1024, true); }},
        {"chess_authored_4096", 4096, 4096, [] { return chess_board(4096, true); }},
    };
    int written = 0;
    for (const Item &it : items) {
        if (!want(it.name)) continue;
        if (!save(d + it.name + ".maray", it.w, it.h, it.make())) { fprintf(stderr, "cannot write into %s\n", argv[1]); return 1; }
        written++;
    }
    if (argc > 2 && written != argc - 2) { fprintf(stderr, "maray_scenes: unknown scene name\n"); return 2; }
    return 0;
}
