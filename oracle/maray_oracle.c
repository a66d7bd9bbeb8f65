/*
 * maray_oracle.c — CPU ORACLE (test infrastructure; see maray_oracle.h).
 *
 * Plain-C restatement of the reference interpreter path of
 * advancedresearch/maray 0.3.8.  Every function cites the reference file:line
 * it follows.  Deliberately a naive tree walk, like the reference.
 */
#define _GNU_SOURCE
#include "maray_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- Expr (src/lib.rs:101-149), current tag numbering ------------------ */
enum {
    T_ARC = 0, T_X, T_Y, T_TAU, T_E, T_VAR, T_NAT, T_NEG, T_ABS, T_RECIP, T_SQRT,
    T_STEP, T_SIN, T_EXP, T_LN, T_ADD, T_MUL, T_MAX, T_MIN, T_LET, T_DECOR, T_APP,
    T_COUNT
};

typedef struct OExpr OExpr;
typedef struct OCtx OCtx;
typedef struct OToken OToken;

struct OToken {          /* src/token.rs:10-38 */
    uint32_t kind;       /* 0 TokenExpr, 1 Str, 2..12 unit variants */
    OExpr *e;
    const uint8_t *s;
    uint64_t slen;
};

struct OCtx {            /* src/lib.rs:51-55 */
    uint64_t n;
    uint64_t *ids;
    OExpr **defs;
};

struct OExpr {
    uint8_t tag;
    uint32_t app_id;     /* App */
    uint64_t u;          /* Var id / Nat value */
    OExpr *a, *b;        /* children (a only for unary/Arc/Decor; Let body in a) */
    OCtx *ctx;           /* Let */
    OToken *toks;        /* Decor */
    uint64_t ntoks;
    uint64_t hash;       /* structural hash (valid after hash_expr) */
    uint8_t hashed;
};

/* ---- arena ------------------------------------------------------------- */
typedef struct Block { struct Block *next; size_t used, cap; } Block;

struct oracle_scene {
    Block *blocks;
    uint32_t w, h;
    int legacy;
    int fixed;
    OExpr *color[3];
    uint8_t *bytes;      /* private copy of the file (Str tokens point here) */
    size_t nbytes;
    uint64_t n_let_vars; /* upper bound on Cache entries */
};

static __thread char g_err[256];
const char *oracle_last_error(void) { return g_err; }
static void set_err(const char *m) { snprintf(g_err, sizeof g_err, "%s", m); }

static void *arena_alloc(oracle_scene *s, size_t n)
{
    n = (n + 15) & ~(size_t)15;
    if (!s->blocks || s->blocks->used + n > s->blocks->cap) {
        size_t cap = n > (1u << 20) ? n : (1u << 20);
        Block *b = (Block *)malloc(sizeof(Block) + cap);
        if (!b) abort();
        b->next = s->blocks; b->used = 0; b->cap = cap;
        s->blocks = b;
    }
    void *p = (char *)(s->blocks + 1) + s->blocks->used;
    s->blocks->used += n;
    memset(p, 0, n);
    return p;
}

static OExpr *new_expr(oracle_scene *s, int tag)
{
    OExpr *e = (OExpr *)arena_alloc(s, sizeof(OExpr));
    e->tag = (uint8_t)tag;
    return e;
}

/* ---- bincode reader (maray::open, src/lib.rs:1227-1235) ----------------- */
typedef struct {
    oracle_scene *s;
    const uint8_t *p;
    size_t len, pos;
    int legacy;
    int err;
    int depth;
} Rd;

static uint32_t rd_u32(Rd *r)
{
    if (r->err || r->pos + 4 > r->len) { r->err = 1; return 0; }
    uint32_t v; memcpy(&v, r->p + r->pos, 4); r->pos += 4; return v;
}
static uint64_t rd_u64(Rd *r)
{
    if (r->err || r->pos + 8 > r->len) { r->err = 1; return 0; }
    uint64_t v; memcpy(&v, r->p + r->pos, 8); r->pos += 8; return v;
}

static OExpr *rd_expr(Rd *r);

static void rd_token(Rd *r, OToken *t)
{
    t->kind = rd_u32(r);
    if (r->err) return;
    if (t->kind == 0) t->e = rd_expr(r);
    else if (t->kind == 1) {
        t->slen = rd_u64(r);
        if (r->err || t->slen > r->len - r->pos) { r->err = 1; return; }
        t->s = r->p + r->pos; r->pos += (size_t)t->slen;
    } else if (t->kind > 12) r->err = 1;
}

static OExpr *rd_expr(Rd *r)
{
    if (r->err) return NULL;
    if (++r->depth > 20000) { r->err = 1; return NULL; }
    uint32_t tag = rd_u32(r);
    if (r->legacy) tag += 1;   /* legacy files have no Arc variant: tag = current - 1 */
    if (r->err || tag >= T_COUNT) { r->err = 1; return NULL; }
    OExpr *e = new_expr(r->s, (int)tag);
    switch (tag) {
    case T_X: case T_Y: case T_TAU: case T_E: break;
    case T_VAR: case T_NAT: e->u = rd_u64(r); break;
    case T_ARC: case T_NEG: case T_ABS: case T_RECIP: case T_SQRT: case T_STEP:
    case T_SIN: case T_EXP: case T_LN:
        e->a = rd_expr(r); break;
    case T_ADD: case T_MUL: case T_MAX: case T_MIN:
        e->a = rd_expr(r); e->b = rd_expr(r); break;
    case T_LET: {
        uint64_t n = rd_u64(r);
        if (r->err || n > (r->len - r->pos) / 12) { r->err = 1; break; }
        OCtx *c = (OCtx *)arena_alloc(r->s, sizeof(OCtx));
        c->n = n;
        c->ids = (uint64_t *)arena_alloc(r->s, (size_t)(n ? n : 1) * sizeof(uint64_t));
        c->defs = (OExpr **)arena_alloc(r->s, (size_t)(n ? n : 1) * sizeof(OExpr *));
        for (uint64_t i = 0; i < n && !r->err; i++) {
            c->ids[i] = rd_u64(r);
            c->defs[i] = rd_expr(r);
        }
        r->s->n_let_vars += n;
        e->ctx = c;
        e->a = rd_expr(r);
        break;
    }
    case T_DECOR: {
        e->a = rd_expr(r);
        uint64_t n = rd_u64(r);
        if (r->err || n > (r->len - r->pos) / 4) { r->err = 1; break; }
        e->ntoks = n;
        e->toks = (OToken *)arena_alloc(r->s, (size_t)(n ? n : 1) * sizeof(OToken));
        for (uint64_t i = 0; i < n && !r->err; i++) rd_token(r, &e->toks[i]);
        break;
    }
    case T_APP:
        e->app_id = rd_u32(r);
        e->a = rd_expr(r); e->b = rd_expr(r); break;
    }
    r->depth--;
    return r->err ? NULL : e;
}

static void free_blocks(oracle_scene *s)
{
    Block *b = s->blocks;
    while (b) { Block *n = b->next; free(b); b = n; }
    s->blocks = NULL;
}

static int try_parse(oracle_scene *s, int legacy)
{
    Rd r = { s, s->bytes, s->nbytes, 0, legacy, 0, 0 };
    s->n_let_vars = 0;
    s->w = rd_u32(&r);
    s->h = rd_u32(&r);
    for (int c = 0; c < 3; c++) s->color[c] = rd_expr(&r);
    if (r.err || r.pos != r.len) return 0;   /* must consume the buffer exactly */
    s->legacy = legacy;
    return 1;
}

oracle_scene *oracle_scene_from_bytes(const uint8_t *buf, size_t len, int legacy)
{
    oracle_scene *s = (oracle_scene *)calloc(1, sizeof *s);
    s->bytes = (uint8_t *)malloc(len ? len : 1);
    memcpy(s->bytes, buf, len);
    s->nbytes = len;
    int ok = 0;
    if (legacy >= 0) ok = try_parse(s, legacy);
    else {
        ok = try_parse(s, 0);
        if (!ok) { free_blocks(s); ok = try_parse(s, 1); }
    }
    if (!ok) {
        set_err("bincode decode failed (neither tag numbering consumes the buffer exactly)");
        oracle_scene_free(s);
        return NULL;
    }
    return s;
}

oracle_scene *oracle_scene_open(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) { set_err("cannot open file"); return NULL; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *b = (uint8_t *)malloc(n > 0 ? (size_t)n : 1);
    size_t got = fread(b, 1, (size_t)n, f);
    fclose(f);
    oracle_scene *s = (got == (size_t)n) ? oracle_scene_from_bytes(b, got, -1) : NULL;
    free(b);
    return s;
}

void oracle_scene_free(oracle_scene *s)
{
    if (!s) return;
    free_blocks(s);
    free(s->bytes);
    free(s);
}

void oracle_scene_size(const oracle_scene *s, uint32_t *w, uint32_t *h) { *w = s->w; *h = s->h; }
int oracle_scene_is_legacy(const oracle_scene *s) { return s->legacy; }

static void count_nodes(const OExpr *e, uint64_t hist[T_COUNT], uint64_t *total)
{
    if (!e) return;
    hist[e->tag]++; (*total)++;
    if (e->tag == T_LET)
        for (uint64_t i = 0; i < e->ctx->n; i++) count_nodes(e->ctx->defs[i], hist, total);
    if (e->tag == T_DECOR)
        for (uint64_t i = 0; i < e->ntoks; i++)
            if (e->toks[i].kind == 0) count_nodes(e->toks[i].e, hist, total);
    count_nodes(e->a, hist, total);
    count_nodes(e->b, hist, total);
}

uint64_t oracle_scene_node_count(const oracle_scene *s, int c)
{
    uint64_t hist[T_COUNT] = {0}, total = 0;
    count_nodes(s->color[c], hist, &total);
    return total;
}

void oracle_scene_tag_histogram(const oracle_scene *s, int c, uint64_t out[22])
{
    uint64_t total = 0;
    memset(out, 0, 22 * sizeof(uint64_t));
    count_nodes(s->color[c], out, &total);
}

/* ---- encoder (current numbering; inverse of the reader) ---------------- */
typedef struct { uint8_t *p; size_t cap, pos; } Wr;
static void wr_bytes(Wr *w, const void *d, size_t n)
{
    if (w->pos + n <= w->cap) memcpy(w->p + w->pos, d, n);
    w->pos += n;
}
static void wr_u32(Wr *w, uint32_t v) { wr_bytes(w, &v, 4); }
static void wr_u64(Wr *w, uint64_t v) { wr_bytes(w, &v, 8); }
static void wr_expr(Wr *w, const OExpr *e)
{
    wr_u32(w, e->tag);
    switch (e->tag) {
    case T_VAR: case T_NAT: wr_u64(w, e->u); break;
    case T_LET:
        wr_u64(w, e->ctx->n);
        for (uint64_t i = 0; i < e->ctx->n; i++) { wr_u64(w, e->ctx->ids[i]); wr_expr(w, e->ctx->defs[i]); }
        wr_expr(w, e->a);
        break;
    case T_DECOR:
        wr_expr(w, e->a);
        wr_u64(w, e->ntoks);
        for (uint64_t i = 0; i < e->ntoks; i++) {
            wr_u32(w, e->toks[i].kind);
            if (e->toks[i].kind == 0) wr_expr(w, e->toks[i].e);
            else if (e->toks[i].kind == 1) { wr_u64(w, e->toks[i].slen); wr_bytes(w, e->toks[i].s, (size_t)e->toks[i].slen); }
        }
        break;
    case T_APP: wr_u32(w, e->app_id); wr_expr(w, e->a); wr_expr(w, e->b); break;
    default:
        if (e->a) wr_expr(w, e->a);
        if (e->b) wr_expr(w, e->b);
    }
}

size_t oracle_scene_encode_channel(const oracle_scene *s, int c, uint8_t *out, size_t cap)
{
    Wr w = { out, out ? cap : 0, 0 };
    wr_expr(&w, s->color[c]);
    return w.pos;
}

/* ---- structural hash / equality (derive(Hash, PartialEq) on Expr) ------ */
static uint64_t mix(uint64_t h, uint64_t v)
{
    h ^= v + 0x9e3779b97f4a7c15ULL + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdULL;
    return h ^ (h >> 32);
}

static uint64_t hash_expr(OExpr *e)
{
    if (!e) return 0;
    if (e->hashed) return e->hash;
    uint64_t h = mix(0x1234, e->tag);
    h = mix(h, e->u); h = mix(h, e->app_id);
    if (e->tag == T_LET) {
        h = mix(h, e->ctx->n);
        for (uint64_t i = 0; i < e->ctx->n; i++) { h = mix(h, e->ctx->ids[i]); h = mix(h, hash_expr(e->ctx->defs[i])); }
    }
    if (e->tag == T_DECOR) {
        h = mix(h, e->ntoks);
        for (uint64_t i = 0; i < e->ntoks; i++) {
            h = mix(h, e->toks[i].kind);
            if (e->toks[i].kind == 0) h = mix(h, hash_expr(e->toks[i].e));
            if (e->toks[i].kind == 1) for (uint64_t k = 0; k < e->toks[i].slen; k++) h = mix(h, e->toks[i].s[k]);
        }
    }
    h = mix(h, hash_expr(e->a));
    h = mix(h, hash_expr(e->b));
    e->hash = h; e->hashed = 1;
    return h;
}

static int expr_eq(OExpr *x, OExpr *y)
{
    if (x == y) return 1;
    if (!x || !y) return 0;
    if (x->tag != y->tag || x->u != y->u || x->app_id != y->app_id) return 0;
    if (hash_expr(x) != hash_expr(y)) return 0;
    if (x->tag == T_LET) {
        if (x->ctx->n != y->ctx->n) return 0;
        for (uint64_t i = 0; i < x->ctx->n; i++)
            if (x->ctx->ids[i] != y->ctx->ids[i] || !expr_eq(x->ctx->defs[i], y->ctx->defs[i])) return 0;
    }
    if (x->tag == T_DECOR) {
        if (x->ntoks != y->ntoks) return 0;
        for (uint64_t i = 0; i < x->ntoks; i++) {
            OToken *a = &x->toks[i], *b = &y->toks[i];
            if (a->kind != b->kind) return 0;
            if (a->kind == 0 && !expr_eq(a->e, b->e)) return 0;
            if (a->kind == 1 && (a->slen != b->slen || memcmp(a->s, b->s, (size_t)a->slen))) return 0;
        }
    }
    return expr_eq(x->a, y->a) && expr_eq(x->b, y->b);
}

/* ---- VarFixer (src/var_fixer.rs:8-70) ---------------------------------- */
typedef struct { OExpr *key; uint64_t id; } IdEnt;
typedef struct {
    oracle_scene *s;
    IdEnt *tab; size_t cap, n;       /* ids: HashMap<Expr,u64>   (:10) */
    uint64_t var_count;              /* var_count                (:12) */
} VarFixer;

typedef struct { uint64_t *old_id, *new_id; size_t n; } RenCtx;   /* Vec<(u64,u64)> */

static void vf_grow(VarFixer *vf)
{
    size_t ncap = vf->cap ? vf->cap * 2 : 1024;
    IdEnt *nt = (IdEnt *)calloc(ncap, sizeof(IdEnt));
    for (size_t i = 0; i < vf->cap; i++)
        if (vf->tab[i].key) {
            size_t j = (size_t)hash_expr(vf->tab[i].key) & (ncap - 1);
            while (nt[j].key) j = (j + 1) & (ncap - 1);
            nt[j] = vf->tab[i];
        }
    free(vf->tab); vf->tab = nt; vf->cap = ncap;
}

static int vf_get(VarFixer *vf, OExpr *k, uint64_t *id)
{
    if (!vf->cap) return 0;
    size_t j = (size_t)hash_expr(k) & (vf->cap - 1);
    while (vf->tab[j].key) {
        if (expr_eq(vf->tab[j].key, k)) { *id = vf->tab[j].id; return 1; }
        j = (j + 1) & (vf->cap - 1);
    }
    return 0;
}

static void vf_put(VarFixer *vf, OExpr *k, uint64_t id)
{
    if ((vf->n + 1) * 2 > vf->cap) vf_grow(vf);
    size_t j = (size_t)hash_expr(k) & (vf->cap - 1);
    while (vf->tab[j].key) j = (j + 1) & (vf->cap - 1);
    vf->tab[j].key = k; vf->tab[j].id = id; vf->n++;
}

/* VarFixer::fix (src/var_fixer.rs:25-70).  Builds a new tree. */
static OExpr *vf_fix(VarFixer *vf, OExpr *e, const RenCtx *ctx)
{
    oracle_scene *s = vf->s;
    switch (e->tag) {
    case T_ARC: return vf_fix(vf, e->a, ctx);                         /* :29 */
    case T_X: case T_Y: case T_TAU: case T_E: case T_NAT: {            /* :30 */
        OExpr *n = new_expr(s, e->tag); n->u = e->u; return n;
    }
    case T_VAR: {                                                      /* :31-36 */
        OExpr *n = new_expr(s, T_VAR); n->u = e->u;
        for (size_t i = 0; i < ctx->n; i++)
            if (e->u == ctx->old_id[i]) { n->u = ctx->new_id[i]; break; }
        return n;
    }
    case T_LET: {                                                      /* :49-66 */
        OExpr *n = new_expr(s, T_LET);
        OCtx *c = (OCtx *)arena_alloc(s, sizeof(OCtx));
        uint64_t k = e->ctx->n;
        c->n = k;
        c->ids = (uint64_t *)arena_alloc(s, (size_t)(k ? k : 1) * 8);
        c->defs = (OExpr **)arena_alloc(s, (size_t)(k ? k : 1) * sizeof(OExpr *));
        RenCtx nc;
        nc.n = (size_t)k;
        nc.old_id = (uint64_t *)malloc((size_t)(k ? k : 1) * 8);
        nc.new_id = (uint64_t *)malloc((size_t)(k ? k : 1) * 8);
        for (uint64_t i = 0; i < k; i++) {
            OExpr *d = vf_fix(vf, e->ctx->defs[i], ctx);   /* definitions see the OUTER ctx (:52) */
            uint64_t id;
            if (!vf_get(vf, d, &id)) {                      /* :53-62 */
                id = vf->var_count++;
                vf_put(vf, d, id);
            }
            nc.old_id[i] = e->ctx->ids[i]; nc.new_id[i] = id;
            c->ids[i] = id; c->defs[i] = d;
        }
        n->ctx = c;
        n->a = vf_fix(vf, e->a, &nc);                       /* body sees only the new ctx (:65) */
        free(nc.old_id); free(nc.new_id);
        return n;
    }
    case T_DECOR: {                                                    /* :67 */
        OExpr *n = new_expr(s, T_DECOR);
        n->a = vf_fix(vf, e->a, ctx); n->toks = e->toks; n->ntoks = e->ntoks;
        return n;
    }
    case T_APP: {                                                      /* :68 */
        OExpr *n = new_expr(s, T_APP); n->app_id = e->app_id;
        n->a = vf_fix(vf, e->a, ctx); n->b = vf_fix(vf, e->b, ctx);
        return n;
    }
    default: {                                                         /* :37-48 */
        OExpr *n = new_expr(s, e->tag);
        n->a = vf_fix(vf, e->a, ctx);
        if (e->b) n->b = vf_fix(vf, e->b, ctx);
        return n;
    }
    }
}

/* var_fixer::fix_color (src/var_fixer.rs:74-82): one VarFixer for R, G, B. */
void oracle_fix_color(oracle_scene *s)
{
    if (s->fixed) return;
    VarFixer vf = { s, NULL, 0, 0, 0 };
    RenCtx empty = { NULL, NULL, 0 };
    for (int c = 0; c < 3; c++) s->color[c] = vf_fix(&vf, s->color[c], &empty);
    free(vf.tab);
    s->fixed = 1;
}

/* ---- scalar semantics --------------------------------------------------- */
/* Rust `f64 as u8` (src/render.rs:27-29,92-94): saturating, NaN -> 0. */
uint8_t oracle_cast_u8(double v)
{
    if (!(v > 0.0)) return 0;          /* NaN, negatives, +-0 */
    if (v >= 255.0) return 255;
    return (uint8_t)v;                 /* truncation toward zero */
}

/* f64::max / f64::min (src/lib.rs:655-658): NaN-ignoring.  For operands that
 * compare equal (+0 vs -0) Rust documents the choice as unspecified; this
 * oracle fixes IEEE 754-2019 maximumNumber/minimumNumber (-0 < +0). */
static inline double rs_max(double a, double b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return signbit(a) ? b : a;
    return a > b ? a : b;
}
static inline double rs_min(double a, double b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return signbit(a) ? a : b;
    return a < b ? a : b;
}

double oracle_op_unary(int tag, double a)
{
    switch (tag) {
    case T_NEG: return -a;                         /* src/lib.rs:640 */
    case T_ABS: return fabs(a);                    /* :641 */
    case T_RECIP: return 1.0 / a;                  /* :642 f64::recip */
    case T_SQRT: return sqrt(a);                   /* :643 */
    case T_STEP: return a >= 0.0 ? 1.0 : 0.0;      /* :644-647 */
    case T_SIN: return sin(a);                     /* :648 platform libm */
    case T_EXP: return exp(a);                     /* :649 */
    case T_LN: return log(a);                      /* :650 */
    }
    return NAN;
}

double oracle_op_binary(int tag, double a, double b)
{
    switch (tag) {
    case T_ADD: return a + b;                      /* :651-652 */
    case T_MUL: return a * b;                      /* :653-654 */
    case T_MAX: return rs_max(a, b);               /* :655-656 */
    case T_MIN: return rs_min(a, b);               /* :657-658 */
    }
    return NAN;
}

/* ---- Cache (src/cache.rs:6-42) ------------------------------------------
 * FnvHashMap<u64,(f64,bool)>.  Stored as two open-addressing tables with
 * generation stamps: x-dependent entries (dropped by clear_dep_x, :18-20) and
 * x-independent ones (dropped by clear / a fresh Cache per row).  Observable
 * behaviour is that of one map with `retain`. */
typedef struct { uint64_t key; double val; uint32_t gen; } CEnt;
typedef struct {
    CEnt *dep, *nodep;
    size_t mask;
    uint32_t gen_dep, gen_nodep;
} OCache;

static void cache_init(OCache *c, uint64_t nvars)
{
    size_t cap = 16;
    while (cap < 4 * nvars + 16) cap <<= 1;
    c->dep = (CEnt *)calloc(cap, sizeof(CEnt));
    c->nodep = (CEnt *)calloc(cap, sizeof(CEnt));
    c->mask = cap - 1;
    c->gen_dep = c->gen_nodep = 1;
}
static void cache_free(OCache *c) { free(c->dep); free(c->nodep); }
static void cache_bump(CEnt *t, size_t mask, uint32_t *gen)
{
    if (++*gen == 0) { memset(t, 0, (mask + 1) * sizeof(CEnt)); *gen = 1; }
}
static void cache_clear(OCache *c) { cache_bump(c->dep, c->mask, &c->gen_dep); cache_bump(c->nodep, c->mask, &c->gen_nodep); }
static void cache_clear_dep_x(OCache *c) { cache_bump(c->dep, c->mask, &c->gen_dep); }

static inline size_t cslot(uint64_t k, size_t mask) { return (size_t)((k * 0x9e3779b97f4a7c15ULL) >> 20) & mask; }
static int ctab_get(const CEnt *t, size_t mask, uint32_t gen, uint64_t k, double *v)
{
    size_t j = cslot(k, mask);
    while (t[j].gen == gen) {
        if (t[j].key == k) { *v = t[j].val; return 1; }
        j = (j + 1) & mask;
    }
    return 0;
}
static void ctab_put(CEnt *t, size_t mask, uint32_t gen, uint64_t k, double v)
{
    size_t j = cslot(k, mask);
    while (t[j].gen == gen && t[j].key != k) j = (j + 1) & mask;
    t[j].key = k; t[j].val = v; t[j].gen = gen;
}

/* ---- Runtime<Textures> (src/lib.rs:72-98, src/textures.rs) ------------- */
typedef struct {
    const oracle_texture *tex;
    uint32_t n_tex;
    int *fault;     /* set when the reference would panic (functions[id] OOB, src/lib.rs:665) */
} ORt;

/* Rust `f64 as u32`: saturating, NaN -> 0. */
static inline uint32_t cast_u32(double v)
{
    if (!(v > 0.0)) return 0;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}

/* textures::functions table (src/textures.rs:54-65): 5 entries per image =
 * fun_color_channel x3 (:27-36), fun_image_width (:40-43), fun_image_height (:47-50). */
static double rt_call(const ORt *rt, uint32_t id, double x, double y)
{
    if (id >= rt->n_tex * 5u) { *rt->fault = 1; return NAN; }
    const oracle_texture *im = &rt->tex[id / 5u];
    uint32_t sel = id % 5u;
    if (sel == 3) return (double)im->w;
    if (sel == 4) return (double)im->h;
    if (x < 0.0 || y < 0.0) return 0.0;                        /* :30 */
    uint32_t xi = cast_u32(x), yi = cast_u32(y);               /* :32-33 */
    if (xi >= im->w || yi >= im->h) return 0.0;                /* :34 */
    return (double)im->rgb[((size_t)yi * im->w + xi) * 3 + sel];   /* :35 */
}

/* ---- eval2 / dep_x / Cache::val ---------------------------------------- */
typedef struct { double v; int dep; } ValDep;
static const OCtx EMPTY_CTX = { 0, NULL, NULL };

static double eval2(const OExpr *e, const ORt *rt, double x, double y, const OCtx *ctx, OCache *c);
static int dep_x(const OExpr *e, const ORt *rt, double x, double y, const OCtx *ctx, OCache *c);

/* Cache::val (src/cache.rs:23-42) */
static ValDep cache_val(OCache *c, const ORt *rt, double x, double y, uint64_t name, const OCtx *ctx)
{
    ValDep r;
    if (ctab_get(c->dep, c->mask, c->gen_dep, name, &r.v)) { r.dep = 1; return r; }       /* :30 */
    if (ctab_get(c->nodep, c->mask, c->gen_nodep, name, &r.v)) { r.dep = 0; return r; }
    for (uint64_t i = 0; i < ctx->n; i++) {                                                /* :32 */
        if (ctx->ids[i] == name) {                                                         /* :33 */
            r.v = eval2(ctx->defs[i], rt, x, y, ctx, c);                                   /* :34 */
            r.dep = dep_x(ctx->defs[i], rt, x, y, ctx, c);                                 /* :35 */
            if (r.dep) ctab_put(c->dep, c->mask, c->gen_dep, name, r.v);                   /* :36 */
            else ctab_put(c->nodep, c->mask, c->gen_nodep, name, r.v);
            return r;
        }
    }
    r.v = NAN; r.dep = 0;                                                                  /* :40 */
    return r;
}

/* Expr::eval2 (src/lib.rs:623-670) */
static double eval2(const OExpr *e, const ORt *rt, double x, double y, const OCtx *ctx, OCache *c)
{
    switch (e->tag) {
    case T_ARC: return eval2(e->a, rt, x, y, ctx, c);                  /* :633 */
    case T_X: return x;                                                /* :634 */
    case T_Y: return y;                                                /* :635 */
    case T_TAU: return 6.283185307179586;                              /* :636 */
    case T_E: return 2.718281828459045;                                /* :637 */
    case T_VAR: return cache_val(c, rt, x, y, e->u, ctx).v;            /* :638 */
    case T_NAT: return (double)e->u;                                   /* :639 `n as f64` */
    case T_NEG: case T_ABS: case T_RECIP: case T_SQRT: case T_STEP:
    case T_SIN: case T_EXP: case T_LN:
        return oracle_op_unary(e->tag, eval2(e->a, rt, x, y, ctx, c)); /* :640-650 */
    case T_ADD: case T_MUL: case T_MAX: case T_MIN: {                  /* :651-658 */
        double a = eval2(e->a, rt, x, y, ctx, c);
        double b = eval2(e->b, rt, x, y, ctx, c);
        return oracle_op_binary(e->tag, a, b);
    }
    case T_LET: return eval2(e->a, rt, x, y, e->ctx, c);               /* :659-662 ctx REPLACED */
    case T_DECOR: return eval2(e->a, rt, x, y, ctx, c);                /* :663 */
    case T_APP: {                                                      /* :664-668 */
        if (e->app_id >= rt->n_tex * 5u) { *rt->fault = 1; return NAN; }   /* index panics before args */
        double a = eval2(e->a, rt, x, y, ctx, c);
        double b = eval2(e->b, rt, x, y, ctx, c);
        return rt_call(rt, e->app_id, a, b);
    }
    }
    return NAN;
}

/* Expr::dep_x (src/lib.rs:675-706) */
static int dep_x(const OExpr *e, const ORt *rt, double x, double y, const OCtx *ctx, OCache *c)
{
    switch (e->tag) {
    case T_ARC: return dep_x(e->a, rt, x, y, ctx, c);
    case T_X: return 1;
    case T_Y: case T_TAU: case T_E: case T_NAT: return 0;
    case T_VAR: return cache_val(c, rt, x, y, e->u, ctx).dep;
    case T_NEG: case T_ABS: case T_RECIP: case T_SQRT: case T_STEP:
    case T_SIN: case T_EXP: case T_LN:
        return dep_x(e->a, rt, x, y, ctx, c);
    case T_ADD: case T_MUL: case T_MAX: case T_MIN: case T_APP: {
        int a = dep_x(e->a, rt, x, y, ctx, c);
        int b = dep_x(e->b, rt, x, y, ctx, c);
        return a || b;
    }
    case T_LET: return dep_x(e->a, rt, x, y, e->ctx, c);
    case T_DECOR: return dep_x(e->a, rt, x, y, ctx, c);
    }
    return 0;
}

double oracle_eval2(const oracle_scene *s, int ch, double x, double y,
                    const oracle_texture *tex, uint32_t n_tex)
{
    int fault = 0;
    ORt rt = { tex, n_tex, &fault };
    OCache c;
    cache_init(&c, s->n_let_vars);
    double v = eval2(s->color[ch], &rt, x, y, &EMPTY_CTX, &c);
    cache_free(&c);
    return v;
}

/* ---- par_gen_to_image (src/render.rs:35-99) ---------------------------- */
typedef struct {
    oracle_scene *s;
    uint32_t w, y0, y1;
    ORt rt;
    uint8_t *rgb8;
    double *rgb64;
    volatile uint32_t *next_row;
} Job;

static void render_row(Job *j, uint32_t y, OCache *c)
{
    oracle_scene *s = j->s;
    cache_clear(c);                                   /* Cache::new() per row (:86) */
    size_t base = (size_t)(y - j->y0) * j->w * 3;
    for (uint32_t x = 0; x < j->w; x++) {             /* :88 */
        cache_clear_dep_x(c);                         /* :89 */
        double px = (double)x, py = (double)y;        /* :90 */
        for (int ch = 0; ch < 3; ch++) {              /* :91-93, R then G then B, shared cache */
            double v = eval2(s->color[ch], &j->rt, px, py, &EMPTY_CTX, c);
            if (j->rgb64) j->rgb64[base + (size_t)x * 3 + ch] = v;
            if (j->rgb8) j->rgb8[base + (size_t)x * 3 + ch] = oracle_cast_u8(v);
        }
    }
}

static void *worker(void *arg)
{
    Job *j = (Job *)arg;
    OCache c;
    cache_init(&c, j->s->n_let_vars);
    for (;;) {
        uint32_t y = __atomic_fetch_add(j->next_row, 1u, __ATOMIC_RELAXED);
        if (y >= j->y1) break;
        render_row(j, y, &c);
    }
    cache_free(&c);
    return NULL;
}

int oracle_render_rows(oracle_scene *s, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                       const oracle_texture *tex, uint32_t n_tex, int threads,
                       uint8_t *rgb8, double *rgb64)
{
    if (y0 > y1 || y1 > h) { set_err("row range out of bounds"); return -1; }
    oracle_fix_color(s);                               /* src/render.rs:51 */
    int fault = 0;
    volatile uint32_t next = y0;
    Job j = { s, w, y0, y1, { tex, n_tex, &fault }, rgb8, rgb64, &next };
    if (threads <= 1) worker(&j);
    else {
        if (threads > 256) threads = 256;
        pthread_t th[256];
        for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, worker, &j);
        for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    }
    if (fault) { set_err("App id out of range of Runtime.functions (reference panics, src/lib.rs:665)"); return -3; }
    return 0;
}


/* ---- "JIT" baseline: the scene as straight-line C --------------------------------------------
 * Stand-in for the reference's wasmer JIT (src/wasm.rs:89-158, src/render.rs:102-192), which
 * cannot be built here (no Rust, no wasmer).  Mirrors gen_expr: one un-shared function per
 * channel, Let variables become locals assigned in definition order (gen_vars :77-86), neg /
 * sqrt / add / mul / max / min are native ops (:41-42,:56-59; max/min with wasm's NaN-propagating
 * semantics), abs / recip / step / sin / exp / ln / app are out-of-line calls like the host imports
 * (:8-34,:40-47).  Rows are pulled from a shared counter by N threads (render.rs:150-183).
 * BASELINE ONLY (bench.py --cpu-jit): compiled by the system cc, never part of the product. */
typedef struct { char *p; size_t n, cap; } SB;
static void sb_put(SB *b, const char *s)
{
    size_t k = strlen(s);
    if (b->n + k + 1 > b->cap) { b->cap = (b->cap + k + 1) * 2; b->p = (char *)realloc(b->p, b->cap); }
    memcpy(b->p + b->n, s, k + 1); b->n += k;
}
static void sb_fmt_u64(SB *b, const char *pre, uint64_t v, const char *post)
{
    char t[64]; snprintf(t, sizeof t, "%s%llu%s", pre, (unsigned long long)v, post); sb_put(b, t);
}

static void emit_locals(SB *b, const OExpr *e)
{
    if (!e) return;
    if (e->tag == T_LET)
        for (uint64_t i = 0; i < e->ctx->n; i++) { sb_fmt_u64(b, "    double v", e->ctx->ids[i], " = 0;\n"); emit_locals(b, e->ctx->defs[i]); }
    emit_locals(b, e->a); emit_locals(b, e->b);
}

static void emit_expr(SB *b, const OExpr *e)
{
    static const char *un[] = {"-", "imp_abs", "imp_recip", "__builtin_sqrt", "imp_step", "imp_sin", "imp_exp", "imp_ln"};
    switch (e->tag) {
    case T_ARC: case T_DECOR: emit_expr(b, e->a); break;
    case T_X: sb_put(b, "x"); break;
    case T_Y: sb_put(b, "y"); break;
    case T_TAU: sb_put(b, "6.283185307179586"); break;
    case T_E: sb_put(b, "2.718281828459045"); break;
    case T_VAR: sb_fmt_u64(b, "v", e->u, ""); break;
    case T_NAT: sb_fmt_u64(b, "", e->u, ".0"); break;
    case T_NEG: case T_ABS: case T_RECIP: case T_SQRT: case T_STEP: case T_SIN: case T_EXP: case T_LN:
        sb_put(b, un[e->tag - T_NEG]); sb_put(b, "("); emit_expr(b, e->a); sb_put(b, ")"); break;
    case T_ADD: case T_MUL:
        sb_put(b, "("); emit_expr(b, e->a); sb_put(b, e->tag == T_ADD ? " + " : " * "); emit_expr(b, e->b); sb_put(b, ")"); break;
    case T_MAX: case T_MIN:
        sb_put(b, e->tag == T_MAX ? "wasm_max(" : "wasm_min("); emit_expr(b, e->a); sb_put(b, ", "); emit_expr(b, e->b); sb_put(b, ")"); break;
    case T_LET:   /* GNU statement expression: definitions in order, then the body */
        sb_put(b, "({ ");
        for (uint64_t i = 0; i < e->ctx->n; i++) { sb_fmt_u64(b, "v", e->ctx->ids[i], " = "); emit_expr(b, e->ctx->defs[i]); sb_put(b, ";\n      "); }
        emit_expr(b, e->a); sb_put(b, "; })");
        break;
    case T_APP: sb_fmt_u64(b, "imp_app(", e->app_id, ", "); emit_expr(b, e->a); sb_put(b, ", "); emit_expr(b, e->b); sb_put(b, ")"); break;
    }
}

/* Returns a malloc'd C translation unit (free with oracle_free).  fix_color is applied first. */
char *oracle_scene_emit_c(oracle_scene *s)
{
    oracle_fix_color(s);
    SB b = {NULL, 0, 0};
    sb_put(&b,
        "#include <math.h>\n#include <pthread.h>\n#include <stdint.h>\n"
        "#define NI __attribute__((noinline)) static\n"
        "NI double imp_abs(double v) { return fabs(v); }\nNI double imp_sin(double v) { return sin(v); }\n"
        "NI double imp_exp(double v) { return exp(v); }\nNI double imp_ln(double v) { return log(v); }\n"
        "NI double imp_recip(double v) { return 1.0 / v; }\nNI double imp_step(double v) { return v >= 0.0 ? 1.0 : 0.0; }\n"
        "NI double imp_app(uint32_t id, double a, double b) { (void)id; (void)a; (void)b; return 0.0; }\n"
        "static inline double wasm_max(double a, double b) { if (a != a || b != b) return NAN; if (a == b) return signbit(a) ? b : a; return a > b ? a : b; }\n"
        "static inline double wasm_min(double a, double b) { if (a != a || b != b) return NAN; if (a == b) return signbit(a) ? a : b; return a < b ? a : b; }\n"
        "static inline uint8_t cast_u8(double v) { if (!(v > 0.0)) return 0; if (v >= 255.0) return 255; return (uint8_t)v; }\n");
    for (int c = 0; c < 3; c++) {
        char t[64]; snprintf(t, sizeof t, "NI double ch%d(double x, double y)\n{\n", c); sb_put(&b, t);
        emit_locals(&b, s->color[c]);
        sb_put(&b, "    return "); emit_expr(&b, s->color[c]); sb_put(&b, ";\n}\n");
    }
    sb_put(&b,
        "typedef struct { uint32_t w, y0, y1; uint8_t *rgb8; volatile uint32_t *next; } Job;\n"
        "static void *worker(void *arg) {\n    Job *j = (Job *)arg;\n    for (;;) {\n"
        "        uint32_t y = __atomic_fetch_add(j->next, 1u, __ATOMIC_RELAXED);\n        if (y >= j->y1) break;\n"
        "        uint8_t *row = j->rgb8 + (size_t)(y - j->y0) * j->w * 3;\n"
        "        for (uint32_t x = 0; x < j->w; x++) {\n"
        "            row[x * 3] = cast_u8(ch0((double)x, (double)y));\n            row[x * 3 + 1] = cast_u8(ch1((double)x, (double)y));\n"
        "            row[x * 3 + 2] = cast_u8(ch2((double)x, (double)y));\n        }\n    }\n    return 0;\n}\n"
        "int jit_render_rows(uint32_t w, uint32_t y0, uint32_t y1, int threads, uint8_t *rgb8) {\n"
        "    volatile uint32_t next = y0;\n    Job j = { w, y0, y1, rgb8, &next };\n    pthread_t th[256];\n"
        "    if (threads < 1) threads = 1;\n    if (threads > 256) threads = 256;\n"
        "    for (int i = 0; i < threads; i++) pthread_create(&th[i], 0, worker, &j);\n"
        "    for (int i = 0; i < threads; i++) pthread_join(th[i], 0);\n    return 0;\n}\n");
    return b.p;
}

void oracle_free(void *p) { free(p); }
