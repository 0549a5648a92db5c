/*
 * synth.c -- deterministic generator of shape-faithful synthetic dictionaries and
 * Japanese-like sentence batches.  Real ipadic / unidic
 * system.dic files are not available offline, so BASELINE.json's configs are
 * realised with these.  Neutral test/bench infrastructure: feeds byte-identical
 * inputs to the CPU oracle and to the HIP product.
 *
 * All randomness is xoshiro256** seeded through splitmix64 from one 64-bit seed.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SYN_API __attribute__((visibility("default")))

typedef struct { uint64_t s[4]; } rng_t;

static uint64_t splitmix64(uint64_t *x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static void rng_seed(rng_t *r, uint64_t seed) { for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&seed); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t rng_next(rng_t *r) {
    uint64_t *s = r->s, result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}
static double rng_u01(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static uint32_t rng_below(rng_t *r, uint32_t n) { return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32); }
static double rng_normal(rng_t *r) {
    double u1 = rng_u01(r), u2 = rng_u01(r);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
static uint32_t rng_poisson(rng_t *r, double lam) {
    double L = exp(-lam), p = 1.0; uint32_t k = 0;
    do { k++; p *= rng_u01(r); } while (p > L);
    return k - 1;
}

/* Zipf(s) sampler over {0..n-1} by inverse CDF */
typedef struct { double *cdf; uint32_t n; } zipf_t;
static void zipf_init(zipf_t *z, uint32_t n, double s) {
    z->n = n; z->cdf = (double *)malloc(sizeof(double) * n);
    double acc = 0;
    for (uint32_t i = 0; i < n; i++) { acc += 1.0 / pow((double)(i + 1), s); z->cdf[i] = acc; }
    for (uint32_t i = 0; i < n; i++) z->cdf[i] /= acc;
}
static uint32_t zipf_draw(const zipf_t *z, rng_t *r) {
    double u = rng_u01(r);
    uint32_t lo = 0, hi = z->n - 1;
    while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (z->cdf[mid] < u) lo = mid + 1; else hi = mid; }
    return lo;
}

typedef struct { char *p; size_t len, cap; } buf_t;
static void buf_reserve(buf_t *b, size_t n) {
    if (b->len + n > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 4096;
        while (nc < b->len + n) nc *= 2;
        b->p = (char *)realloc(b->p, nc); b->cap = nc;
    }
}
static void buf_put(buf_t *b, const void *s, size_t n) { buf_reserve(b, n); memcpy(b->p + b->len, s, n); b->len += n; }
static void buf_cp(buf_t *b, uint32_t cp) {
    char t[4]; int n;
    if (cp < 0x80) { t[0] = (char)cp; n = 1; }
    else if (cp < 0x800) { t[0] = (char)(0xC0 | (cp >> 6)); t[1] = (char)(0x80 | (cp & 63)); n = 2; }
    else if (cp < 0x10000) { t[0] = (char)(0xE0 | (cp >> 12)); t[1] = (char)(0x80 | ((cp >> 6) & 63)); t[2] = (char)(0x80 | (cp & 63)); n = 3; }
    else { t[0] = (char)(0xF0 | (cp >> 18)); t[1] = (char)(0x80 | ((cp >> 12) & 63)); t[2] = (char)(0x80 | ((cp >> 6) & 63)); t[3] = (char)(0x80 | (cp & 63)); n = 4; }
    buf_put(b, t, n);
}

typedef struct syn_dict {
    uint32_t n_words, num_right, num_left;
    buf_t lex;            /* lex.csv text */
    buf_t unk;            /* unk.def text */
    buf_t surf;           /* concatenated surfaces (utf-8) */
    uint32_t *surf_off;   /* n_words+1 */
    uint16_t *surf_chars; /* chars per surface */
    uint64_t seed;
} syn_dict;

/* Knobs of the lexicon law.  The default law is SURVEY.md 8(d)'s; the "dense" law raises the duplicate-surface rate,
 * shortens the words and shrinks the kanji pool so that the lattices get as dense as SURVEY 8(a) estimates for
 * the real unidic (>= 12 nodes and >= 80 connection pairs per character). */
typedef struct { double dup_p, len_lambda, kanji_zipf; uint32_t n_kanji; } syn_law;
static const syn_law SYN_LAW_DEFAULT = {0.15, 2.2, 0.8, 3000};

#define N_KANJI 3000
#define HIRA_LO 0x3041
#define HIRA_N 83 /* U+3041..U+3093 */
#define KATA_LO 0x30A1
#define KATA_N 86 /* U+30A1..U+30F6 */
static const char ALNUM[] = "abcdefghijklmnopqrstuvwxyz0123456789";

/* char.def for the synthetic dictionaries: the ipadic category set (same shape as
 * the reference fixture vibrato/src/tests/resources/char.def, re-typed here). */
static const char SYN_CHAR_DEF[] =
    "DEFAULT 0 1 0\nSPACE 0 1 0\nKANJI 0 0 2\nSYMBOL 1 1 0\nNUMERIC 1 1 0\nALPHA 1 1 0\n"
    "HIRAGANA 0 1 2\nKATAKANA 1 1 2\nKANJINUMERIC 1 1 0\nGREEK 1 1 0\nCYRILLIC 1 1 0\n"
    "0x0020 SPACE\n0x3000 SPACE\n0x0021..0x002F SYMBOL\n0x003A..0x0040 SYMBOL\n0x3001..0x3003 SYMBOL\n"
    "0x0030..0x0039 NUMERIC\n0xFF10..0xFF19 NUMERIC\n0x0041..0x005A ALPHA\n0x0061..0x007A ALPHA\n"
    "0xFF21..0xFF3A ALPHA\n0xFF41..0xFF5A ALPHA\n0x3041..0x309F HIRAGANA\n0x30A1..0x30FF KATAKANA\n"
    "0x30FC KATAKANA HIRAGANA\n0x3400..0x4DBF KANJI\n0x4E00..0x9FFF KANJI\n0xF900..0xFAFF KANJI\n"
    "0x4E00 KANJINUMERIC KANJI\n0x4E8C KANJINUMERIC KANJI\n0x4E09 KANJINUMERIC KANJI\n0x56DB KANJINUMERIC KANJI\n"
    "0x0391..0x03C9 GREEK\n0x0410..0x044F CYRILLIC\n";

SYN_API const char *syn_char_def(size_t *len) { *len = sizeof(SYN_CHAR_DEF) - 1; return SYN_CHAR_DEF; }

static const char *CATS[11] = {"DEFAULT", "SPACE", "KANJI", "SYMBOL", "NUMERIC", "ALPHA", "HIRAGANA", "KATAKANA", "KANJINUMERIC", "GREEK", "CYRILLIC"};
static const int UNK_ROWS[11] = {6, 1, 6, 5, 3, 4, 4, 6, 2, 2, 1}; /* 40 rows */

SYN_API void syn_dict_free(syn_dict *d) {
    if (!d) return;
    free(d->lex.p); free(d->unk.p); free(d->surf.p); free(d->surf_off); free(d->surf_chars); free(d);
}

/* connection cost of the synthetic matrix: i16(hash(l, r) mod 16001 - 8000) */
static inline int16_t syn_cost(uint32_t left, uint32_t right, uint64_t seed) {
    uint64_t x = ((uint64_t)left << 32) | right;
    x ^= seed;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return (int16_t)((int)(x % 16001) - 8000);
}

/* fills data[left*num_right + right] */
SYN_API void syn_fill_matrix(const syn_dict *d, int16_t *out) {
    for (uint32_t l = 0; l < d->num_left; l++) {
        int16_t *row = out + (size_t)l * d->num_right;
        for (uint32_t r = 0; r < d->num_right; r++) row[r] = syn_cost(l, r, d->seed);
    }
}


/* open-addressing set of generated surfaces (uniqueness apart from the explicit multi-POS duplicates) */
typedef struct { uint32_t *slot; uint32_t cap; } sset_t;
static uint64_t hash_bytes(const char *p, uint32_t n) {
    uint64_t h = 1469598103934665603ULL;
    for (uint32_t i = 0; i < n; i++) { h ^= (uint8_t)p[i]; h *= 1099511628211ULL; }
    return h;
}
/* returns 1 if inserted (was absent); word w's bytes are surf.p[off[w]..off[w]+len) */
static int sset_insert(sset_t *s, const syn_dict *d, uint32_t w, uint32_t len) {
    const char *p = d->surf.p + d->surf_off[w];
    uint32_t i = (uint32_t)(hash_bytes(p, len) & (s->cap - 1));
    for (;; i = (i + 1) & (s->cap - 1)) {
        uint32_t o = s->slot[i];
        if (o == 0xFFFFFFFFu) { s->slot[i] = w; return 1; }
        uint32_t ol = d->surf_off[o + 1] - d->surf_off[o];
        if (ol == len && memcmp(d->surf.p + d->surf_off[o], p, len) == 0) return 0;
    }
}

static syn_dict *syn_dict_build(uint32_t n_words, uint32_t num_right, uint32_t num_left, uint64_t seed, syn_law law) {
    syn_dict *d = (syn_dict *)calloc(1, sizeof(*d));
    d->n_words = n_words; d->num_right = num_right; d->num_left = num_left; d->seed = seed;
    d->surf_off = (uint32_t *)malloc(sizeof(uint32_t) * (n_words + 1));
    d->surf_chars = (uint16_t *)malloc(sizeof(uint16_t) * n_words);
    rng_t r; rng_seed(&r, seed);
    zipf_t zk, zl, zr;
    zipf_init(&zk, law.n_kanji, law.kanji_zipf);
    uint32_t nl = num_left > 1 ? num_left - 1 : 1, nr = num_right > 1 ? num_right - 1 : 1;
    zipf_init(&zl, nl, 1.1);
    zipf_init(&zr, nr, 1.1);
    /* kanji pool: a fixed pseudo-random subset of U+4E00..U+9FA5 */
    uint32_t pool[N_KANJI];
    {
        rng_t pr; rng_seed(&pr, 0xC0FFEEULL);
        uint8_t *seen = (uint8_t *)calloc(0x51A6, 1);
        for (int i = 0; i < N_KANJI;) {
            uint32_t c = rng_below(&pr, 0x51A6);
            if (!seen[c]) { seen[c] = 1; pool[i++] = 0x4E00 + c; }
        }
        free(seen);
    }
    uint32_t w = 0;
    buf_t tmp = {0};
    sset_t ss;
    ss.cap = 1024;
    while (ss.cap < n_words * 2u) ss.cap <<= 1;
    ss.slot = (uint32_t *)malloc(sizeof(uint32_t) * ss.cap);
    memset(ss.slot, 0xFF, sizeof(uint32_t) * ss.cap);
    /* all single kana first ("all single kana included") */
    for (int k = 0; k < HIRA_N + KATA_N && w < n_words; k++, w++) {
        d->surf_off[w] = (uint32_t)d->surf.len;
        buf_cp(&d->surf, k < HIRA_N ? HIRA_LO + k : KATA_LO + (k - HIRA_N));
        d->surf_chars[w] = 1;
        d->surf_off[w + 1] = (uint32_t)d->surf.len;
        sset_insert(&ss, d, w, d->surf_off[w + 1] - d->surf_off[w]);
    }
    for (; w < n_words; w++) {
        d->surf_off[w] = (uint32_t)d->surf.len;
        if (w > 1000 && rng_u01(&r) < law.dup_p) { /* duplicate surface (multi-POS) */
            uint32_t src = rng_below(&r, w);
            const uint32_t sl = d->surf_off[src + 1] - d->surf_off[src];
            buf_reserve(&d->surf, sl);  /* (the source lies in the same buffer: grow first, then take the pointer) */
            buf_put(&d->surf, d->surf.p + d->surf_off[src], sl);
            d->surf_chars[w] = d->surf_chars[src];
            d->surf_off[w + 1] = (uint32_t)d->surf.len;
            continue;
        }
        for (int attempt = 0;; attempt++) {
            d->surf.len = d->surf_off[w];
            uint32_t len = 1 + rng_poisson(&r, law.len_lambda) + (uint32_t)(attempt / 4);
            if (len > 12) len = 12;
            double u = rng_u01(&r);
            uint32_t nch = 0;
            if (u < 0.55) { /* kanji, optionally with a hiragana tail (okurigana) */
                uint32_t tail = (len >= 2 && rng_u01(&r) < 0.3) ? 1 + rng_below(&r, len >= 3 ? 2 : 1) : 0;
                for (uint32_t i = 0; i < len - tail; i++, nch++) buf_cp(&d->surf, pool[zipf_draw(&zk, &r)]);
                for (uint32_t i = 0; i < tail; i++, nch++) buf_cp(&d->surf, HIRA_LO + rng_below(&r, HIRA_N));
            } else if (u < 0.80) {
                for (uint32_t i = 0; i < len; i++, nch++) buf_cp(&d->surf, HIRA_LO + rng_below(&r, HIRA_N));
            } else if (u < 0.95) {
                if (len < 2) len = 2;
                for (uint32_t i = 0; i < len; i++, nch++) buf_cp(&d->surf, KATA_LO + rng_below(&r, KATA_N));
            } else {
                if (len < 2) len = 2;
                for (uint32_t i = 0; i < len; i++, nch++) buf_cp(&d->surf, (uint32_t)ALNUM[rng_below(&r, 36)]);
            }
            d->surf_chars[w] = (uint16_t)nch;
            d->surf_off[w + 1] = (uint32_t)d->surf.len;
            if (sset_insert(&ss, d, w, d->surf_off[w + 1] - d->surf_off[w])) break;
        }
    }
    free(ss.slot);
    d->surf_off[n_words] = (uint32_t)d->surf.len;
    /* lex.csv */
    char line[256];
    for (w = 0; w < n_words; w++) {
        uint32_t left = 1 + zipf_draw(&zl, &r), right = 1 + zipf_draw(&zr, &r);
        if (left >= num_left) left = num_left - 1;
        if (right >= num_right) right = num_right - 1;
        double c = 6000.0 + 3000.0 * rng_normal(&r);
        if (c < -2000) c = -2000; if (c > 20000) c = 20000;
        const char *s = d->surf.p + d->surf_off[w];
        uint32_t sl = d->surf_off[w + 1] - d->surf_off[w];
        buf_put(&d->lex, s, sl);
        int n = snprintf(line, sizeof(line), ",%u,%u,%d,POS%u,sub%u,*,*,*,*,", left, right, (int)c, left % 13, right % 7);
        buf_put(&d->lex, line, (size_t)n);
        buf_put(&d->lex, s, sl);
        n = snprintf(line, sizeof(line), ",w%u\n", w);
        buf_put(&d->lex, line, (size_t)n);
    }
    /* unk.def: 1-6 rows per category, 40 total */
    for (int c = 0; c < 11; c++) {
        for (int k = 0; k < UNK_ROWS[c]; k++) {
            uint32_t left = 1 + zipf_draw(&zl, &r), right = 1 + zipf_draw(&zr, &r);
            if (left >= num_left) left = num_left - 1;
            if (right >= num_right) right = num_right - 1;
            int cost = 3000 + (int)rng_below(&r, 12000);
            int n = snprintf(line, sizeof(line), "%s,%u,%u,%d,UNK-%s,%d,*,*,*,*\n", CATS[c], left, right, cost, CATS[c], k);
            buf_put(&d->unk, line, (size_t)n);
        }
    }
    free(zk.cdf); free(zl.cdf); free(zr.cdf); free(tmp.p);
    return d;
}

SYN_API syn_dict *syn_dict_new(uint32_t n_words, uint32_t num_right, uint32_t num_left, uint64_t seed) {
    return syn_dict_build(n_words, num_right, num_left, seed, SYN_LAW_DEFAULT);
}
SYN_API syn_dict *syn_dict_new_law(uint32_t n_words, uint32_t num_right, uint32_t num_left, uint64_t seed, double dup_p,
                                   double len_lambda, double kanji_zipf, uint32_t n_kanji) {
    syn_law law = {dup_p, len_lambda, kanji_zipf, n_kanji < 16 ? 16 : n_kanji > N_KANJI ? N_KANJI : n_kanji};
    return syn_dict_build(n_words, num_right, num_left, seed, law);
}

SYN_API const char *syn_dict_lex(const syn_dict *d, size_t *len) { *len = d->lex.len; return d->lex.p; }
SYN_API const char *syn_dict_unk(const syn_dict *d, size_t *len) { *len = d->unk.len; return d->unk.p; }

/* user.csv of n compounds: 2-3 lexicon words concatenated, cost U[-1000,0] (cfg5) */
SYN_API char *syn_user_csv(const syn_dict *d, uint32_t n, uint64_t seed, size_t *len) {
    rng_t r; rng_seed(&r, seed ^ 0x5EEDULL);
    zipf_t zw; zipf_init(&zw, d->n_words, 1.0);
    buf_t b = {0};
    char line[128];
    for (uint32_t i = 0; i < n; i++) {
        uint32_t k = 2 + rng_below(&r, 2);
        for (uint32_t j = 0; j < k; j++) {
            uint32_t w = zipf_draw(&zw, &r);
            buf_put(&b, d->surf.p + d->surf_off[w], d->surf_off[w + 1] - d->surf_off[w]);
        }
        uint32_t left = 1 + rng_below(&r, d->num_left - 1), right = 1 + rng_below(&r, d->num_right - 1);
        int m = snprintf(line, sizeof(line), ",%u,%u,%d,USER,compound,%u\n", left, right, -(int)rng_below(&r, 1001), i);
        buf_put(&b, line, (size_t)m);
    }
    free(zw.cdf);
    *len = b.len;
    return b.p;
}

/* length laws */
enum { SYN_LEN_UNIFORM_5_20 = 0, SYN_LEN_LOGNORMAL_40 = 1, SYN_LEN_MIXED = 2 };

static uint32_t draw_len(rng_t *r, int law) {
    double v;
    switch (law) {
    case SYN_LEN_UNIFORM_5_20: return 5 + rng_below(r, 16);
    case SYN_LEN_LOGNORMAL_40:
        v = exp(log(40.0) + 0.6 * rng_normal(r));
        if (v < 1) v = 1; if (v > 400) v = 400;
        return (uint32_t)v;
    default: {
        double u = rng_u01(r);
        if (u < 0.70) { v = exp(log(30.0) + 0.6 * rng_normal(r)); if (v < 1) v = 1; if (v > 400) v = 400; return (uint32_t)v; }
        if (u < 0.95) { v = exp(log(120.0) + 0.6 * rng_normal(r)); if (v < 1) v = 1; if (v > 400) v = 400; return (uint32_t)v; }
        return 500 + rng_below(r, 1501);
    }
    }
}

/* Generates n sentences. Returns malloc'ed utf-8 text (no separators); offsets[n+1]
 * (caller-allocated) receives byte offsets. space_p > 0 injects space runs (cfg5). */
SYN_API char *syn_sentences(const syn_dict *d, uint64_t n, uint64_t seed, int len_law, double space_p,
                            uint64_t *offsets, size_t *out_len) {
    rng_t r; rng_seed(&r, seed ^ 0x53454E54ULL);
    zipf_t zw; zipf_init(&zw, d->n_words, 1.0);
    buf_t b = {0};
    for (uint64_t s = 0; s < n; s++) {
        offsets[s] = b.len;
        uint32_t target = draw_len(&r, len_law), nch = 0;
        if (space_p > 0 && rng_u01(&r) < space_p) { uint32_t k = 1 + rng_below(&r, 3); for (uint32_t i = 0; i < k; i++) buf_cp(&b, 0x20); nch += k; }
        while (nch + 1 < target) {
            if (rng_u01(&r) < 0.05) { /* unknown run */
                double u = rng_u01(&r);
                if (u < 0.4) { uint32_t k = 3 + rng_below(&r, 6); for (uint32_t i = 0; i < k; i++) buf_cp(&b, KATA_LO + rng_below(&r, KATA_N)); nch += k; }
                else if (u < 0.75) { uint32_t k = 2 + rng_below(&r, 9); for (uint32_t i = 0; i < k; i++) buf_cp(&b, (uint32_t)('a' + rng_below(&r, 26))); nch += k; }
                else { uint32_t k = 1 + rng_below(&r, 6); for (uint32_t i = 0; i < k; i++) buf_cp(&b, (uint32_t)('0' + rng_below(&r, 10))); nch += k; }
            } else {
                uint32_t w = zipf_draw(&zw, &r);
                /* spread the head of the Zipf over the id space: word ids are in generation order */
                w = (uint32_t)(((uint64_t)w * 2654435761ULL) % d->n_words);
                buf_put(&b, d->surf.p + d->surf_off[w], d->surf_off[w + 1] - d->surf_off[w]);
                nch += d->surf_chars[w];
            }
            if (space_p > 0 && rng_u01(&r) < space_p) { uint32_t k = 1 + rng_below(&r, 3); for (uint32_t i = 0; i < k; i++) buf_cp(&b, 0x20); nch += k; }
        }
        buf_cp(&b, 0x3002); /* 。 */
        if (space_p > 0 && rng_u01(&r) < space_p * 0.5) { uint32_t k = 1 + rng_below(&r, 3); for (uint32_t i = 0; i < k; i++) buf_cp(&b, 0x20); }
    }
    offsets[n] = b.len;
    free(zw.cdf);
    *out_len = b.len;
    return b.p;
}

SYN_API void syn_free(void *p) { free(p); }
