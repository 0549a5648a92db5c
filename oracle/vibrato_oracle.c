/*
 * vibrato_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C, single-threaded restatement of the tokenize() hot path of
 * daac-tools/vibrato 0.5.2 (reference tree at /root/reference, never read at
 * run time).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (vibrato_amd/, libvibrato_hip.so)
 * never links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 * every known-answer vector the reference's own tests hold for the path
 * (vibrato/src/tests/tokenizer.rs, tokenizer.rs:208-361, tests/lexicon.rs,
 * lexicon.rs:232-272, tests/connector.rs, matrix_connector.rs:131-183,
 * character.rs:288-298), transcribed to tests/golden/ by
 * tests/golden/make_golden.py.
 *
 * Third-party algorithm not under /root/reference: crawdad 0.3.0 (double-array
 * trie, vibrato/Cargo.toml:21).  Its published layout (code mapper ordered by
 * character frequency, nodes {base,check}, leaf values hung on END_CODE=0) is
 * restated in the "double-array trie" section; results depend only on the
 * enumeration contract pinned by lexicon.rs:232-272 / tests/lexicon.rs:8-57.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/vibrato/src unless stated otherwise).
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <stdarg.h>
#include <limits.h>

#define ORA_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ utils */

static void set_err(char *err, size_t cap, const char *fmt, ...) {
    if (!err || cap == 0) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, cap, fmt, ap);
    va_end(ap);
}

typedef struct { char *p; size_t len, cap; } bytebuf;

static void bb_push(bytebuf *b, const void *src, size_t n) {
    if (b->len + n > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 256;
        while (nc < b->len + n) nc *= 2;
        b->p = (char *)realloc(b->p, nc);
        b->cap = nc;
    }
    memcpy(b->p + b->len, src, n);
    b->len += n;
}

/* Decode one UTF-8 scalar. Returns byte length (1..4) or 0 when invalid.
 * Rust's &str guarantees validity (sentence.rs:40-46 iterates char_indices). */
static int utf8_decode(const uint8_t *s, size_t n, uint32_t *cp) {
    if (n == 0) return 0;
    uint8_t b0 = s[0];
    if (b0 < 0x80) { *cp = b0; return 1; }
    if (b0 < 0xC2) return 0;
    if (b0 < 0xE0) {
        if (n < 2 || (s[1] & 0xC0) != 0x80) return 0;
        *cp = ((uint32_t)(b0 & 0x1F) << 6) | (s[1] & 0x3F);
        return 2;
    }
    if (b0 < 0xF0) {
        if (n < 3 || (s[1] & 0xC0) != 0x80 || (s[2] & 0xC0) != 0x80) return 0;
        uint32_t c = ((uint32_t)(b0 & 0x0F) << 12) | ((uint32_t)(s[1] & 0x3F) << 6) | (s[2] & 0x3F);
        if (c < 0x800 || (c >= 0xD800 && c <= 0xDFFF)) return 0;
        *cp = c;
        return 3;
    }
    if (b0 < 0xF5) {
        if (n < 4 || (s[1] & 0xC0) != 0x80 || (s[2] & 0xC0) != 0x80 || (s[3] & 0xC0) != 0x80) return 0;
        uint32_t c = ((uint32_t)(b0 & 0x07) << 18) | ((uint32_t)(s[1] & 0x3F) << 12) |
                     ((uint32_t)(s[2] & 0x3F) << 6) | (s[3] & 0x3F);
        if (c < 0x10000 || c > 0x10FFFF) return 0;
        *cp = c;
        return 4;
    }
    return 0;
}

/* Rust `str::parse::<iN/uN>()`: optional '+' (and '-' for signed), digits only. */
static int parse_int(const char *s, size_t n, int allow_neg, long long lo, long long hi, long long *out) {
    size_t i = 0;
    int neg = 0;
    if (n == 0) return 0;
    if (s[0] == '+') i = 1;
    else if (s[0] == '-') { if (!allow_neg) return 0; neg = 1; i = 1; }
    if (i >= n) return 0;
    long long v = 0;
    for (; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return 0;
        v = v * 10 + (s[i] - '0');
        if (v > (1LL << 40)) return 0;
    }
    if (neg) v = -v;
    if (v < lo || v > hi) return 0;
    *out = v;
    return 1;
}

#include "raw_connector.c"
#include "dual_connector.c"

/* ------------------------------------------------- lexicon CSV (lexicon.rs) */

typedef struct {
    char *surface;      /* unquoted field 0, NUL terminated */
    uint32_t surface_len;
    uint16_t left_id, right_id;
    int16_t word_cost;
    const char *feature; /* raw tail of the record, points into the caller's buffer */
    uint32_t feature_len;
} raw_entry;

typedef struct { raw_entry *v; size_t n, cap; } entry_vec;

static void entry_vec_free(entry_vec *ev) {
    for (size_t i = 0; i < ev->n; i++) free(ev->v[i].surface);
    free(ev->v);
    ev->v = NULL; ev->n = ev->cap = 0;
}

/* Lexicon::parse_csv, lexicon.rs:111-200.  csv_core defaults: ',' delimiter,
 * '"' quoting with doubled quotes, record terminator = any of \r, \n, \r\n,
 * blank lines skipped.  Fields 0-3 are unquoted values; the feature is the RAW
 * byte tail after the 4th delimiter up to (not including) the terminator
 * (lexicon.rs:157-159,177).  Rows with < 5 fields are an error (l.170-176);
 * rows with an empty surface are skipped and take no word id (l.178-183). */
static int parse_lex_csv(const char *buf, size_t len, const char *name, entry_vec *out, char *err, size_t errcap) {
    size_t pos = 0;
    bytebuf field = {0};
    memset(out, 0, sizeof(*out));
    while (pos < len) {
        /* StartRecord: skip terminator bytes (blank lines) */
        if (buf[pos] == '\n' || buf[pos] == '\r') { pos++; continue; }
        size_t rec_start = pos;
        int field_cnt = 0;
        raw_entry e;
        memset(&e, 0, sizeof(e));
        size_t feat_start = 0;
        int rec_done = 0;
        while (!rec_done) {
            /* read one field */
            field.len = 0;
            int at_end = 0; /* 1: delimiter, 2: terminator/eof */
            if (pos < len && buf[pos] == '"') {
                pos++;
                for (;;) {
                    if (pos >= len) { at_end = 2; break; }
                    char c = buf[pos];
                    if (c == '"') {
                        if (pos + 1 < len && buf[pos + 1] == '"') { bb_push(&field, "\"", 1); pos += 2; continue; }
                        pos++;
                        break; /* closing quote; rest handled as unquoted */
                    }
                    bb_push(&field, &c, 1);
                    pos++;
                }
            }
            while (!at_end) {
                if (pos >= len) { at_end = 2; break; }
                char c = buf[pos];
                if (c == ',') { pos++; at_end = 1; break; }
                if (c == '\n' || c == '\r') { at_end = 2; break; }
                bb_push(&field, &c, 1);
                pos++;
            }
            long long v;
            switch (field_cnt) {
            case 0:
                e.surface = (char *)malloc(field.len + 1);
                memcpy(e.surface, field.p, field.len);
                e.surface[field.len] = 0;
                e.surface_len = (uint32_t)field.len;
                break;
            case 1:
                if (!parse_int(field.p, field.len, 0, 0, 65535, &v)) goto bad_int;
                e.left_id = (uint16_t)v;
                break;
            case 2:
                if (!parse_int(field.p, field.len, 0, 0, 65535, &v)) goto bad_int;
                e.right_id = (uint16_t)v;
                break;
            case 3:
                if (!parse_int(field.p, field.len, 1, -32768, 32767, &v)) goto bad_int;
                e.word_cost = (int16_t)v;
                feat_start = pos;
                break;
            default: break;
            }
            if (at_end == 2) {
                size_t rec_end = pos;
                /* consume the terminator: \r\n, \r or \n */
                if (pos < len && buf[pos] == '\r') { pos++; if (pos < len && buf[pos] == '\n') pos++; }
                else if (pos < len && buf[pos] == '\n') pos++;
                if (field_cnt <= 3) {
                    set_err(err, errcap, "%s: A csv row of lexicon must have five items at least, \"%.*s\"",
                            name, (int)(rec_end - rec_start), buf + rec_start);
                    free(e.surface);
                    goto fail;
                }
                e.feature = buf + feat_start;
                e.feature_len = (uint32_t)(rec_end - feat_start);
                /* validate utf8 of surface */
                {
                    size_t i = 0; uint32_t cp;
                    while (i < e.surface_len) {
                        int k = utf8_decode((const uint8_t *)e.surface + i, e.surface_len - i, &cp);
                        if (!k) { set_err(err, errcap, "%s: invalid utf-8", name); free(e.surface); goto fail; }
                        i += k;
                    }
                }
                if (e.surface_len == 0) {
                    free(e.surface); /* "Skipped an empty surface" */
                } else {
                    if (out->n == out->cap) {
                        out->cap = out->cap ? out->cap * 2 : 1024;
                        out->v = (raw_entry *)realloc(out->v, out->cap * sizeof(raw_entry));
                    }
                    out->v[out->n++] = e;
                }
                rec_done = 1;
            } else {
                field_cnt++;
            }
            continue;
        bad_int:
            set_err(err, errcap, "%s: invalid integer \"%.*s\"", name, (int)field.len, field.p);
            free(e.surface);
            goto fail;
        }
    }
    free(field.p);
    return 1;
fail:
    free(field.p);
    entry_vec_free(out);
    return 0;
}

/* ----------------------------------------- double-array trie (crawdad 0.3.0)
 * Restatement of the published crawdad::Trie layout: a code
 * mapper (char -> code, codes handed out by descending character frequency,
 * END_CODE = 0), nodes {base,check} with OFFSET_MASK = 0x7fffffff,
 * is_leaf = base>>31, has_leaf = check>>31; child(n,c) = (base[n]&MASK)^c valid
 * iff (check[child]&MASK)==n; a key ending at node n hangs a leaf on END_CODE
 * whose base field holds the value.  Call site: lexicon/map/trie.rs:49-57. */

#define DA_MASK 0x7fffffffu
#define DA_INVALID 0xffffffffu

typedef struct {
    uint32_t *mapper;      /* code point -> code, DA_INVALID when unmapped */
    uint32_t mapper_len;
    uint32_t alphabet;     /* number of codes incl. END_CODE */
    uint32_t *base, *check;
    uint32_t n_nodes;
} da_trie;

typedef struct {
    uint32_t *codes; /* mapped codes */
    uint32_t len;
    uint32_t value;
} da_key;

typedef struct {
    da_trie *t;
    uint32_t cap;
    uint32_t block;         /* power of two > max code */
    uint32_t *nxt, *prv;    /* free list (valid for free slots) */
    uint8_t *used;
    uint32_t head, tail;    /* free list ends or DA_INVALID */
    uint32_t search_head;   /* roving start for multi-child placements */
} da_builder;

static void da_grow(da_builder *b) {
    uint32_t old = b->cap, nc = old + b->block;
    da_trie *t = b->t;
    t->base = (uint32_t *)realloc(t->base, (size_t)nc * 4);
    t->check = (uint32_t *)realloc(t->check, (size_t)nc * 4);
    b->nxt = (uint32_t *)realloc(b->nxt, (size_t)nc * 4);
    b->prv = (uint32_t *)realloc(b->prv, (size_t)nc * 4);
    b->used = (uint8_t *)realloc(b->used, nc);
    for (uint32_t i = old; i < nc; i++) {
        t->base[i] = DA_MASK; /* unused marker: never equals a node index */
        t->check[i] = DA_MASK;
        b->used[i] = 0;
        b->nxt[i] = (i + 1 < nc) ? i + 1 : DA_INVALID;
        b->prv[i] = (i > old) ? i - 1 : b->tail;
    }
    if (b->head == DA_INVALID) b->head = old;
    else b->nxt[b->tail] = old;
    b->tail = nc - 1;
    b->cap = nc;
}

static void da_take(da_builder *b, uint32_t i) {
    /* unlink slot i from the free list */
    uint32_t n = b->nxt[i], p = b->prv[i];
    if (p != DA_INVALID) b->nxt[p] = n; else b->head = n;
    if (n != DA_INVALID) b->prv[n] = p; else b->tail = p;
    if (b->search_head == i) b->search_head = n;
    b->used[i] = 1;
}

static uint32_t da_find_base(da_builder *b, const uint32_t *codes, uint32_t k) {
    for (;;) {
        uint32_t e = (k == 1) ? b->head : (b->search_head != DA_INVALID ? b->search_head : b->head);
        uint32_t tries = 0;
        for (; e != DA_INVALID; e = b->nxt[e]) {
            uint32_t base = e ^ codes[0];
            uint32_t j = 1;
            for (; j < k; j++) if (b->used[base ^ codes[j]]) break;
            if (j == k) {
                if (k > 1 && tries > 64) b->search_head = e;
                return base;
            }
            tries++;
        }
        da_grow(b);
        if (k > 1 && b->search_head == DA_INVALID) b->search_head = b->cap - b->block;
    }
}

static int key_cmp(const void *a, const void *b) {
    const da_key *x = (const da_key *)a, *y = (const da_key *)b;
    uint32_t n = x->len < y->len ? x->len : y->len;
    for (uint32_t i = 0; i < n; i++)
        if (x->codes[i] != y->codes[i]) return x->codes[i] < y->codes[i] ? -1 : 1;
    return (x->len > y->len) - (x->len < y->len);
}

typedef struct { uint32_t node, lo, hi, depth; } da_frame;

/* crawdad::Trie::from_records (map/trie.rs:43): keys must be distinct, non-empty. */
static void da_build(da_trie *t, da_key *keys, uint32_t nkeys, uint32_t alphabet) {
    da_builder b;
    memset(&b, 0, sizeof(b));
    b.t = t;
    b.block = 256;
    while (b.block <= alphabet) b.block <<= 1;
    b.head = b.tail = DA_INVALID;
    b.search_head = DA_INVALID;
    t->alphabet = alphabet;
    t->base = t->check = NULL;
    da_grow(&b);
    da_take(&b, 0); /* root */
    t->base[0] = 0;
    t->check[0] = DA_MASK; /* root has no parent */
    qsort(keys, nkeys, sizeof(da_key), key_cmp);

    da_frame *stack = (da_frame *)malloc(sizeof(da_frame) * 1024);
    size_t sp = 0, scap = 1024;
    uint32_t *codes = (uint32_t *)malloc(sizeof(uint32_t) * (alphabet + 2));
    uint32_t *starts = (uint32_t *)malloc(sizeof(uint32_t) * (alphabet + 3));
    if (nkeys) stack[sp++] = (da_frame){0, 0, nkeys, 0};
    while (sp) {
        da_frame f = stack[--sp];
        uint32_t k = 0, i = f.lo;
        int has_end = 0;
        uint32_t end_value = 0;
        if (keys[i].len == f.depth) { /* key ends here: END_CODE child */
            has_end = 1;
            end_value = keys[i].value;
            codes[k] = 0; starts[k] = i; k++;
            i++;
        }
        while (i < f.hi) {
            uint32_t c = keys[i].codes[f.depth];
            codes[k] = c; starts[k] = i; k++;
            while (i < f.hi && keys[i].codes[f.depth] == c) i++;
        }
        starts[k] = f.hi;
        uint32_t base = da_find_base(&b, codes, k);
        t->base[f.node] = (t->base[f.node] & ~DA_MASK) | base;
        if (has_end) t->check[f.node] |= 0x80000000u; /* has_leaf */
        for (uint32_t j = 0; j < k; j++) {
            uint32_t child = base ^ codes[j];
            da_take(&b, child);
            t->check[child] = f.node;
            t->base[child] = 0;
        }
        if (has_end) {
            uint32_t leaf = base; /* base ^ END_CODE */
            t->base[leaf] = 0x80000000u | end_value; /* is_leaf, value */
        }
        for (uint32_t j = k; j-- > (uint32_t)has_end;) {
            if (sp == scap) { scap *= 2; stack = (da_frame *)realloc(stack, sizeof(da_frame) * scap); }
            stack[sp++] = (da_frame){base ^ codes[j], starts[j], starts[j + 1], f.depth + 1};
        }
    }
    t->n_nodes = b.cap;
    free(stack); free(codes); free(starts);
    free(b.nxt); free(b.prv); free(b.used);
}

static void da_free(da_trie *t) {
    free(t->mapper); free(t->base); free(t->check);
    memset(t, 0, sizeof(*t));
}

/* ----------------------------------------------------------- dictionary types */

typedef struct { uint16_t left_id, right_id; int16_t word_cost; } word_param; /* lexicon/param.rs:6-10 */

typedef struct {
    da_trie trie;            /* lexicon/map.rs:14-17 */
    uint32_t *postings;      /* lexicon/map/posting.rs:7-14: [len, id...] */
    uint32_t n_postings;
    word_param *params;      /* lexicon/param.rs:24-26 */
    char **features;         /* lexicon/feature.rs:4-6 */
    uint32_t *feature_lens;
    uint32_t n_words;
    uint8_t lex_type;        /* dictionary.rs:30-40: System=0 User=1 Unknown=2 */
} lexicon;

typedef struct {
    uint16_t cate_id, left_id, right_id; int16_t word_cost; /* unknown.rs:20-27 */
    char *feature; uint32_t feature_len;
} unk_entry;

typedef struct ora_dict {
    lexicon sys;
    lexicon user;
    int has_user;
    int16_t *matrix;         /* matrix_connector.rs:11-15: data[left*num_right+right] */
    uint32_t num_right, num_left;
    uint32_t chr2inf[65536]; /* character.rs:105-108 */
    char **categories;
    uint32_t n_categories;
    uint32_t *unk_offsets;   /* unknown.rs:63-66 */
    unk_entry *unk_entries;
    uint32_t n_unk;
    uint16_t *map_left, *map_right; /* ConnIdMapper, mapper.rs:9-12 (NULL = None) */
    struct ora_raw_connector *raw;   /* ConnectorWrapper::Raw (connector.rs:30-35): matrix == NULL then */
    struct ora_dual_connector *dual; /* ConnectorWrapper::Dual: matrix == NULL, raw == NULL */
} ora_dict;

/* CharInfo bit layout, character.rs:10-24,56-95 */
#define CI_CATESET(x) ((x) & 0x3FFFFu)
#define CI_BASE(x) (((x) >> 18) & 0xFFu)
#define CI_INVOKE(x) (((x) >> 26) & 1u)
#define CI_GROUP(x) (((x) >> 27) & 1u)
#define CI_LENGTH(x) ((x) >> 28)

static void lexicon_free(lexicon *lx) {
    da_free(&lx->trie);
    free(lx->postings);
    free(lx->params);
    for (uint32_t i = 0; i < lx->n_words; i++) free(lx->features[i]);
    free(lx->features);
    free(lx->feature_lens);
    memset(lx, 0, sizeof(*lx));
}

typedef struct { const char *s; uint32_t len; uint32_t id; } surf_ref;

static int surf_cmp(const void *a, const void *b) {
    const surf_ref *x = (const surf_ref *)a, *y = (const surf_ref *)b;
    uint32_t n = x->len < y->len ? x->len : y->len;
    int c = memcmp(x->s, y->s, n);
    if (c) return c;
    if (x->len != y->len) return x->len < y->len ? -1 : 1;
    return (x->id > y->id) - (x->id < y->id); /* ids ascending within a surface (map.rs:57-59) */
}

typedef struct { uint32_t cp, cnt; } cp_count;
static int cpc_cmp(const void *a, const void *b) {
    const cp_count *x = (const cp_count *)a, *y = (const cp_count *)b;
    if (x->cnt != y->cnt) return x->cnt > y->cnt ? -1 : 1;
    return (x->cp > y->cp) - (x->cp < y->cp);
}

/* Lexicon::from_entries, lexicon.rs:85-96 -> WordMap::new map.rs:20-31 ->
 * WordMapBuilder::build map.rs:61-72 (distinct surfaces, ids in insertion
 * order) -> PostingsBuilder::push posting.rs:35-40. */
static void lexicon_build(lexicon *lx, const entry_vec *ev, uint8_t lex_type) {
    memset(lx, 0, sizeof(*lx));
    lx->lex_type = lex_type;
    uint32_t n = (uint32_t)ev->n;
    lx->n_words = n;
    lx->params = (word_param *)malloc(sizeof(word_param) * (n ? n : 1));
    lx->features = (char **)malloc(sizeof(char *) * (n ? n : 1));
    lx->feature_lens = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    surf_ref *refs = (surf_ref *)malloc(sizeof(surf_ref) * (n ? n : 1));
    for (uint32_t i = 0; i < n; i++) {
        const raw_entry *e = &ev->v[i];
        lx->params[i] = (word_param){e->left_id, e->right_id, e->word_cost};
        lx->features[i] = (char *)malloc(e->feature_len + 1);
        memcpy(lx->features[i], e->feature, e->feature_len);
        lx->features[i][e->feature_len] = 0;
        lx->feature_lens[i] = e->feature_len;
        refs[i] = (surf_ref){e->surface, e->surface_len, i};
    }
    qsort(refs, n, sizeof(surf_ref), surf_cmp);
    /* postings + distinct keys */
    lx->postings = (uint32_t *)malloc(sizeof(uint32_t) * (2 * (size_t)n + 1));
    da_key *keys = (da_key *)malloc(sizeof(da_key) * (n ? n : 1));
    uint32_t nkeys = 0, np = 0;
    /* frequency of code points over distinct keys */
    uint32_t max_cp = 0;
    size_t total_chars = 0;
    for (uint32_t i = 0; i < n;) {
        uint32_t j = i;
        while (j < n && refs[j].len == refs[i].len && memcmp(refs[j].s, refs[i].s, refs[i].len) == 0) j++;
        uint32_t off = np;
        lx->postings[np++] = j - i;
        for (uint32_t k = i; k < j; k++) lx->postings[np++] = refs[k].id;
        /* decode key */
        uint32_t *cps = (uint32_t *)malloc(sizeof(uint32_t) * refs[i].len);
        uint32_t nc = 0;
        for (uint32_t p = 0; p < refs[i].len;) {
            uint32_t cp;
            int k = utf8_decode((const uint8_t *)refs[i].s + p, refs[i].len - p, &cp);
            cps[nc++] = cp;
            if (cp > max_cp) max_cp = cp;
            p += k;
        }
        keys[nkeys++] = (da_key){cps, nc, off};
        total_chars += nc;
        i = j;
    }
    lx->n_postings = np;
    /* CodeMapper: codes by descending frequency */
    uint32_t mlen = max_cp + 1;
    uint32_t *freq = (uint32_t *)calloc(mlen, 4);
    for (uint32_t i = 0; i < nkeys; i++)
        for (uint32_t j = 0; j < keys[i].len; j++) freq[keys[i].codes[j]]++;
    uint32_t ndist = 0;
    for (uint32_t c = 0; c < mlen; c++) if (freq[c]) ndist++;
    cp_count *cc = (cp_count *)malloc(sizeof(cp_count) * (ndist ? ndist : 1));
    uint32_t q = 0;
    for (uint32_t c = 0; c < mlen; c++) if (freq[c]) cc[q++] = (cp_count){c, freq[c]};
    qsort(cc, ndist, sizeof(cp_count), cpc_cmp);
    lx->trie.mapper = (uint32_t *)malloc(sizeof(uint32_t) * mlen);
    lx->trie.mapper_len = mlen;
    for (uint32_t c = 0; c < mlen; c++) lx->trie.mapper[c] = DA_INVALID;
    for (uint32_t i = 0; i < ndist; i++) lx->trie.mapper[cc[i].cp] = i + 1; /* END_CODE = 0 */
    for (uint32_t i = 0; i < nkeys; i++)
        for (uint32_t j = 0; j < keys[i].len; j++) keys[i].codes[j] = lx->trie.mapper[keys[i].codes[j]];
    da_build(&lx->trie, keys, nkeys, ndist + 1);
    for (uint32_t i = 0; i < nkeys; i++) free(keys[i].codes);
    free(keys); free(cc); free(freq); free(refs);
    (void)total_chars;
}

/* Lexicon::verify, lexicon.rs:68-82 */
static int lexicon_verify(const lexicon *lx, uint32_t num_left, uint32_t num_right) {
    for (uint32_t i = 0; i < lx->n_words; i++) {
        if (num_left <= lx->params[i].left_id) return 0;
        if (num_right <= lx->params[i].right_id) return 0;
    }
    return 1;
}

/* -------------------------------------------- matrix.def (matrix_connector.rs) */

static size_t next_line(const char *buf, size_t len, size_t pos, size_t *ls, size_t *le) {
    /* BufRead::lines(): split on '\n'; '\r' is stripped only as part of "\r\n" (not from an unterminated last line) */
    *ls = pos;
    size_t i = pos;
    while (i < len && buf[i] != '\n') i++;
    size_t e = i;
    if (i < len && e > pos && buf[e - 1] == '\r') e--;
    *le = e;
    return i < len ? i + 1 : len;
}

static int split_space(const char *s, size_t n, const char **cols, size_t *lens, int maxcols) {
    /* str::split(' '): empty items are kept */
    int k = 0;
    size_t st = 0;
    for (size_t i = 0; i <= n; i++) {
        if (i == n || s[i] == ' ') {
            if (k < maxcols) { cols[k] = s + st; lens[k] = i - st; }
            k++;
            st = i + 1;
        }
    }
    return k;
}

/* MatrixConnector::from_reader, matrix_connector.rs:27-51; parse_header 53-64;
 * parse_body 66-77.  Storage: data[left_id * num_right + right_id] (l.47). */
static int parse_matrix_def(const char *buf, size_t len, ora_dict *d, char *err, size_t errcap) {
    size_t pos = 0, ls, le;
    if (len == 0) { set_err(err, errcap, "matrix.def: empty input"); return 0; }
    pos = next_line(buf, len, pos, &ls, &le);
    const char *cols[4]; size_t lens[4];
    int k = split_space(buf + ls, le - ls, cols, lens, 4);
    long long nr, nl;
    if (k != 2 || !parse_int(cols[0], lens[0], 0, 0, 65535, &nr) || !parse_int(cols[1], lens[1], 0, 0, 65535, &nl)) {
        set_err(err, errcap, "matrix.def: The header must consists of two integers separated by spaces, %.*s",
                (int)(le - ls), buf + ls);
        return 0;
    }
    d->num_right = (uint32_t)nr; d->num_left = (uint32_t)nl;
    d->matrix = (int16_t *)calloc(((size_t)nr * nl) > 0 ? (size_t)nr * nl : 1, 2);
    while (pos < len) {
        pos = next_line(buf, len, pos, &ls, &le);
        if (le == ls) continue;
        k = split_space(buf + ls, le - ls, cols, lens, 4);
        long long r, l, c;
        if (k != 3 || !parse_int(cols[0], lens[0], 0, 0, 1LL << 32, &r) || !parse_int(cols[1], lens[1], 0, 0, 1LL << 32, &l) ||
            !parse_int(cols[2], lens[2], 1, -32768, 32767, &c)) {
            set_err(err, errcap,
                    "matrix.def: A row other than the header must consists of three integers separated by spaces, %.*s",
                    (int)(le - ls), buf + ls);
            return 0;
        }
        if (nr <= r || nl <= l) {
            set_err(err, errcap, "matrix.def: left/right_id must be within num_left/right.");
            return 0;
        }
        d->matrix[(size_t)l * nr + r] = (int16_t)c;
    }
    return 1;
}

/* ------------------------------------------------- char.def (character.rs) */

static int is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

static int find_category(const ora_dict *d, const char *s, size_t n) {
    for (uint32_t i = 0; i < d->n_categories; i++)
        if (strlen(d->categories[i]) == n && memcmp(d->categories[i], s, n) == 0) return (int)i;
    return -1;
}

typedef struct { uint32_t start, end; int ncat; int cats[32]; } char_range;

/* CharProperty::from_reader, character.rs:140-191; parse_char_category 218-243;
 * parse_char_range 245-281; encode_cate_info 193-216. */
static int parse_char_def(const char *buf, size_t len, ora_dict *d, char *err, size_t errcap) {
    /* cate_map: name -> id, DEFAULT = 0 (l.145); cate2info: id -> CharInfo (l.160-163) */
    uint32_t cate2info[256];
    uint8_t defined[256];
    memset(defined, 0, sizeof(defined));
    d->categories = (char **)calloc(256, sizeof(char *));
    d->categories[0] = strdup("DEFAULT");
    d->n_categories = 1;
    char_range *ranges = NULL;
    size_t nranges = 0, rcap = 0;
    /* range lines may name categories before they are defined: keep names, resolve later */
    typedef struct { const char *s; size_t n; } tok;
    tok (*rtoks)[32] = NULL;

    size_t pos = 0, ls, le;
    int ok = 1;
    while (pos < len && ok) {
        pos = next_line(buf, len, pos, &ls, &le);
        while (ls < le && is_ws(buf[ls])) ls++;
        while (le > ls && is_ws(buf[le - 1])) le--;
        if (ls == le || buf[ls] == '#') continue;
        tok cols[40];
        int nc = 0;
        for (size_t i = ls; i < le;) {
            while (i < le && is_ws(buf[i])) i++;
            if (i >= le) break;
            size_t st = i;
            while (i < le && !is_ws(buf[i])) i++;
            if (nc < 40) { cols[nc].s = buf + st; cols[nc].n = i - st; nc++; }
        }
        if (!(le - ls >= 2 && buf[ls] == '0' && buf[ls + 1] == 'x')) {
            if (nc < 4) {
                set_err(err, errcap, "char.def: A character category must consists of four items separated by spaces, %.*s",
                        (int)(le - ls), buf + ls);
                ok = 0; break;
            }
            int invoke, group;
            if (cols[1].n == 1 && (cols[1].s[0] == '0' || cols[1].s[0] == '1')) invoke = cols[1].s[0] == '1';
            else { set_err(err, errcap, "char.def: INVOKE must be 1 or 0."); ok = 0; break; }
            if (cols[2].n == 1 && (cols[2].s[0] == '0' || cols[2].s[0] == '1')) group = cols[2].s[0] == '1';
            else { set_err(err, errcap, "char.def: GROUP must be 1 or 0."); ok = 0; break; }
            long long length;
            if (!parse_int(cols[3].s, cols[3].n, 0, 0, 65535, &length)) { set_err(err, errcap, "char.def: invalid LENGTH"); ok = 0; break; }
            int id = find_category(d, cols[0].s, cols[0].n);
            if (id < 0) {
                id = (int)d->n_categories;
                if (id >= 18) { set_err(err, errcap, "char.def: too many categories"); ok = 0; break; }
                d->categories[id] = strndup(cols[0].s, cols[0].n);
                d->n_categories++;
            }
            if (length >= 16) { set_err(err, errcap, "char.def: LENGTH must be < 16"); ok = 0; break; }
            /* CharInfo::new(0, cate_id, invoke, group, length), character.rs:39-63 */
            cate2info[id] = ((uint32_t)id << 18) | ((uint32_t)invoke << 26) | ((uint32_t)group << 27) | ((uint32_t)length << 28);
            defined[id] = 1;
        } else {
            if (nc < 2) {
                set_err(err, errcap, "char.def: A character range must have two items at least, %.*s", (int)(le - ls), buf + ls);
                ok = 0; break;
            }
            /* cols[0] = 0xAAAA or 0xAAAA..0xBBBB */
            const char *r = cols[0].s; size_t rn = cols[0].n;
            size_t dd = 0; int has_dd = 0;
            for (size_t i = 0; i + 1 < rn; i++) if (r[i] == '.' && r[i + 1] == '.') { dd = i; has_dd = 1; break; }
            unsigned long long st = 0, en = 0;
            {
                const char *a = r; size_t an = has_dd ? dd : rn;
                while (an >= 2 && a[0] == '0' && a[1] == 'x') { a += 2; an -= 2; } /* trim_start_matches("0x") */
                if (an == 0 || an > 8) { set_err(err, errcap, "char.def: invalid range"); ok = 0; break; }
                for (size_t i = 0; i < an; i++) {
                    char c = a[i]; int v;
                    if (c >= '0' && c <= '9') v = c - '0'; else if (c >= 'a' && c <= 'f') v = c - 'a' + 10;
                    else if (c >= 'A' && c <= 'F') v = c - 'A' + 10; else { v = -1; }
                    if (v < 0) { ok = 0; break; }
                    st = st * 16 + v;
                }
                if (!ok) { set_err(err, errcap, "char.def: invalid range"); break; }
            }
            if (has_dd) {
                const char *a = r + dd + 2; size_t an = rn - dd - 2;
                while (an >= 2 && a[0] == '0' && a[1] == 'x') { a += 2; an -= 2; }
                if (an == 0 || an > 8) { set_err(err, errcap, "char.def: invalid range"); ok = 0; break; }
                for (size_t i = 0; i < an; i++) {
                    char c = a[i]; int v;
                    if (c >= '0' && c <= '9') v = c - '0'; else if (c >= 'a' && c <= 'f') v = c - 'a' + 10;
                    else if (c >= 'A' && c <= 'F') v = c - 'A' + 10; else { v = -1; }
                    if (v < 0) { ok = 0; break; }
                    en = en * 16 + v;
                }
                if (!ok) { set_err(err, errcap, "char.def: invalid range"); break; }
                en += 1;
            } else en = st + 1;
            if (st >= en) { set_err(err, errcap, "char.def: The start of a character range must be no more than the end"); ok = 0; break; }
            if (st > 0xFFFF || en > 0x10000) { set_err(err, errcap, "char.def: A character range must be no more 0xFFFF"); ok = 0; break; }
            if (nranges == rcap) {
                rcap = rcap ? rcap * 2 : 64;
                ranges = (char_range *)realloc(ranges, rcap * sizeof(char_range));
                rtoks = realloc(rtoks, rcap * sizeof(*rtoks));
            }
            char_range *cr = &ranges[nranges];
            cr->start = (uint32_t)st; cr->end = (uint32_t)en; cr->ncat = 0;
            for (int i = 1; i < nc && cr->ncat < 32; i++) {
                if (cols[i].s[0] == '#') break;
                rtoks[nranges][cr->ncat].s = cols[i].s;
                rtoks[nranges][cr->ncat].n = cols[i].n;
                cr->ncat++;
            }
            nranges++;
        }
    }
    if (ok) {
        if (!defined[0]) { set_err(err, errcap, "char.def: Undefined category: DEFAULT"); ok = 0; }
    }
    if (ok) {
        /* init: encode_cate_info(["DEFAULT"]) */
        uint32_t init = cate2info[0] | (1u << 0);
        for (uint32_t i = 0; i < 65536; i++) d->chr2inf[i] = init;
        for (size_t r = 0; r < nranges && ok; r++) {
            char_range *cr = &ranges[r];
            if (cr->ncat == 0) { set_err(err, errcap, "char.def: range without category"); ok = 0; break; }
            uint32_t info = 0, set = 0;
            for (int i = 0; i < cr->ncat; i++) {
                int id = find_category(d, rtoks[r][i].s, rtoks[r][i].n);
                if (id < 0 || !defined[id]) {
                    set_err(err, errcap, "char.def: Undefined category: %.*s", (int)rtoks[r][i].n, rtoks[r][i].s);
                    ok = 0; break;
                }
                if (i == 0) info = cate2info[id];
                set |= 1u << CI_BASE(cate2info[id]);
            }
            if (!ok) break;
            info = (info & ~0x3FFFFu) | set;
            for (uint32_t c = cr->start; c < cr->end; c++) d->chr2inf[c] = info;
        }
    }
    free(ranges);
    free(rtoks);
    return ok;
}

/* --------------------------------------------------- unk.def (unknown.rs) */

/* UnkHandler::from_reader, unknown.rs:230-263: rows grouped by category id,
 * stable within a category; offsets has n_categories+1 items. */
static int parse_unk_def(const char *buf, size_t len, ora_dict *d, char *err, size_t errcap) {
    entry_vec ev;
    if (!parse_lex_csv(buf, len, "unk.def", &ev, err, errcap)) return 0;
    uint32_t ncat = d->n_categories;
    int *cat = (int *)malloc(sizeof(int) * (ev.n ? ev.n : 1));
    for (size_t i = 0; i < ev.n; i++) {
        cat[i] = find_category(d, ev.v[i].surface, ev.v[i].surface_len);
        if (cat[i] < 0) {
            set_err(err, errcap, "unk.def: Undefined category: %s", ev.v[i].surface);
            free(cat); entry_vec_free(&ev);
            return 0;
        }
    }
    d->unk_offsets = (uint32_t *)calloc(ncat + 1, 4);
    d->unk_entries = (unk_entry *)calloc(ev.n ? ev.n : 1, sizeof(unk_entry));
    uint32_t k = 0;
    for (uint32_t c = 0; c < ncat; c++) {
        d->unk_offsets[c] = k;
        for (size_t i = 0; i < ev.n; i++) {
            if ((uint32_t)cat[i] != c) continue;
            unk_entry *u = &d->unk_entries[k++];
            u->cate_id = (uint16_t)c;
            u->left_id = ev.v[i].left_id; u->right_id = ev.v[i].right_id; u->word_cost = ev.v[i].word_cost;
            u->feature = strndup(ev.v[i].feature, ev.v[i].feature_len);
            u->feature_len = ev.v[i].feature_len;
        }
    }
    d->unk_offsets[ncat] = k;
    d->n_unk = k;
    free(cat);
    entry_vec_free(&ev);
    return 1;
}

/* ------------------------------------------------------------- dictionary API */

ORA_API void ora_dict_free(ora_dict *d) {
    if (!d) return;
    lexicon_free(&d->sys);
    if (d->has_user) lexicon_free(&d->user);
    free(d->matrix);
    raw_free(d->raw);
    dual_free(d->dual);
    if (d->categories) {
        for (uint32_t i = 0; i < d->n_categories; i++) free(d->categories[i]);
        free(d->categories);
    }
    free(d->map_left); free(d->map_right);
    free(d->unk_offsets);
    if (d->unk_entries) {
        for (uint32_t i = 0; i < d->n_unk; i++) free(d->unk_entries[i].feature);
        free(d->unk_entries);
    }
    free(d);
}

/* SystemDictionaryBuilder::from_readers builder.rs:64-89 + build 16-47.
 * If matrix_bin != NULL the connection matrix is taken from a binary i16 array
 * laid out data[left*num_right+right] (synthetic dictionaries; no text form). */
static ora_dict *dict_build(const char *lex, size_t lex_len, const char *mat, size_t mat_len,
                            const int16_t *matrix_bin, uint32_t num_right, uint32_t num_left,
                            const char *chr, size_t chr_len, const char *unk, size_t unk_len,
                            char *err, size_t errcap) {
    ora_dict *d = (ora_dict *)calloc(1, sizeof(ora_dict));
    entry_vec ev;
    memset(&ev, 0, sizeof(ev));
    if (!parse_lex_csv(lex, lex_len, "lex.csv", &ev, err, errcap)) goto fail;
    if (matrix_bin) {
        d->num_right = num_right; d->num_left = num_left;
        size_t n = (size_t)num_right * num_left;
        d->matrix = (int16_t *)malloc(n ? n * 2 : 2);
        memcpy(d->matrix, matrix_bin, n * 2);
    } else if (!mat) { /* compact connector: only the id ranges are known here */
        d->num_right = num_right; d->num_left = num_left;
    } else if (!parse_matrix_def(mat, mat_len, d, err, errcap)) goto fail;
    if (!parse_char_def(chr, chr_len, d, err, errcap)) goto fail;
    if (!parse_unk_def(unk, unk_len, d, err, errcap)) goto fail;
    lexicon_build(&d->sys, &ev, 0);
    entry_vec_free(&ev);
    if (!lexicon_verify(&d->sys, d->num_left, d->num_right)) {
        set_err(err, errcap, "system_lexicon_rdr includes invalid connection ids.");
        goto fail2;
    }
    for (uint32_t i = 0; i < d->n_unk; i++) { /* UnkHandler::verify unknown.rs:214-227 */
        if (d->num_left <= d->unk_entries[i].left_id || d->num_right <= d->unk_entries[i].right_id) {
            set_err(err, errcap, "unk_handler_rdr includes invalid connection ids.");
            goto fail2;
        }
    }
    return d;
fail:
    entry_vec_free(&ev);
fail2:
    ora_dict_free(d);
    return NULL;
}

/* SystemDictionaryBuilder::from_readers_with_bigram_info, builder.rs:111-160: a RawConnector, or with dual_connector = true a
 * DualConnector (builder.rs:134-150) */
ORA_API ora_dict *ora_dict_from_sources_bigram2(const char *lex, size_t lex_len, const char *right, size_t right_len, const char *left,
                                                size_t left_len, const char *cost, size_t cost_len, const char *chr, size_t chr_len,
                                                const char *unk, size_t unk_len, int dual, char *err, size_t errcap) {
    if (dual) {
        ora_dual_connector *dc = dual_from_sources(right, right_len, left, left_len, cost, cost_len, err, errcap);
        if (!dc) return NULL;
        ora_dict *d = dict_build(lex, lex_len, NULL, 0, NULL, dc->num_right, dc->num_left, chr, chr_len, unk, unk_len, err, errcap);
        if (!d) { dual_free(dc); return NULL; }
        d->dual = dc;
        return d;
    }
    ora_raw_connector *rc = raw_from_sources(right, right_len, left, left_len, cost, cost_len, err, errcap);
    if (!rc) return NULL;
    ora_dict *d = dict_build(lex, lex_len, NULL, 0, NULL, rc->num_right, rc->num_left, chr, chr_len, unk, unk_len, err, errcap);
    if (!d) { raw_free(rc); return NULL; }
    d->raw = rc;
    return d;
}
ORA_API ora_dict *ora_dict_from_sources_bigram(const char *lex, size_t lex_len, const char *right, size_t right_len, const char *left,
                                               size_t left_len, const char *cost, size_t cost_len, const char *chr, size_t chr_len,
                                               const char *unk, size_t unk_len, char *err, size_t errcap) {
    return ora_dict_from_sources_bigram2(lex, lex_len, right, right_len, left, left_len, cost, cost_len, chr, chr_len, unk, unk_len, 0, err, errcap);
}

ORA_API ora_dict *ora_dict_from_sources(const char *lex, size_t lex_len, const char *mat, size_t mat_len,
                                        const char *chr, size_t chr_len, const char *unk, size_t unk_len,
                                        char *err, size_t errcap) {
    return dict_build(lex, lex_len, mat, mat_len, NULL, 0, 0, chr, chr_len, unk, unk_len, err, errcap);
}

ORA_API ora_dict *ora_dict_from_sources_binmatrix(const char *lex, size_t lex_len, const int16_t *matrix,
                                                  uint32_t num_right, uint32_t num_left, const char *chr,
                                                  size_t chr_len, const char *unk, size_t unk_len, char *err,
                                                  size_t errcap) {
    return dict_build(lex, lex_len, NULL, 0, matrix, num_right, num_left, chr, chr_len, unk, unk_len, err, errcap);
}

/* Dictionary::reset_user_lexicon_from_reader, dictionary.rs:209-229 (no id
 * mapper in this oracle: mapper is always None for text-built dictionaries). */
ORA_API int ora_dict_set_user_lexicon(ora_dict *d, const char *csv, size_t len, char *err, size_t errcap) {
    if (d->has_user) { lexicon_free(&d->user); d->has_user = 0; }
    if (!csv) return 1;
    entry_vec ev;
    if (!parse_lex_csv(csv, len, "lex.csv", &ev, err, errcap)) return 0;
    lexicon lx;
    lexicon_build(&lx, &ev, 1);
    entry_vec_free(&ev);
    if (d->map_left) { /* dictionary.rs:214-217 */
        for (uint32_t i = 0; i < lx.n_words; i++) {
            if (lx.params[i].left_id >= d->num_left || lx.params[i].right_id >= d->num_right) {
                lexicon_free(&lx);
                set_err(err, errcap, "user_lexicon_rdr: includes invalid connection ids.");
                return 0;
            }
            lx.params[i].left_id = d->map_left[lx.params[i].left_id];
            lx.params[i].right_id = d->map_right[lx.params[i].right_id];
        }
    }
    if (!lexicon_verify(&lx, d->num_left, d->num_right)) {
        lexicon_free(&lx);
        set_err(err, errcap, "user_lexicon_rdr: includes invalid connection ids.");
        return 0;
    }
    d->user = lx;
    d->has_user = 1;
    return 1;
}

/* ConnIdMapper::parse, mapper.rs:49-80 */
static uint16_t *parse_id_map(const uint16_t *map, size_t n, size_t *out_len, char *err, size_t errcap) {
    size_t len = n + 1;
    if (len > 0x10000) { set_err(err, errcap, "map: too many ids"); return NULL; }
    uint16_t *new_ids = (uint16_t *)malloc(2 * len);
    for (size_t i = 0; i < len; i++) new_ids[i] = 0xFFFF;
    new_ids[0] = 0;
    for (size_t new_id = 1; new_id < len; new_id++) {
        uint32_t old_id = map[new_id - 1];
        if (old_id == 0) { set_err(err, errcap, "map: Id 0 is reserved."); free(new_ids); return NULL; }
        if (old_id >= len) { set_err(err, errcap, "map: ids are out of range."); free(new_ids); return NULL; }
        if (new_ids[old_id] != 0xFFFF) { set_err(err, errcap, "map: ids are duplicate."); free(new_ids); return NULL; }
        new_ids[old_id] = (uint16_t)new_id;
    }
    *out_len = len;
    return new_ids;
}

/* Dictionary::map_connection_ids_from_iter, dictionary.rs:245-259 (lexicon/param.rs:48-53,
 * matrix_connector.rs:99-116, unknown.rs:206-211) */
ORA_API int ora_dict_map_connection_ids(ora_dict *d, const uint16_t *lmap, size_t nl, const uint16_t *rmap, size_t nr,
                                        char *err, size_t errcap) {
    size_t ll, rl;
    uint16_t *ml = parse_id_map(lmap, nl, &ll, err, errcap);
    if (!ml) return 0;
    uint16_t *mr = parse_id_map(rmap, nr, &rl, err, errcap);
    if (!mr) { free(ml); return 0; }
    if (ll != d->num_left || rl != d->num_right) {
        set_err(err, errcap, "map: the mappings must cover every connection id except 0");
        free(ml); free(mr);
        return 0;
    }
    lexicon *lxs[2] = {&d->sys, d->has_user ? &d->user : NULL};
    for (int k = 0; k < 2; k++) {
        if (!lxs[k]) continue;
        for (uint32_t i = 0; i < lxs[k]->n_words; i++) {
            lxs[k]->params[i].left_id = ml[lxs[k]->params[i].left_id];
            lxs[k]->params[i].right_id = mr[lxs[k]->params[i].right_id];
        }
    }
    if (d->raw) raw_map_ids(d->raw, ml, mr);
    else if (d->dual) dual_map_ids(d->dual, ml, mr);
    else {
        size_t n = (size_t)d->num_left * d->num_right;
        int16_t *mapped = (int16_t *)malloc(n ? n * 2 : 2);
        for (uint32_t r = 0; r < d->num_right; r++)
            for (uint32_t l = 0; l < d->num_left; l++)
                mapped[(size_t)ml[l] * d->num_right + mr[r]] = d->matrix[(size_t)l * d->num_right + r];
        free(d->matrix);
        d->matrix = mapped;
    }
    for (uint32_t i = 0; i < d->n_unk; i++) {
        d->unk_entries[i].left_id = ml[d->unk_entries[i].left_id];
        d->unk_entries[i].right_id = mr[d->unk_entries[i].right_id];
    }
    free(d->map_left); free(d->map_right);
    d->map_left = ml; d->map_right = mr;
    return 1;
}

ORA_API uint32_t ora_dict_num_words(const ora_dict *d, int lex_type) {
    if (lex_type == 0) return d->sys.n_words;
    if (lex_type == 1) return d->has_user ? d->user.n_words : 0;
    return d->n_unk;
}
ORA_API uint32_t ora_dict_num_left(const ora_dict *d) { return d->num_left; }
ORA_API uint32_t ora_dict_num_right(const ora_dict *d) { return d->num_right; }
/* ConnectorCost::cost(right_id, left_id), matrix_connector.rs:119-125 */
ORA_API int32_t ora_dict_conn_cost(const ora_dict *d, uint32_t right_id, uint32_t left_id) {
    if (d->raw) return raw_cost(d->raw, right_id, left_id);
    if (d->dual) return dual_cost(d->dual, right_id, left_id);
    return (int32_t)d->matrix[(size_t)left_id * d->num_right + right_id];
}
/* CharProperty::char_info, character.rs:112-116 */
ORA_API uint32_t ora_dict_char_info(const ora_dict *d, uint32_t cp) { return cp < 65536 ? d->chr2inf[cp] : d->chr2inf[0]; }
ORA_API int ora_dict_cate_id(const ora_dict *d, const char *name) { return find_category(d, name, strlen(name)); }
ORA_API uint32_t ora_dict_num_categories(const ora_dict *d) { return d->n_categories; }
ORA_API uint32_t ora_dict_unk_offset(const ora_dict *d, uint32_t cate) { return d->unk_offsets[cate]; }

/* Dictionary::word_feature dictionary.rs:108-114 / word_param 98-104 */
ORA_API const char *ora_dict_word_feature(const ora_dict *d, int lex_type, uint32_t word_id, uint32_t *len) {
    if (lex_type == 0) { *len = d->sys.feature_lens[word_id]; return d->sys.features[word_id]; }
    if (lex_type == 1) { *len = d->user.feature_lens[word_id]; return d->user.features[word_id]; }
    *len = d->unk_entries[word_id].feature_len;
    return d->unk_entries[word_id].feature;
}
ORA_API void ora_dict_word_param(const ora_dict *d, int lex_type, uint32_t word_id, int32_t out[3]) {
    if (lex_type == 2) {
        out[0] = d->unk_entries[word_id].left_id; out[1] = d->unk_entries[word_id].right_id; out[2] = d->unk_entries[word_id].word_cost;
    } else {
        const lexicon *lx = lex_type == 0 ? &d->sys : &d->user;
        out[0] = lx->params[word_id].left_id; out[1] = lx->params[word_id].right_id; out[2] = lx->params[word_id].word_cost;
    }
}

/* ---------------------------------------------------------------- counters */

typedef struct ora_counters {
    uint64_t n_sentences, n_bytes, n_chars, n_trie_steps, n_trie_hits, n_lex_matches, n_unk_nodes;
    uint64_t n_nodes, n_pairs_ref, n_pairs_dedup, n_tokens;
} ora_counters;

/* ------------------------------------------------------ tokenizer / worker */

typedef struct ora_tokenizer {
    const ora_dict *dict;
    int has_space; uint32_t space_cateset; /* tokenizer.rs:16,50 */
    int has_max_grouping; uint32_t max_grouping_len; /* tokenizer.rs:17,67-74 */
} ora_tokenizer;

typedef struct {
    uint32_t word_id;
    uint8_t lex_type;
    uint32_t start_node, start_word;
    uint16_t left_id, right_id, min_idx;
    int32_t min_cost;
} node; /* tokenizer/lattice.rs:13-23 */

typedef struct { node *v; uint32_t n, cap; } node_vec;

typedef struct ora_token {
    uint32_t start_char, end_char, start_byte, end_byte;
    uint32_t word_idx; /* lex_type << 30 | word_id */
    int32_t total_cost;
} ora_token;

typedef struct ora_worker {
    const ora_tokenizer *tok;
    /* Sentence, sentence.rs:4-10 */
    const uint8_t *input; size_t input_len;
    uint32_t *chars, *c2b, *cinfos, *groupable;
    uint32_t len_char, char_cap;
    /* Lattice, lattice.rs:39-43 */
    node_vec *ends; uint32_t ends_len;
    node eos;
    /* top_nodes, worker.rs:17 */
    uint32_t *top_end; node *top_node; uint32_t n_top, top_cap;
    /* dedup stamps for counters */
    uint32_t *left_stamp; uint32_t stamp;
    ora_counters cnt;
} ora_worker;

#define MAX_COST INT32_MAX
#define INVALID_IDX 0xFFFFu

/* Tokenizer::new + ignore_space + max_grouping_len, tokenizer.rs:26-74 */
ORA_API ora_tokenizer *ora_tokenizer_new(const ora_dict *d, int ignore_space, uint32_t max_grouping_len, char *err, size_t errcap) {
    ora_tokenizer *t = (ora_tokenizer *)calloc(1, sizeof(*t));
    t->dict = d;
    if (ignore_space) {
        int id = find_category(d, "SPACE", 5);
        if (id < 0) {
            set_err(err, errcap, "dict: SPACE is not defined in the input dictionary (i.e., char.def).");
            free(t);
            return NULL;
        }
        t->has_space = 1;
        t->space_cateset = 1u << id;
    }
    if (max_grouping_len != 0) { t->has_max_grouping = 1; t->max_grouping_len = max_grouping_len; }
    return t;
}
ORA_API void ora_tokenizer_free(ora_tokenizer *t) { free(t); }

ORA_API ora_worker *ora_worker_new(const ora_tokenizer *t) {
    ora_worker *w = (ora_worker *)calloc(1, sizeof(*w));
    w->tok = t;
    w->left_stamp = (uint32_t *)calloc(t->dict->num_left ? t->dict->num_left : 1, 4);
    return w;
}
ORA_API void ora_worker_free(ora_worker *w) {
    if (!w) return;
    free(w->chars); free(w->c2b); free(w->cinfos); free(w->groupable);
    for (uint32_t i = 0; i < w->ends_len; i++) free(w->ends[i].v);
    free(w->ends);
    free(w->top_end); free(w->top_node); free(w->left_stamp);
    free(w);
}

/* Worker::reset_sentence worker.rs:34-45 -> Sentence::compile sentence.rs:34-71.
 * Returns 0 on invalid UTF-8 (unrepresentable as a Rust &str). */
ORA_API int ora_worker_reset_sentence(ora_worker *w, const uint8_t *s, size_t len) {
    const ora_dict *d = w->tok->dict;
    w->n_top = 0;
    w->len_char = 0;
    w->input = s; w->input_len = len;
    if (len == 0) return 1;
    if (w->char_cap < len + 1) {
        w->char_cap = (uint32_t)(len + 1) * 2;
        w->chars = (uint32_t *)realloc(w->chars, w->char_cap * 4);
        w->c2b = (uint32_t *)realloc(w->c2b, w->char_cap * 4);
        w->cinfos = (uint32_t *)realloc(w->cinfos, w->char_cap * 4);
        w->groupable = (uint32_t *)realloc(w->groupable, w->char_cap * 4);
    }
    /* compute_basic, sentence.rs:40-46 */
    uint32_t n = 0;
    for (size_t bi = 0; bi < len;) {
        uint32_t cp;
        int k = utf8_decode(s + bi, len - bi, &cp);
        if (!k) { w->len_char = 0; return 0; }
        w->chars[n] = cp;
        w->c2b[n] = (uint32_t)bi;
        n++;
        bi += k;
    }
    w->c2b[n] = (uint32_t)len;
    w->len_char = n;
    /* compute_categories, sentence.rs:48-55 */
    for (uint32_t i = 0; i < n; i++) w->cinfos[i] = w->chars[i] < 65536 ? d->chr2inf[w->chars[i]] : d->chr2inf[0];
    /* compute_groupable, sentence.rs:57-71 */
    for (uint32_t i = 0; i < n; i++) w->groupable[i] = 1;
    uint32_t rhs = CI_CATESET(w->cinfos[n - 1]);
    for (uint32_t i = n - 1; i >= 1; i--) {
        uint32_t lhs = CI_CATESET(w->cinfos[i - 1]);
        if ((lhs & rhs) != 0) w->groupable[i - 1] = w->groupable[i] + 1;
        rhs = lhs;
    }
    return 1;
}

/* Lattice::reset lattice.rs:46-64 + insert_bos 72-83 */
static void lattice_reset(ora_worker *w, uint32_t len_char) {
    for (uint32_t i = 0; i < w->ends_len; i++) w->ends[i].n = 0;
    if (w->ends_len <= len_char + 1) {
        uint32_t nl = len_char + 1;
        w->ends = (node_vec *)realloc(w->ends, sizeof(node_vec) * nl);
        for (uint32_t i = w->ends_len; i < nl; i++) {
            w->ends[i].v = (node *)malloc(sizeof(node) * 16);
            w->ends[i].cap = 16;
            w->ends[i].n = 0;
        }
        w->ends_len = nl;
    }
    node bos = {UINT32_MAX, 0, UINT32_MAX, UINT32_MAX, 0xFFFF, 0, INVALID_IDX, 0};
    w->ends[0].v[0] = bos;
    w->ends[0].n = 1;
}

/* Lattice::search_min_node lattice.rs:129-151 (`<=`: ties -> last inserted) */
static inline void search_min_node(const ora_worker *w, uint32_t start_node, uint16_t left_id, uint16_t *min_idx, int32_t *min_cost) {
    const ora_dict *d = w->tok->dict;
    const node_vec *e = &w->ends[start_node];
    const int16_t *row = (d->raw || d->dual) ? NULL : d->matrix + (size_t)left_id * d->num_right; /* matrix_connector.rs:79-85 */
    uint16_t mi = INVALID_IDX;
    int32_t mc = MAX_COST;
    for (uint32_t i = 0; i < e->n; i++) {
        int32_t conn = row ? (int32_t)row[e->v[i].right_id]
                     : d->raw ? raw_cost(d->raw, e->v[i].right_id, left_id)    /* raw_connector.rs:153-161 */
                              : dual_cost(d->dual, e->v[i].right_id, left_id); /* dual_connector.rs:267-279 */
        int32_t nc = (int32_t)((uint32_t)e->v[i].min_cost + (uint32_t)conn); /* wrapping add (release build) */
        if (nc <= mc) { mi = (uint16_t)i; mc = nc; }
    }
    *min_idx = mi; *min_cost = mc;
}

/* Lattice::insert_node lattice.rs:103-127 */
static inline void insert_node(ora_worker *w, int count, uint32_t start_node, uint32_t start_word, uint32_t end_word,
                               uint8_t lex_type, uint32_t word_id, uint16_t left_id, uint16_t right_id, int16_t word_cost) {
    uint16_t mi; int32_t mc;
    search_min_node(w, start_node, left_id, &mi, &mc);
    node_vec *e = &w->ends[end_word];
    if (e->n == e->cap) { e->cap *= 2; e->v = (node *)realloc(e->v, sizeof(node) * e->cap); }
    node *nd = &e->v[e->n++];
    nd->word_id = word_id; nd->lex_type = lex_type; nd->start_node = start_node; nd->start_word = start_word;
    nd->left_id = left_id; nd->right_id = right_id; nd->min_idx = mi;
    nd->min_cost = (int32_t)((uint32_t)mc + (uint32_t)(int32_t)word_cost);
    if (count) {
        w->cnt.n_nodes++;
        w->cnt.n_pairs_ref += w->ends[start_node].n;
        if (w->left_stamp[left_id] != w->stamp) { /* distinct (start_node,left_id) group: SURVEY 8(d) */
            w->left_stamp[left_id] = w->stamp;
            w->cnt.n_pairs_dedup += w->ends[start_node].n;
        }
    }
}

/* Lexicon::common_prefix_iterator lexicon.rs:33-46 -> WordMap map.rs:33-42 ->
 * crawdad common_prefix_search (trie.rs:49-57) -> Postings::ids posting.rs:16-22.
 * Calls insert_node for each match in enumeration order. Returns has_matched. */
static inline int lex_insert_matches(ora_worker *w, int count, const lexicon *lx, uint32_t start_node, uint32_t start_word) {
    const da_trie *t = &lx->trie;
    int matched = 0;
    uint32_t node_idx = 0;
    for (uint32_t i = start_word; i < w->len_char; i++) {
        uint32_t cp = w->chars[i];
        if (count) w->cnt.n_trie_steps++;
        if (cp >= t->mapper_len) break;
        uint32_t mc = t->mapper[cp];
        if (mc == DA_INVALID) break;
        if (t->base[node_idx] >> 31) break; /* leaf has no children */
        uint32_t child = (t->base[node_idx] & DA_MASK) ^ mc;
        if ((t->check[child] & DA_MASK) != node_idx) break;
        node_idx = child;
        if (t->check[node_idx] >> 31) { /* has_leaf */
            uint32_t leaf = t->base[node_idx] & DA_MASK; /* ^ END_CODE */
            uint32_t off = t->base[leaf] & DA_MASK;
            uint32_t n = lx->postings[off];
            if (count) { w->cnt.n_trie_hits++; w->cnt.n_lex_matches += n; }
            for (uint32_t k = 0; k < n; k++) {
                uint32_t wid = lx->postings[off + 1 + k];
                word_param p = lx->params[wid];
                insert_node(w, count, start_node, start_word, i + 1, lx->lex_type, wid, p.left_id, p.right_id, p.word_cost);
                matched = 1;
            }
        }
    }
    return matched;
}

/* UnkHandler::scan_entries unknown.rs:119-137 */
static inline void scan_entries(ora_worker *w, int count, uint32_t start_node, uint32_t start_char, uint32_t end_char, uint32_t cinfo) {
    const ora_dict *d = w->tok->dict;
    uint32_t s = d->unk_offsets[CI_BASE(cinfo)], e = d->unk_offsets[CI_BASE(cinfo) + 1];
    for (uint32_t wid = s; wid < e; wid++) {
        const unk_entry *u = &d->unk_entries[wid];
        if (count) w->cnt.n_unk_nodes++;
        insert_node(w, count, start_node, start_char, end_char, 2, (uint32_t)(uint16_t)wid, u->left_id, u->right_id, u->word_cost);
    }
}

/* UnkHandler::gen_unk_words unknown.rs:69-116 */
static inline void gen_unk_words(ora_worker *w, int count, uint32_t start_node, uint32_t start_char, int has_matched) {
    const ora_tokenizer *t = w->tok;
    uint32_t cinfo = w->cinfos[start_char];
    if (has_matched && !CI_INVOKE(cinfo)) return;
    int grouped = 0;
    uint32_t groupable = w->groupable[start_char];
    if (CI_GROUP(cinfo)) {
        grouped = 1;
        if (!t->has_max_grouping || groupable - 1 <= t->max_grouping_len) {
            scan_entries(w, count, start_node, start_char, start_char + groupable, cinfo);
            has_matched = 1;
        }
    }
    uint32_t lim = CI_LENGTH(cinfo) < groupable ? CI_LENGTH(cinfo) : groupable;
    for (uint32_t i = 1; i <= lim; i++) {
        if (grouped && i == groupable) continue;
        uint32_t end_char = start_char + i;
        if (w->len_char < end_char) break;
        scan_entries(w, count, start_node, start_char, end_char, cinfo);
        has_matched = 1;
    }
    if (!has_matched) scan_entries(w, count, start_node, start_char, start_char + 1, cinfo);
}

/* Tokenizer::add_lattice_edges tokenizer.rs:141-199 */
static inline void add_lattice_edges(ora_worker *w, int count, uint32_t start_node, uint32_t start_word) {
    const ora_dict *d = w->tok->dict;
    int has_matched = 0;
    if (count) w->stamp++;
    if (d->has_user) has_matched |= lex_insert_matches(w, count, &d->user, start_node, start_word);
    has_matched |= lex_insert_matches(w, count, &d->sys, start_node, start_word);
    gen_unk_words(w, count, start_node, start_word, has_matched);
}

/* Tokenizer::build_lattice_inner tokenizer.rs:94-139, Lattice::insert_eos
 * lattice.rs:85-101, Lattice::append_top_nodes lattice.rs:159-168,
 * Worker::tokenize worker.rs:49-55. */
static inline void tokenize_impl(ora_worker *w, int count) {
    const ora_tokenizer *t = w->tok;
    if (w->len_char == 0) return;
    uint32_t len = w->len_char;
    lattice_reset(w, len);
    uint32_t start_node = 0, start_word = 0;
    while (start_word < len) {
        if (w->ends[start_node].n == 0) { /* has_previous_node lattice.rs:155-157 */
            start_word += 1;
            start_node = start_word;
            continue;
        }
        if (t->has_space) {
            int is_space = (CI_CATESET(w->cinfos[start_node]) & t->space_cateset) != 0;
            start_word += is_space ? w->groupable[start_node] : 0;
        }
        if (start_word == len) break;
        add_lattice_edges(w, count, start_node, start_word);
        start_word += 1;
        start_node = start_word;
    }
    /* insert_eos */
    uint16_t mi; int32_t mc;
    if (count) { w->cnt.n_pairs_ref += w->ends[start_node].n; w->cnt.n_pairs_dedup += w->ends[start_node].n; }
    search_min_node(w, start_node, 0, &mi, &mc);
    w->eos.start_node = start_node; w->eos.start_word = len; w->eos.min_idx = mi; w->eos.min_cost = mc;
    /* append_top_nodes */
    uint32_t end_node = start_node;
    uint16_t min_idx = mi;
    while (end_node != 0) {
        const node *nd = &w->ends[end_node].v[min_idx];
        if (w->n_top == w->top_cap) {
            w->top_cap = w->top_cap ? w->top_cap * 2 : 64;
            w->top_end = (uint32_t *)realloc(w->top_end, 4 * w->top_cap);
            w->top_node = (node *)realloc(w->top_node, sizeof(node) * w->top_cap);
        }
        w->top_end[w->n_top] = end_node;
        w->top_node[w->n_top] = *nd;
        w->n_top++;
        end_node = nd->start_node;
        min_idx = nd->min_idx;
    }
    if (count) { w->cnt.n_sentences++; w->cnt.n_bytes += w->input_len; w->cnt.n_chars += len; w->cnt.n_tokens += w->n_top; }
}

/* Lattice::add_connid_counts, lattice.rs:170-183 (Worker::update_connid_counts, worker.rs:86-93).
 * lid / rid: caller-owned counters of num_left / num_right entries, incremented in place. */
ORA_API void ora_worker_add_connid_counts(const ora_worker *w, uint64_t *lid, uint64_t *rid) {
    if (w->len_char == 0) return;
    for (uint32_t end_char = 1; end_char <= w->len_char; end_char++) {
        const node_vec *e = &w->ends[end_char];
        for (uint32_t a = 0; a < e->n; a++) {
            const node *r = &e->v[a];
            const node_vec *le = &w->ends[r->start_node];
            for (uint32_t b = 0; b < le->n; b++) { lid[r->left_id] += 1; rid[le->v[b].right_id] += 1; }
        }
    }
    const node_vec *le = &w->ends[w->len_char];
    for (uint32_t b = 0; b < le->n; b++) { lid[0] += 1; rid[le->v[b].right_id] += 1; }
}

/* Shape of the lattice of the last tokenized sentence (statistics for DESIGN.md's sizing of the sweep kernel; not part of the
 * reference): per end position the number of inserted nodes (np_out[i] = ends[i].n, i in 0..len_char) and per start_node the
 * number of nodes inserted from it (nc_out[i]).  Returns len_char. */
ORA_API uint32_t ora_worker_lattice_shape(const ora_worker *w, uint32_t *np_out, uint32_t *nc_out) {
    for (uint32_t i = 0; i <= w->len_char; i++) { np_out[i] = w->ends[i].n; nc_out[i] = 0; }
    for (uint32_t e = 1; e <= w->len_char; e++)
        for (uint32_t a = 0; a < w->ends[e].n; a++) nc_out[w->ends[e].v[a].start_node]++;
    return w->len_char;
}

/* Where the gathered matrix cells fall (statistics for DESIGN.md 3.6, the "hot corner in LDS" question; not part of the
 * reference): for every (node, predecessor) pair search_min_node evaluated in the sentence just tokenized -- the cells a sweep
 * gathers, lattice.rs:137-147 -- m = max(rank_left[node.left_id], rank_right[pred.right_id]) is the smallest k for which the cell
 * lies inside the top-k x top-k corner of a matrix renumbered by those ranks; hist[b] += 1 for the first bound with m < bounds[b]
 * (hist[n_bounds]: beyond all).  EOS (left id 0) against ends[eos.start_node] included. */
ORA_API void ora_worker_add_corner_hist(const ora_worker *w, const uint32_t *rank_left, const uint32_t *rank_right,
                                        const uint32_t *bounds, uint32_t n_bounds, uint64_t *hist) {
    if (w->len_char == 0) return;
    for (uint32_t end_char = 1; end_char <= w->len_char; end_char++) {
        const node_vec *e = &w->ends[end_char];
        for (uint32_t a = 0; a < e->n; a++) {
            const node_vec *le = &w->ends[e->v[a].start_node];
            const uint32_t rl = rank_left[e->v[a].left_id];
            for (uint32_t b = 0; b < le->n; b++) {
                const uint32_t rr = rank_right[le->v[b].right_id], m = rl > rr ? rl : rr;
                uint32_t k = 0;
                while (k < n_bounds && m >= bounds[k]) k++;
                hist[k] += 1;
            }
        }
    }
    const node_vec *le = &w->ends[w->eos.start_node];
    for (uint32_t b = 0; b < le->n; b++) {
        const uint32_t rr = rank_right[le->v[b].right_id], m = rank_left[0] > rr ? rank_left[0] : rr;
        uint32_t k = 0;
        while (k < n_bounds && m >= bounds[k]) k++;
        hist[k] += 1;
    }
}

ORA_API void ora_worker_tokenize(ora_worker *w) { tokenize_impl(w, 0); }
ORA_API void ora_worker_tokenize_counted(ora_worker *w) { tokenize_impl(w, 1); }
ORA_API uint32_t ora_worker_num_tokens(const ora_worker *w) { return w->n_top; }
ORA_API int32_t ora_worker_eos_cost(const ora_worker *w) { return w->eos.min_cost; }
ORA_API void ora_worker_counters(const ora_worker *w, ora_counters *out) { *out = w->cnt; }
ORA_API void ora_worker_reset_counters(ora_worker *w) { memset(&w->cnt, 0, sizeof(w->cnt)); }

/* Worker::token worker.rs:65-68 (index = n-1-i) + Token accessors token.rs:21-92 */
ORA_API void ora_worker_token(const ora_worker *w, uint32_t i, ora_token *out) {
    uint32_t idx = w->n_top - i - 1;
    const node *nd = &w->top_node[idx];
    out->start_char = nd->start_word;
    out->end_char = w->top_end[idx];
    out->start_byte = w->c2b[nd->start_word];
    out->end_byte = w->c2b[w->top_end[idx]];
    out->word_idx = ((uint32_t)nd->lex_type << 30) | nd->word_id;
    out->total_cost = nd->min_cost;
}

ORA_API void ora_worker_token_ids(const ora_worker *w, uint32_t i, int32_t out[2]) {
    const node *nd = &w->top_node[w->n_top - i - 1];
    out[0] = nd->left_id; out[1] = nd->right_id;
}

/* Batch driver used by differential tests and by bench.py's cpu_baseline leg:
 * the 3-call pattern of tokenize/src/main.rs:78-82 over n sentences.
 * tokens: capacity tok_cap records, sentence s occupies [tok_off[s], tok_off[s+1]).
 * Returns total tokens, or (uint64_t)-1 when tok_cap is too small / utf-8 invalid. */
ORA_API uint64_t ora_tokenize_batch(ora_worker *w, const uint8_t *text, const uint64_t *offsets, uint64_t n,
                                    ora_token *tokens, uint64_t tok_cap, uint64_t *tok_off, int counted) {
    uint64_t total = 0;
    for (uint64_t s = 0; s < n; s++) {
        if (!ora_worker_reset_sentence(w, text + offsets[s], (size_t)(offsets[s + 1] - offsets[s]))) return (uint64_t)-1;
        if (counted) tokenize_impl(w, 1); else tokenize_impl(w, 0);
        if (tok_off) tok_off[s] = total;
        if (tokens) {
            if (total + w->n_top > tok_cap) return (uint64_t)-1;
            for (uint32_t i = 0; i < w->n_top; i++) ora_worker_token(w, i, &tokens[total + i]);
        }
        total += w->n_top;
    }
    if (tok_off) tok_off[n] = total;
    return total;
}

/* What `tokenize` does per input line (tokenize/src/main.rs:78-127): tokenize, then print every token in the chosen output mode
 * (0 mecab, 1 wakati, 2 detail).  The cpu_baseline companion of the product's vbt_batch_format: text in, formatted text out, one
 * thread.  Appends to out[0..cap); returns the bytes the whole output takes (more than cap: nothing beyond cap was written), or
 * (uint64_t)-1 on invalid UTF-8. */
ORA_API uint64_t ora_tokenize_format_batch(ora_worker *w, const uint8_t *text, const uint64_t *offsets, uint64_t n, int mode,
                                           char *out, uint64_t cap) {
    static const char *const lex_names[3] = {"System", "User", "Unknown"};
    const ora_dict *d = w->tok->dict;
    uint64_t at = 0;
    char num[160];
#define ORA_PUT(ptr, len_) do { uint64_t l_ = (len_); if (at + l_ <= cap) memcpy(out + at, (ptr), l_); at += l_; } while (0)
    for (uint64_t s = 0; s < n; s++) {
        const uint8_t *sent = text + offsets[s];
        if (!ora_worker_reset_sentence(w, sent, (size_t)(offsets[s + 1] - offsets[s]))) return (uint64_t)-1;
        tokenize_impl(w, 0);
        for (uint32_t i = 0; i < w->n_top; i++) {
            ora_token t;
            ora_worker_token(w, i, &t);
            const node *nd = &w->top_node[w->n_top - i - 1];
            if (mode == 1 && i) ORA_PUT(" ", 1);
            ORA_PUT(sent + t.start_byte, t.end_byte - t.start_byte);
            if (mode == 1) continue;
            uint32_t flen;
            const char *f = ora_dict_word_feature(d, nd->lex_type, nd->word_id, &flen);
            ORA_PUT("\t", 1);
            ORA_PUT(f, flen);
            if (mode == 2) {
                int32_t prm[3];
                ora_dict_word_param(d, nd->lex_type, nd->word_id, prm);
                int k = snprintf(num, sizeof(num), "\tlex_type=%s\tleft_id=%u\tright_id=%u\tword_cost=%d\ttotal_cost=%d", lex_names[nd->lex_type],
                                 (unsigned)nd->left_id, (unsigned)nd->right_id, (int)prm[2], (int)nd->min_cost);
                ORA_PUT(num, (uint64_t)k);
            }
            ORA_PUT("\n", 1);
        }
        if (mode == 1) ORA_PUT("\n", 1); else ORA_PUT("EOS\n", 4);
    }
#undef ORA_PUT
    return at;
}

/* common-prefix enumeration for the lexicon tests (tests/lexicon.rs:8-57,
 * lexicon.rs:232-272): input = code points. out rows: word_id,end_char,left,right,cost */
ORA_API uint32_t ora_dict_common_prefix(const ora_dict *d, int lex_type, const uint32_t *cps, uint32_t n,
                                        int32_t *out, uint32_t out_cap) {
    const lexicon *lx = lex_type == 0 ? &d->sys : &d->user;
    const da_trie *t = &lx->trie;
    uint32_t node_idx = 0, m = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (cps[i] >= t->mapper_len) break;
        uint32_t mc = t->mapper[cps[i]];
        if (mc == DA_INVALID) break;
        if (t->base[node_idx] >> 31) break;
        uint32_t child = (t->base[node_idx] & DA_MASK) ^ mc;
        if ((t->check[child] & DA_MASK) != node_idx) break;
        node_idx = child;
        if (t->check[node_idx] >> 31) {
            uint32_t off = t->base[t->base[node_idx] & DA_MASK] & DA_MASK;
            for (uint32_t k = 0; k < lx->postings[off]; k++) {
                uint32_t wid = lx->postings[off + 1 + k];
                if (m < out_cap) {
                    out[m * 5 + 0] = (int32_t)wid; out[m * 5 + 1] = (int32_t)(i + 1);
                    out[m * 5 + 2] = lx->params[wid].left_id; out[m * 5 + 3] = lx->params[wid].right_id;
                    out[m * 5 + 4] = lx->params[wid].word_cost;
                }
                m++;
            }
        }
    }
    return m;
}

ORA_API uint32_t ora_dict_trie_nodes(const ora_dict *d, int lex_type) { return lex_type == 0 ? d->sys.trie.n_nodes : d->user.trie.n_nodes; }
