/*
 * raw_connector.c -- CPU ORACLE (test infrastructure only, see vibrato_oracle.c) for the compact connectors:
 * a plain-C restatement of vibrato's RawConnector and of the Scorer it is built on.  Included by vibrato_oracle.c.
 *
 * Follows (paths relative to /root/reference/vibrato/src/dictionary/connector):
 *   raw_connector.rs:45-105   RawConnector::from_readers (template size rounded up to 8, row 0 = BOS/EOS zeros, padding
 *                             with INVALID_FEATURE_ID)
 *   raw_connector.rs:153-161  ConnectorCost::cost = Scorer::accumulate_cost(right row, left row)
 *   raw_connector.rs:186-330  RawConnectorBuilder::from_readers / parse_features / parse_cost
 *   raw_connector/scorer.rs:103-168  ScorerBuilder::insert / build
 *   raw_connector/scorer.rs:257-282  Scorer::retrieve_cost / accumulate_cost (scalar; the AVX2 path of l.284-345 adds the
 *                             same eight lanes)
 *   (DualConnector: dual_connector.c, built on the RawConnectorBuilder below.)
 *   ../../utils.rs:40-61      parse_csv_row (csv_core defaults)
 * Parity: pinned by the reference's unit vectors (scorer.rs:407-512, raw_connector.rs:331-520, dual_connector.rs:287-384),
 * transcribed by tests/golden/make_golden.py into tests/golden/unit_golden.json.
 */

#define RAW_INVALID_FEATURE 0x7FFFFFFFu /* INVALID_FEATURE_ID = U31::MAX, raw_connector.rs:17-19 */
#define RAW_UNUSED_CHECK 0xFFFFFFFFu    /* scorer.rs:15 */
#define RAW_SIMD 8u                     /* scorer.rs:17 */

typedef struct { uint32_t key2; int32_t cost; } sc_pair;
typedef struct { sc_pair *v; uint32_t n, cap; } sc_row; /* one BTreeMap<U31, i32>: kept sorted by key2 */

typedef struct {
    uint32_t *bases; uint32_t n_bases;
    uint32_t *checks; int32_t *costs; uint32_t n_checks;
} ora_scorer;

typedef struct {
    sc_row *rows; uint32_t n_rows, cap_rows;
} ora_scorer_builder;

static void scb_insert(ora_scorer_builder *b, uint32_t key1, uint32_t key2, int32_t cost) { /* scorer.rs:113-119 */
    if (key1 >= b->n_rows) {
        if (key1 >= b->cap_rows) {
            uint32_t nc = b->cap_rows ? b->cap_rows : 16;
            while (nc <= key1) nc *= 2;
            b->rows = (sc_row *)realloc(b->rows, sizeof(sc_row) * nc);
            memset(b->rows + b->cap_rows, 0, sizeof(sc_row) * (nc - b->cap_rows));
            b->cap_rows = nc;
        }
        b->n_rows = key1 + 1;
    }
    sc_row *r = &b->rows[key1];
    uint32_t i = 0;
    while (i < r->n && r->v[i].key2 < key2) i++;
    if (i < r->n && r->v[i].key2 == key2) { r->v[i].cost = cost; return; } /* BTreeMap::insert overwrites */
    if (r->n == r->cap) { r->cap = r->cap ? r->cap * 2 : 4; r->v = (sc_pair *)realloc(r->v, sizeof(sc_pair) * r->cap); }
    memmove(r->v + i + 1, r->v + i, sizeof(sc_pair) * (r->n - i));
    r->v[i].key2 = key2; r->v[i].cost = cost; r->n++;
}

static void scb_free(ora_scorer_builder *b) {
    for (uint32_t i = 0; i < b->cap_rows; i++) free(b->rows[i].v);
    free(b->rows);
    memset(b, 0, sizeof(*b));
}

static void scorer_free(ora_scorer *s) { free(s->bases); free(s->checks); free(s->costs); memset(s, 0, sizeof(*s)); }

static void scorer_build(const ora_scorer_builder *b, ora_scorer *s) { /* scorer.rs:131-168 */
    memset(s, 0, sizeof(*s));
    s->n_bases = b->n_rows;
    s->bases = (uint32_t *)calloc(b->n_rows ? b->n_rows : 1, 4);
    uint32_t cap = 0;
    for (uint32_t key1 = 0; key1 < b->n_rows; key1++) {
        const sc_row *r = &b->rows[key1];
        uint32_t base = 0;
        for (;; base++) { /* check_base, scorer.rs:121-130 */
            int ok = 1;
            for (uint32_t i = 0; i < r->n; i++) {
                uint32_t pos = base ^ r->v[i].key2;
                if (pos < s->n_checks && s->checks[pos] != RAW_UNUSED_CHECK) { ok = 0; break; }
            }
            if (ok) break;
        }
        s->bases[key1] = base;
        for (uint32_t i = 0; i < r->n; i++) {
            uint32_t pos = base ^ r->v[i].key2;
            if (pos >= s->n_checks) {
                if (pos >= cap) {
                    uint32_t nc = cap ? cap : 64;
                    while (nc <= pos) nc *= 2;
                    s->checks = (uint32_t *)realloc(s->checks, 4 * (size_t)nc);
                    s->costs = (int32_t *)realloc(s->costs, 4 * (size_t)nc);
                    cap = nc;
                }
                for (uint32_t q = s->n_checks; q <= pos; q++) { s->checks[q] = RAW_UNUSED_CHECK; s->costs[q] = 0; }
                s->n_checks = pos + 1;
            }
            s->checks[pos] = key1;
            s->costs[pos] = r->v[i].cost;
        }
    }
}

/* Scorer::retrieve_cost, scorer.rs:257-270: 1 and *out when the pair is present */
static int scorer_retrieve(const ora_scorer *s, uint32_t key1, uint32_t key2, int32_t *out) {
    if (key1 < s->n_bases) {
        uint32_t pos = s->bases[key1] ^ key2;
        if (pos < s->n_checks && s->checks[pos] == key1) { *out = s->costs[pos]; return 1; }
    }
    return 0;
}

/* Scorer::accumulate_cost, scorer.rs:272-282 */
static int32_t scorer_accumulate(const ora_scorer *s, const uint32_t *keys1, const uint32_t *keys2, uint32_t n) {
    uint32_t score = 0; /* i32 `+=` (wrapping in a release build) */
    for (uint32_t i = 0; i < n; i++) {
        int32_t w;
        if (scorer_retrieve(s, keys1[i], keys2[i], &w)) score += (uint32_t)w;
    }
    return (int32_t)score;
}

/* ---- string -> id map (HashMap<String, U31>, insertion order = id) ---- */
typedef struct { char **keys; uint32_t *lens; uint32_t *slots; uint32_t n, cap_keys, cap_slots; } str_map;

static uint64_t sm_hash(const char *s, uint32_t n) {
    uint64_t h = 1469598103934665603ULL;
    for (uint32_t i = 0; i < n; i++) { h ^= (uint8_t)s[i]; h *= 1099511628211ULL; }
    return h;
}
static void sm_rehash(str_map *m, uint32_t cap) {
    free(m->slots);
    m->slots = (uint32_t *)malloc(4 * (size_t)cap);
    memset(m->slots, 0xFF, 4 * (size_t)cap);
    m->cap_slots = cap;
    for (uint32_t i = 0; i < m->n; i++) {
        uint32_t p = (uint32_t)(sm_hash(m->keys[i], m->lens[i]) & (cap - 1));
        while (m->slots[p] != 0xFFFFFFFFu) p = (p + 1) & (cap - 1);
        m->slots[p] = i;
    }
}
/* id of the key, or 0xFFFFFFFF when absent (insert = 0) / its new id (insert = 1) */
static uint32_t sm_get(str_map *m, const char *s, uint32_t n, int insert) {
    if (m->cap_slots == 0) sm_rehash(m, 64);
    uint32_t p = (uint32_t)(sm_hash(s, n) & (m->cap_slots - 1));
    while (m->slots[p] != 0xFFFFFFFFu) {
        uint32_t i = m->slots[p];
        if (m->lens[i] == n && memcmp(m->keys[i], s, n) == 0) return i;
        p = (p + 1) & (m->cap_slots - 1);
    }
    if (!insert) return 0xFFFFFFFFu;
    if (m->n == m->cap_keys) {
        m->cap_keys = m->cap_keys ? m->cap_keys * 2 : 64;
        m->keys = (char **)realloc(m->keys, sizeof(char *) * m->cap_keys);
        m->lens = (uint32_t *)realloc(m->lens, 4 * (size_t)m->cap_keys);
    }
    m->keys[m->n] = (char *)malloc(n ? n : 1);
    memcpy(m->keys[m->n], s, n);
    m->lens[m->n] = n;
    m->slots[p] = m->n;
    m->n++;
    if (m->n * 2 > m->cap_slots) sm_rehash(m, m->cap_slots * 2);
    return m->n - 1;
}
static void sm_free(str_map *m) {
    for (uint32_t i = 0; i < m->n; i++) free(m->keys[i]);
    free(m->keys); free(m->lens); free(m->slots);
    memset(m, 0, sizeof(*m));
}

/* utils::parse_csv_row (utils.rs:40-61) on one row: calls emit(field, len) per field */
typedef void (*csv_field_fn)(void *ctx, const char *f, uint32_t n);
static void csv_row_fields(const char *row, size_t len, csv_field_fn emit, void *ctx) {
    bytebuf field = {0};
    size_t pos = 0;
    for (;;) {
        field.len = 0;
        if (pos < len && row[pos] == '"') {
            pos++;
            while (pos < len) {
                if (row[pos] == '"') {
                    if (pos + 1 < len && row[pos + 1] == '"') { bb_push(&field, "\"", 1); pos += 2; continue; }
                    pos++;
                    break;
                }
                bb_push(&field, row + pos, 1);
                pos++;
            }
            while (pos < len && row[pos] != ',') { bb_push(&field, row + pos, 1); pos++; }
        } else {
            while (pos < len && row[pos] != ',') { bb_push(&field, row + pos, 1); pos++; }
        }
        emit(ctx, field.p ? field.p : "", (uint32_t)field.len);
        if (pos >= len) break;
        pos++;
        if (pos == len) { emit(ctx, "", 0); break; }
    }
    free(field.p);
}

typedef struct { uint32_t *v; uint32_t n, cap; } u32_vec;
typedef struct { str_map *ids; u32_vec *out; } feat_ctx;
static void feat_emit(void *ctx_, const char *f, uint32_t n) {
    feat_ctx *ctx = (feat_ctx *)ctx_;
    uint32_t id = sm_get(ctx->ids, f, n, 0);
    u32_vec *o = ctx->out;
    if (o->n == o->cap) { o->cap = o->cap ? o->cap * 2 : 16; o->v = (uint32_t *)realloc(o->v, 4 * (size_t)o->cap); }
    o->v[o->n++] = id == 0xFFFFFFFFu ? RAW_INVALID_FEATURE : id;
}

typedef struct ora_raw_connector {
    uint32_t *right_feats, *left_feats; /* (num + 1) x width feature ids */
    uint32_t width, num_right, num_left;
    ora_scorer scorer;
} ora_raw_connector;

static void raw_free(ora_raw_connector *c) {
    if (!c) return;
    free(c->right_feats); free(c->left_feats);
    scorer_free(&c->scorer);
    free(c);
}

/* lines of a reader the way BufRead::lines() yields them */
static size_t raw_next_line(const char *buf, size_t len, size_t pos, size_t *ls, size_t *le) {
    size_t e = pos;
    while (e < len && buf[e] != '\n') e++;
    *ls = pos;
    *le = (e < len && e > pos && buf[e - 1] == '\r') ? e - 1 : e; /* '\r' goes only as part of "\r\n" */
    return e < len ? e + 1 : len;
}

static int raw_parse_rows(const char *buf, size_t len, str_map *ids, const char *name, u32_vec **rows_out, uint32_t *n_rows,
                          uint32_t *template_size, char *err, size_t errcap) {
    u32_vec *rows = NULL;
    uint32_t n = 0, cap = 0;
    size_t pos = 0;
    while (pos < len) {
        size_t ls, le;
        pos = raw_next_line(buf, len, pos, &ls, &le);
        const char *line = buf + ls;
        size_t ll = le - ls;
        const char *tab = (const char *)memchr(line, '\t', ll);
        if (!tab || memchr(tab + 1, '\t', (size_t)(line + ll - tab - 1))) { set_err(err, errcap, "%s: The format must be id<tab>csv_row", name); goto fail; }
        long long id;
        if (!parse_int(line, (size_t)(tab - line), 0, 0, 0x7FFFFFFF, &id)) { set_err(err, errcap, "%s: invalid id", name); goto fail; }
        if ((uint64_t)id != (uint64_t)n + 1) { set_err(err, errcap, "%s: must be ascending order", name); goto fail; }
        if (n == cap) { cap = cap ? cap * 2 : 64; rows = (u32_vec *)realloc(rows, sizeof(u32_vec) * cap); }
        memset(&rows[n], 0, sizeof(u32_vec));
        feat_ctx ctx = {ids, &rows[n]};
        csv_row_fields(tab + 1, (size_t)(line + ll - tab - 1), feat_emit, &ctx);
        if (rows[n].n > *template_size) *template_size = rows[n].n;
        n++;
    }
    *rows_out = rows; *n_rows = n;
    return 1;
fail:
    for (uint32_t i = 0; i < n; i++) free(rows[i].v);
    free(rows);
    return 0;
}

static uint32_t *raw_feature_matrix(const u32_vec *rows, uint32_t n, uint32_t width) {
    size_t total = ((size_t)n + 1) * width;
    uint32_t *m = (uint32_t *)malloc(4 * (total ? total : 1));
    for (size_t i = 0; i < total; i++) m[i] = i < width ? 0u : RAW_INVALID_FEATURE;
    for (uint32_t i = 0; i < n; i++) memcpy(m + ((size_t)i + 1) * width, rows[i].v, 4 * (size_t)rows[i].n);
    return m;
}

/* RawConnectorBuilder (raw_connector.rs:170-253): what from_readers leaves for RawConnector / DualConnector to finish */
typedef struct {
    ora_scorer_builder sb;
    u32_vec *rrows, *lrows; /* right_feat_ids_tmp / left_feat_ids_tmp: one feature-id vector per line */
    uint32_t nr, nl, tsize; /* rows, feat_template_size (longest row, not rounded) */
} raw_builder;

static void raw_builder_free(raw_builder *b) {
    for (uint32_t i = 0; i < b->nr; i++) free(b->rrows[i].v);
    for (uint32_t i = 0; i < b->nl; i++) free(b->lrows[i].v);
    free(b->rrows); free(b->lrows);
    scb_free(&b->sb);
    memset(b, 0, sizeof(*b));
}

/* RawConnectorBuilder::from_readers, raw_connector.rs:186-245 */
static int raw_builder_from_sources(raw_builder *b, const char *right, size_t right_len, const char *left, size_t left_len, const char *cost,
                                    size_t cost_len, char *err, size_t errcap) {
    str_map rids = {0}, lids = {0};
    int ok = 0;
    memset(b, 0, sizeof(*b));
    sm_get(&rids, "", 0, 1); /* raw_connector.rs:193-196 */
    sm_get(&lids, "", 0, 1);
    size_t pos = 0;
    while (pos < cost_len) { /* parse_cost, raw_connector.rs:294-325 */
        size_t ls, le;
        pos = raw_next_line(cost, cost_len, pos, &ls, &le);
        const char *line = cost + ls;
        size_t ll = le - ls;
        const char *tab = (const char *)memchr(line, '\t', ll);
        if (!tab || memchr(tab + 1, '\t', (size_t)(line + ll - tab - 1))) { set_err(err, errcap, "bigram.cost: The format must be right/left<tab>cost"); goto done; }
        long long cv;
        if (!parse_int(tab + 1, (size_t)(line + ll - tab - 1), 1, -2147483648LL, 2147483647LL, &cv)) { set_err(err, errcap, "bigram.cost: invalid cost"); goto done; }
        const char *slash = (const char *)memchr(line, '/', (size_t)(tab - line));
        if (!slash || memchr(slash + 1, '/', (size_t)(tab - slash - 1))) { set_err(err, errcap, "bigram.cost: The format must be right/left<tab>cost"); goto done; }
        uint32_t rid = sm_get(&rids, line, (uint32_t)(slash - line), 1);
        uint32_t lid = sm_get(&lids, slash + 1, (uint32_t)(tab - slash - 1), 1);
        scb_insert(&b->sb, rid, lid, (int32_t)cv);
    }
    if (!raw_parse_rows(right, right_len, &rids, "bigram.right", &b->rrows, &b->nr, &b->tsize, err, errcap)) goto done;
    if (!raw_parse_rows(left, left_len, &lids, "bigram.left", &b->lrows, &b->nl, &b->tsize, err, errcap)) goto done;
    ok = 1;
done:
    sm_free(&rids); sm_free(&lids);
    if (!ok) raw_builder_free(b);
    return ok;
}

/* RawConnector::from_readers, raw_connector.rs:45-105 */
static ora_raw_connector *raw_from_sources(const char *right, size_t right_len, const char *left, size_t left_len, const char *cost,
                                           size_t cost_len, char *err, size_t errcap) {
    raw_builder b;
    if (!raw_builder_from_sources(&b, right, right_len, left, left_len, cost, cost_len, err, errcap)) return NULL;
    uint32_t tsize = b.tsize;
    if (tsize) tsize = ((tsize - 1) / RAW_SIMD + 1) * RAW_SIMD; /* raw_connector.rs:58-60 */
    ora_raw_connector *c = (ora_raw_connector *)calloc(1, sizeof(*c));
    c->width = tsize; c->num_right = b.nr + 1; c->num_left = b.nl + 1;
    c->right_feats = raw_feature_matrix(b.rrows, b.nr, tsize);
    c->left_feats = raw_feature_matrix(b.lrows, b.nl, tsize);
    scorer_build(&b.sb, &c->scorer);
    raw_builder_free(&b);
    return c;
}

/* ConnectorCost::cost(right_id, left_id), raw_connector.rs:153-161 */
static int32_t raw_cost(const ora_raw_connector *c, uint32_t right_id, uint32_t left_id) {
    return scorer_accumulate(&c->scorer, c->right_feats + (size_t)right_id * c->width, c->left_feats + (size_t)left_id * c->width, c->width);
}

/* Connector::map_connection_ids, raw_connector.rs:118-146: new row = old row of the id mapped to it */
static void raw_map_ids(ora_raw_connector *c, const uint16_t *ml, const uint16_t *mr) {
    uint32_t *nr = (uint32_t *)malloc(4 * (size_t)c->num_right * c->width + 4), *nl = (uint32_t *)malloc(4 * (size_t)c->num_left * c->width + 4);
    for (uint32_t r = 0; r < c->num_right; r++) memcpy(nr + (size_t)mr[r] * c->width, c->right_feats + (size_t)r * c->width, 4 * (size_t)c->width);
    for (uint32_t l = 0; l < c->num_left; l++) memcpy(nl + (size_t)ml[l] * c->width, c->left_feats + (size_t)l * c->width, 4 * (size_t)c->width);
    free(c->right_feats); free(c->left_feats);
    c->right_feats = nr; c->left_feats = nl;
}

/* ---- test entry points for the unit vectors (scorer.rs:407-512) ---- */
ORA_API void *ora_scorer_new(const uint32_t *triples, uint32_t n) { /* triples: key1, key2, cost */
    ora_scorer_builder sb = {0};
    for (uint32_t i = 0; i < n; i++) scb_insert(&sb, triples[3 * i], triples[3 * i + 1], (int32_t)triples[3 * i + 2]);
    ora_scorer *s = (ora_scorer *)calloc(1, sizeof(*s));
    scorer_build(&sb, s);
    scb_free(&sb);
    return s;
}
ORA_API void ora_scorer_free(void *s) { if (s) { scorer_free((ora_scorer *)s); free(s); } }
ORA_API int ora_scorer_retrieve(const void *s, uint32_t key1, uint32_t key2, int32_t *out) { return scorer_retrieve((const ora_scorer *)s, key1, key2, out); }
ORA_API int32_t ora_scorer_accumulate(const void *s, const uint32_t *k1, const uint32_t *k2, uint32_t n) { return scorer_accumulate((const ora_scorer *)s, k1, k2, n); }
ORA_API void *ora_raw_connector_new(const char *right, size_t rl, const char *left, size_t ll, const char *cost, size_t cl, char *err, size_t errcap) {
    return raw_from_sources(right, rl, left, ll, cost, cl, err, errcap);
}
ORA_API void ora_raw_connector_free(void *c) { raw_free((ora_raw_connector *)c); }
ORA_API int32_t ora_raw_connector_cost(const void *c, uint32_t right_id, uint32_t left_id) { return raw_cost((const ora_raw_connector *)c, right_id, left_id); }
ORA_API uint32_t ora_raw_connector_num(const void *c, int left) { return left ? ((const ora_raw_connector *)c)->num_left : ((const ora_raw_connector *)c)->num_right; }
ORA_API void ora_raw_connector_map(void *c, const uint16_t *ml, const uint16_t *mr) { raw_map_ids((ora_raw_connector *)c, ml, mr); }
