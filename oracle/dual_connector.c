/*
 * dual_connector.c -- CPU ORACLE (test infrastructure only, see vibrato_oracle.c): a plain-C restatement of vibrato's
 * DualConnector.  Included by vibrato_oracle.c behind raw_connector.c (it finishes the same RawConnectorBuilder).
 *
 * Follows /root/reference/vibrato/src/dictionary/connector/dual_connector.rs:
 *   l.25-70    remove_feature_templates_greedy: eight rounds; a round tries every remaining template index and keeps the one
 *              whose removal leaves the smallest (distinct right rows) x (distinct left rows), `<=` so the LAST tried of several
 *              equally good ones wins.  The reference walks a HashSet there -- its order differs from run to run -- so "last" is
 *              not defined by the reference; this restatement tries the indices in ascending order (the highest index among
 *              equals wins), which is also what the product does.  The choice moves cost between the matrix and the raw part.
 *   l.72-110   create_matrix_connector: classes of the kept templates per side (class 0 = the all-zero row, the rows of BOS/EOS),
 *              cell = accumulate_cost over the class rows padded to a multiple of 8 with feature 0 (U31x8::to_simd_vec), CLAMPED
 *              to i16 (l.103)
 *   l.112-143  create_raw_connector: row 0 = eight zeros, then per line its eight removed templates (INVALID where the line is
 *              shorter); the scorer is pruned to the pairs those rows can ask for (same costs)
 *   l.145-199  from_readers
 *   l.211-264  map_connection_ids: rows and id maps permuted, the small matrix renumbered by first use
 *   l.267-279  cost = matrix(left class, right class) + accumulate_cost(raw right row, raw left row): i32, never clamped
 * Parity: pinned by the reference's own vectors (dual_connector.rs:284-326 from_readers_test, 328-384 mapping_test; transcribed
 * by tests/golden/make_golden.py into tests/golden/unit_golden.json: "dual" cases).
 */

typedef struct ora_dual_connector {
    int16_t *matrix; /* [m_num_left][m_num_right] */
    uint32_t m_num_right, m_num_left;
    uint16_t *right_map, *left_map; /* connection id -> class of its kept templates */
    uint32_t *right_feats, *left_feats; /* num x 8 */
    ora_scorer scorer;
    uint32_t num_right, num_left;
} ora_dual_connector;

static void dual_free(ora_dual_connector *c) {
    if (!c) return;
    free(c->matrix); free(c->right_map); free(c->left_map); free(c->right_feats); free(c->left_feats);
    scorer_free(&c->scorer);
    free(c);
}

/* a row restricted to the template indices in `keep` (ascending); only the positions the row has (row.get(i) is Some) */
static uint32_t dual_restrict(const u32_vec *row, const uint32_t *keep, uint32_t n_keep, uint32_t without, uint32_t *out) {
    uint32_t n = 0;
    for (uint32_t k = 0; k < n_keep; k++)
        if (keep[k] != without && keep[k] < row->n) out[n++] = row->v[keep[k]];
    return n;
}

static uint32_t g_dual_width; /* comparator context: vectors are stored as {len, ids..., padding} of g_dual_width + 1 words */
static int dual_cmp(const void *a, const void *b) { return memcmp(a, b, 4 * ((size_t)g_dual_width + 1)); }

static uint32_t dual_distinct(const u32_vec *rows, uint32_t n_rows, const uint32_t *keep, uint32_t n_keep, uint32_t without) {
    if (n_rows == 0) return 0;
    const size_t w = (size_t)n_keep + 1;
    uint32_t *buf = (uint32_t *)calloc(n_rows * w, 4);
    for (uint32_t r = 0; r < n_rows; r++) buf[r * w] = dual_restrict(&rows[r], keep, n_keep, without, buf + r * w + 1);
    g_dual_width = n_keep;
    qsort(buf, n_rows, 4 * w, dual_cmp);
    uint32_t d = 1;
    for (uint32_t r = 1; r < n_rows; r++) d += memcmp(buf + (r - 1) * w, buf + r * w, 4 * w) != 0;
    free(buf);
    return d;
}

/* generate_feature_map (l.79-93): class ids in order of first appearance, class 0 = zeros; returns the class rows (n_cls x n_keep) */
static uint32_t *dual_classes(const u32_vec *rows, uint32_t n_rows, const uint32_t *keep, uint32_t n_keep, uint16_t *id_map, uint32_t *n_cls_out) {
    uint32_t cap = 16, n_cls = 1;
    uint32_t *cls = (uint32_t *)calloc((size_t)cap * (n_keep ? n_keep : 1), 4);
    uint32_t *tmp = (uint32_t *)malloc(4 * ((size_t)n_keep + 1));
    id_map[0] = 0;
    for (uint32_t r = 0; r < n_rows; r++) {
        for (uint32_t k = 0; k < n_keep; k++) tmp[k] = keep[k] < rows[r].n ? rows[r].v[keep[k]] : RAW_INVALID_FEATURE;
        uint32_t c = 0;
        for (; c < n_cls; c++)
            if (memcmp(cls + (size_t)c * n_keep, tmp, 4 * (size_t)n_keep) == 0) break;
        if (c == n_cls) {
            if (n_cls == cap) { cap *= 2; cls = (uint32_t *)realloc(cls, 4 * (size_t)cap * (n_keep ? n_keep : 1)); }
            memcpy(cls + (size_t)c * n_keep, tmp, 4 * (size_t)n_keep);
            n_cls++;
        }
        id_map[r + 1] = (uint16_t)c;
    }
    free(tmp);
    *n_cls_out = n_cls;
    return cls;
}

/* DualConnector::from_readers, dual_connector.rs:145-199 */
static ora_dual_connector *dual_from_sources(const char *right, size_t right_len, const char *left, size_t left_len, const char *cost,
                                             size_t cost_len, char *err, size_t errcap) {
    raw_builder b;
    if (!raw_builder_from_sources(&b, right, right_len, left, left_len, cost, cost_len, err, errcap)) return NULL;
    const uint32_t T = b.tsize;
    if (T < RAW_SIMD) { /* the reference computes feat_template_size - SIMD_SIZE in usize here (l.82) */
        set_err(err, errcap, "bigram: a dual connector needs at least eight feature templates");
        raw_builder_free(&b);
        return NULL;
    }
    ora_scorer full;
    scorer_build(&b.sb, &full); /* l.157 */
    /* remove_feature_templates_greedy(SIMD_SIZE, ...), l.25-70 */
    uint32_t *keep = (uint32_t *)malloc(4 * (size_t)T), n_keep = T;
    for (uint32_t i = 0; i < T; i++) keep[i] = i;
    for (uint32_t round = 0; round < RAW_SIMD; round++) {
        uint32_t candidate = 0;
        uint64_t best = (uint64_t)b.nl * b.nr;
        for (uint32_t k = 0; k < n_keep; k++) {
            const uint64_t size = (uint64_t)dual_distinct(b.rrows, b.nr, keep, n_keep, keep[k]) * dual_distinct(b.lrows, b.nl, keep, n_keep, keep[k]);
            if (size <= best) { best = size; candidate = keep[k]; }
        }
        uint32_t n = 0;
        for (uint32_t k = 0; k < n_keep; k++)
            if (keep[k] != candidate) keep[n++] = keep[k];
        n_keep = n;
    }
    uint32_t raw_idx[RAW_SIMD], n_raw = 0;
    for (uint32_t i = 0, k = 0; i < T; i++) {
        if (k < n_keep && keep[k] == i) { k++; continue; }
        if (n_raw < RAW_SIMD) raw_idx[n_raw++] = i;
    }
    ora_dual_connector *c = (ora_dual_connector *)calloc(1, sizeof(*c));
    c->num_right = b.nr + 1; c->num_left = b.nl + 1;
    c->right_map = (uint16_t *)calloc(c->num_right, 2);
    c->left_map = (uint16_t *)calloc(c->num_left, 2);
    /* create_matrix_connector, l.72-110 */
    uint32_t n_rc, n_lc;
    uint32_t *rc = dual_classes(b.rrows, b.nr, keep, n_keep, c->right_map, &n_rc);
    uint32_t *lc = dual_classes(b.lrows, b.nl, keep, n_keep, c->left_map, &n_lc);
    const uint32_t padded = n_keep ? ((n_keep - 1) / RAW_SIMD + 1) * RAW_SIMD : 0; /* U31x8::to_simd_vec: rows padded with feature 0 */
    uint32_t *ra = (uint32_t *)calloc(padded ? padded : 1, 4), *la = (uint32_t *)calloc(padded ? padded : 1, 4);
    c->m_num_right = n_rc; c->m_num_left = n_lc;
    c->matrix = (int16_t *)calloc((size_t)n_rc * n_lc, 2);
    for (uint32_t r = 0; r < n_rc; r++) {
        memcpy(ra, rc + (size_t)r * n_keep, 4 * (size_t)n_keep);
        for (uint32_t l = 0; l < n_lc; l++) {
            memcpy(la, lc + (size_t)l * n_keep, 4 * (size_t)n_keep);
            int32_t v = scorer_accumulate(&full, ra, la, padded);
            if (v < -32768) v = -32768; /* l.103 */
            if (v > 32767) v = 32767;
            c->matrix[(size_t)l * n_rc + r] = (int16_t)v;
        }
    }
    free(ra); free(la); free(rc); free(lc);
    /* create_raw_connector, l.112-143 */
    c->right_feats = (uint32_t *)calloc((size_t)c->num_right * RAW_SIMD, 4);
    c->left_feats = (uint32_t *)calloc((size_t)c->num_left * RAW_SIMD, 4);
    for (uint32_t r = 0; r < b.nr; r++)
        for (uint32_t k = 0; k < RAW_SIMD; k++)
            c->right_feats[((size_t)r + 1) * RAW_SIMD + k] = k < n_raw && raw_idx[k] < b.rrows[r].n ? b.rrows[r].v[raw_idx[k]] : RAW_INVALID_FEATURE;
    for (uint32_t l = 0; l < b.nl; l++)
        for (uint32_t k = 0; k < RAW_SIMD; k++)
            c->left_feats[((size_t)l + 1) * RAW_SIMD + k] = k < n_raw && raw_idx[k] < b.lrows[l].n ? b.lrows[l].v[raw_idx[k]] : RAW_INVALID_FEATURE;
    {   /* the scorer of the raw part keeps only the pairs its rows can ask for (l.128-141) */
        uint32_t max_r = 0, max_l = 0;
        for (size_t i = 0; i < (size_t)c->num_right * RAW_SIMD; i++) if (c->right_feats[i] != RAW_INVALID_FEATURE && c->right_feats[i] > max_r) max_r = c->right_feats[i];
        for (size_t i = 0; i < (size_t)c->num_left * RAW_SIMD; i++) if (c->left_feats[i] != RAW_INVALID_FEATURE && c->left_feats[i] > max_l) max_l = c->left_feats[i];
        uint8_t *ru = (uint8_t *)calloc((size_t)max_r + 1, 1), *lu = (uint8_t *)calloc((size_t)max_l + 1, 1);
        for (size_t i = 0; i < (size_t)c->num_right * RAW_SIMD; i++) if (c->right_feats[i] != RAW_INVALID_FEATURE) ru[c->right_feats[i]] = 1;
        for (size_t i = 0; i < (size_t)c->num_left * RAW_SIMD; i++) if (c->left_feats[i] != RAW_INVALID_FEATURE) lu[c->left_feats[i]] = 1;
        for (uint32_t k1 = 0; k1 < b.sb.n_rows; k1++) {
            sc_row *row = &b.sb.rows[k1];
            if (k1 > max_r || !ru[k1]) { row->n = 0; continue; }
            uint32_t n = 0;
            for (uint32_t i = 0; i < row->n; i++)
                if (row->v[i].key2 <= max_l && lu[row->v[i].key2]) row->v[n++] = row->v[i];
            row->n = n;
        }
        free(ru); free(lu);
        scorer_build(&b.sb, &c->scorer);
    }
    scorer_free(&full);
    free(keep);
    raw_builder_free(&b);
    return c;
}

/* ConnectorCost::cost(right_id, left_id), dual_connector.rs:267-279 */
static int32_t dual_cost(const ora_dual_connector *c, uint32_t right_id, uint32_t left_id) {
    const int32_t m = (int32_t)c->matrix[(size_t)c->left_map[left_id] * c->m_num_right + c->right_map[right_id]];
    const int32_t raw = scorer_accumulate(&c->scorer, c->right_feats + (size_t)right_id * RAW_SIMD, c->left_feats + (size_t)left_id * RAW_SIMD, RAW_SIMD);
    return (int32_t)((uint32_t)m + (uint32_t)raw);
}

/* Connector::map_connection_ids, dual_connector.rs:211-264 (ml / mr: old id -> new id) */
static void dual_map_ids(ora_dual_connector *c, const uint16_t *ml, const uint16_t *mr) {
    uint32_t *nrf = (uint32_t *)calloc((size_t)c->num_right * RAW_SIMD, 4), *nlf = (uint32_t *)calloc((size_t)c->num_left * RAW_SIMD, 4);
    uint16_t *nrm = (uint16_t *)calloc(c->num_right, 2), *nlm = (uint16_t *)calloc(c->num_left, 2);
    for (uint32_t r = 0; r < c->num_right; r++) { memcpy(nrf + (size_t)mr[r] * RAW_SIMD, c->right_feats + (size_t)r * RAW_SIMD, 4 * RAW_SIMD); nrm[mr[r]] = c->right_map[r]; }
    for (uint32_t l = 0; l < c->num_left; l++) { memcpy(nlf + (size_t)ml[l] * RAW_SIMD, c->left_feats + (size_t)l * RAW_SIMD, 4 * RAW_SIMD); nlm[ml[l]] = c->left_map[l]; }
    free(c->right_feats); free(c->left_feats); free(c->right_map); free(c->left_map);
    c->right_feats = nrf; c->left_feats = nlf; c->right_map = nrm; c->left_map = nlm;
    /* the small matrix is renumbered by first use in the new id order (l.236-261), then permuted (matrix_connector.rs:99-116) */
    uint16_t *mm_l = (uint16_t *)malloc(2 * (size_t)c->m_num_left), *mm_r = (uint16_t *)malloc(2 * (size_t)c->m_num_right);
    memset(mm_l, 0xFF, 2 * (size_t)c->m_num_left); memset(mm_r, 0xFF, 2 * (size_t)c->m_num_right);
    uint16_t next_l = 0, next_r = 0;
    for (uint32_t l = 0; l < c->num_left; l++) {
        uint16_t *m = &mm_l[c->left_map[l]];
        if (*m == 0xFFFF) *m = next_l++;
        c->left_map[l] = *m;
    }
    for (uint32_t r = 0; r < c->num_right; r++) {
        uint16_t *m = &mm_r[c->right_map[r]];
        if (*m == 0xFFFF) *m = next_r++;
        c->right_map[r] = *m;
    }
    /* (a class no id uses keeps 0xFFFF in the reference and indexes out of bounds there; every class here is used: classes come
       from the rows, class 0 from id 0) */
    int16_t *nm = (int16_t *)calloc((size_t)c->m_num_left * c->m_num_right, 2);
    for (uint32_t r = 0; r < c->m_num_right; r++)
        for (uint32_t l = 0; l < c->m_num_left; l++)
            if (mm_l[l] != 0xFFFF && mm_r[r] != 0xFFFF) nm[(size_t)mm_l[l] * c->m_num_right + mm_r[r]] = c->matrix[(size_t)l * c->m_num_right + r];
    free(c->matrix); free(mm_l); free(mm_r);
    c->matrix = nm;
}

/* ---- test entry points for the unit vectors (dual_connector.rs:284-384) ---- */
ORA_API void *ora_dual_connector_new(const char *right, size_t rl, const char *left, size_t ll, const char *cost, size_t cl, char *err, size_t errcap) {
    return dual_from_sources(right, rl, left, ll, cost, cl, err, errcap);
}
ORA_API void ora_dual_connector_free(void *c) { dual_free((ora_dual_connector *)c); }
ORA_API int32_t ora_dual_connector_cost(const void *c, uint32_t right_id, uint32_t left_id) { return dual_cost((const ora_dual_connector *)c, right_id, left_id); }
ORA_API uint32_t ora_dual_connector_num(const void *c, int left) { return left ? ((const ora_dual_connector *)c)->num_left : ((const ora_dual_connector *)c)->num_right; }
ORA_API void ora_dual_connector_map(void *c, const uint16_t *ml, const uint16_t *mr) { dual_map_ids((ora_dual_connector *)c, ml, mr); }
ORA_API void ora_dual_connector_matrix_shape(const void *c, uint32_t *num_right, uint32_t *num_left) {
    *num_right = ((const ora_dual_connector *)c)->m_num_right; *num_left = ((const ora_dual_connector *)c)->m_num_left;
}
