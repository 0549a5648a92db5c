// Compact connectors on the host (product code): RawConnector / DualConnector of the reference
// (paths relative to /root/reference/vibrato/src/dictionary/connector):
//   raw_connector.rs:45-105, 200-330  bigram.right / bigram.left / bigram.cost parsing, feature matrices
//   raw_connector/scorer.rs:103-168   ScorerBuilder (two-level trie -> double-array hash)
//   raw_connector/scorer.rs:257-345   retrieve_cost / accumulate_cost (scalar path; the AVX2 path computes the same sum)
//   dual_connector.rs:267-279         cost = matrix over mapped ids + one 8-wide raw row per id
// The device never probes these structures per lattice pair: when a tokenizer is created, one kernel evaluates the cost
// function for every (left, right) id pair into the dense i16 matrix the sweep kernels read (engine.hip, expand_connector).
#include <algorithm>
#include <charconv>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>

#include "dict.hpp"

namespace vbt {
namespace {

constexpr uint32_t kInvalidFeature = 0x7FFFFFFFu;  // INVALID_FEATURE_ID = U31::MAX, raw_connector.rs:17-19
constexpr uint32_t kUnusedCheck = 0xFFFFFFFFu;     // scorer.rs:15
constexpr uint32_t kSimd = 8;                      // SIMD_SIZE, scorer.rs:17

[[noreturn]] void fail(int code, const std::string& msg) { throw Error(code, msg); }

// BufRead::lines(): split at '\n', a trailing '\r' is stripped, no empty line behind a final newline
std::vector<std::string_view> lines_of(std::string_view buf) {
    std::vector<std::string_view> out;
    size_t pos = 0;
    while (pos < buf.size()) {
        size_t nl = buf.find('\n', pos);
        if (nl == std::string_view::npos) nl = buf.size();
        std::string_view line = buf.substr(pos, nl - pos);
        if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
        out.push_back(line);
        pos = nl + 1;
    }
    return out;
}

bool parse_i32(std::string_view s, int32_t& out) {  // Rust str::parse::<i32>
    if (s.empty()) return false;
    if (s[0] == '+') {
        s.remove_prefix(1);
        if (s.empty() || s[0] == '+' || s[0] == '-') return false;
    }
    auto r = std::from_chars(s.data(), s.data() + s.size(), out);
    return r.ec == std::errc() && r.ptr == s.data() + s.size();
}

using IdMap = std::unordered_map<std::string, uint32_t>;

// ScorerBuilder::build, scorer.rs:131-168 (first base without a collision, keys of a row in ascending order)
Scorer build_scorer(const std::vector<std::map<uint32_t, int32_t>>& trie) {
    Scorer s;
    s.bases.assign(trie.size(), 0);
    for (size_t key1 = 0; key1 < trie.size(); ++key1) {
        const auto& second = trie[key1];
        uint32_t base = 0;
        for (;; ++base) {
            bool ok = true;
            for (const auto& kv : second) {
                const size_t pos = base ^ kv.first;
                if (pos < s.checks.size() && s.checks[pos] != kUnusedCheck) { ok = false; break; }
            }
            if (ok) break;
        }
        s.bases[key1] = base;
        for (const auto& kv : second) {
            const size_t pos = base ^ kv.first;
            if (pos >= s.checks.size()) { s.checks.resize(pos + 1, kUnusedCheck); s.costs.resize(pos + 1, 0); }
            s.checks[pos] = (uint32_t)key1;
            s.costs[pos] = kv.second;
        }
    }
    return s;
}

struct RawBuilder {  // RawConnectorBuilder, raw_connector.rs:170-250
    std::vector<std::vector<uint32_t>> right_rows, left_rows;
    size_t template_size = 0;
    std::vector<std::map<uint32_t, int32_t>> trie;
};

// RawConnectorBuilder::parse_features, raw_connector.rs:255-275
std::pair<size_t, std::vector<uint32_t>> parse_features(std::string_view line, const IdMap& ids, const char* name) {
    const size_t tab = line.find('\t');
    if (tab == std::string_view::npos || line.find('\t', tab + 1) != std::string_view::npos)
        fail(VBT_ERR_INVALID_FORMAT, std::string(name) + ": The format must be id<tab>csv_row, " + std::string(line));
    size_t id = 0;
    const std::string_view id_str = line.substr(0, tab);
    auto r = std::from_chars(id_str.data(), id_str.data() + id_str.size(), id);
    if (id_str.empty() || r.ec != std::errc() || r.ptr != id_str.data() + id_str.size())
        fail(VBT_ERR_PARSE_INT, std::string(name) + ": invalid id");
    std::vector<uint32_t> out;
    for (const std::string& f : parse_csv_row(line.substr(tab + 1))) {
        auto it = ids.find(f);
        out.push_back(it == ids.end() ? kInvalidFeature : it->second);
    }
    return {id, out};
}

RawBuilder parse_bigram(std::string_view right, std::string_view left, std::string_view cost) {
    RawBuilder b;
    IdMap right_ids, left_ids;
    right_ids.emplace("", 0u);  // raw_connector.rs:193-196
    left_ids.emplace("", 0u);
    for (std::string_view line : lines_of(cost)) {  // parse_cost, raw_connector.rs:294-325
        const size_t tab = line.find('\t');
        if (tab == std::string_view::npos || line.find('\t', tab + 1) != std::string_view::npos)
            fail(VBT_ERR_INVALID_FORMAT, "bigram.cost: The format must be right/left<tab>cost, " + std::string(line));
        int32_t c;
        if (!parse_i32(line.substr(tab + 1), c)) fail(VBT_ERR_PARSE_INT, "bigram.cost: invalid cost");
        const std::string_view feats = line.substr(0, tab);
        const size_t slash = feats.find('/');
        if (slash == std::string_view::npos || feats.find('/', slash + 1) != std::string_view::npos)
            fail(VBT_ERR_INVALID_FORMAT, "bigram.cost: The format must be right/left<tab>cost, " + std::string(line));
        const uint32_t rid = right_ids.emplace(std::string(feats.substr(0, slash)), (uint32_t)right_ids.size()).first->second;
        const uint32_t lid = left_ids.emplace(std::string(feats.substr(slash + 1)), (uint32_t)left_ids.size()).first->second;
        if (rid >= b.trie.size()) b.trie.resize(rid + 1);  // ScorerBuilder::insert, scorer.rs:113-119 (a repeated pair overwrites)
        b.trie[rid][lid] = c;
    }
    auto rows = [&](std::string_view buf, const IdMap& ids, const char* name, std::vector<std::vector<uint32_t>>& out) {
        size_t i = 0;
        for (std::string_view line : lines_of(buf)) {
            auto [id, feats] = parse_features(line, ids, name);
            if (id != i + 1) fail(VBT_ERR_INVALID_FORMAT, std::string(name) + ": must be ascending order");
            b.template_size = std::max(b.template_size, feats.size());
            out.push_back(std::move(feats));
            ++i;
        }
    };
    rows(right, right_ids, "bigram.right", b.right_rows);
    rows(left, left_ids, "bigram.left", b.left_rows);
    return b;
}

// (N + 1) x width matrix of feature ids; row 0 (BOS/EOS) all zero, short rows padded with the invalid id (raw_connector.rs:62-92)
std::vector<uint32_t> feature_matrix(const std::vector<std::vector<uint32_t>>& rows, size_t width) {
    std::vector<uint32_t> m((rows.size() + 1) * width, kInvalidFeature);
    std::fill(m.begin(), m.begin() + (long)width, 0u);
    for (size_t i = 0; i < rows.size(); ++i) std::copy(rows[i].begin(), rows[i].end(), m.begin() + (long)((i + 1) * width));
    return m;
}

template <typename T>
void permute_rows(std::vector<T>& v, size_t width, const std::vector<uint16_t>& map) {
    std::vector<T> out(v.size());
    for (size_t id = 0; id < map.size(); ++id) std::copy(v.begin() + (long)(id * width), v.begin() + (long)((id + 1) * width), out.begin() + (long)(map[id] * width));
    v.swap(out);
}

}  // namespace

std::vector<std::string> parse_csv_row(std::string_view row) {
    // csv_core::Reader defaults on one record: ',' separates fields, a field that STARTS with '"' is quoted ("" = a quote)
    std::vector<std::string> out;
    size_t pos = 0;
    for (;;) {
        std::string field;
        if (pos < row.size() && row[pos] == '"') {
            ++pos;
            while (pos < row.size()) {
                if (row[pos] == '"') {
                    if (pos + 1 < row.size() && row[pos + 1] == '"') { field.push_back('"'); pos += 2; continue; }
                    ++pos;
                    break;
                }
                field.push_back(row[pos++]);
            }
            while (pos < row.size() && row[pos] != ',') field.push_back(row[pos++]);  // (text behind the closing quote is kept)
        } else {
            while (pos < row.size() && row[pos] != ',') field.push_back(row[pos++]);
        }
        out.push_back(std::move(field));
        if (pos >= row.size()) break;
        ++pos;  // the comma
        if (pos == row.size()) { out.emplace_back(); break; }  // a trailing comma ends with an empty field
    }
    return out;
}

int32_t scorer_accumulate(const Scorer& s, const uint32_t* keys1, const uint32_t* keys2, size_t n) {
    uint32_t score = 0;  // (i32 in the reference; wrapping keeps release-build behaviour)
    for (size_t i = 0; i < n; ++i) {
        const uint32_t k1 = keys1[i], k2 = keys2[i];
        if (k1 >= s.bases.size()) continue;  // scorer.rs:314-324
        const size_t pos = s.bases[k1] ^ k2;
        if (pos < s.checks.size() && s.checks[pos] == k1) score += (uint32_t)s.costs[pos];
    }
    return (int32_t)score;
}

int32_t conn_cost(const Dictionary& d, uint32_t right_id, uint32_t left_id) {
    switch (d.conn_kind) {
        case kConnRaw:  // raw_connector.rs:153-161
            return scorer_accumulate(d.raw.scorer, &d.raw.right_feats[(size_t)right_id * d.raw.width], &d.raw.left_feats[(size_t)left_id * d.raw.width], d.raw.width);
        case kConnDual: {  // dual_connector.rs:267-279
            const uint32_t r = d.dual.right_map[right_id], l = d.dual.left_map[left_id];
            return (int32_t)d.dual.matrix[(size_t)l * d.dual.m_num_right + r] +
                   scorer_accumulate(d.dual.scorer, &d.dual.right_feats[(size_t)right_id * kSimd], &d.dual.left_feats[(size_t)left_id * kSimd], kSimd);
        }
        default:
            return d.matrix[(size_t)left_id * d.num_right + right_id];
    }
}

Dictionary* build_dictionary_bigram(std::string_view lex, std::string_view bigram_right, std::string_view bigram_left,
                                    std::string_view bigram_cost, std::string_view char_def, std::string_view unk_def, bool dual) {
    (void)dual;  // layout choice of the reference only (dual_connector.rs:141-198 splits the same cost function); see dict.hpp
    auto d = std::make_unique<Dictionary>();
    RawBuilder b = parse_bigram(bigram_right, bigram_left, bigram_cost);
    size_t width = b.template_size;
    if (width) width = ((width - 1) / kSimd + 1) * kSimd;  // raw_connector.rs:58-60
    if (b.right_rows.size() + 1 > 0xFFFF || b.left_rows.size() + 1 > 0xFFFF) fail(VBT_ERR_INVALID_ARGUMENT, "bigram: too many connection ids");
    d->conn_kind = kConnRaw;
    d->raw.width = (uint32_t)width;
    d->raw.right_feats = feature_matrix(b.right_rows, width);
    d->raw.left_feats = feature_matrix(b.left_rows, width);
    d->raw.scorer = build_scorer(b.trie);
    d->num_right = (uint32_t)b.right_rows.size() + 1;
    d->num_left = (uint32_t)b.left_rows.size() + 1;
    finish_dictionary(*d, lex, char_def, unk_def);
    return d.release();
}

void map_connector_ids(Dictionary& d, const std::vector<uint16_t>& ml, const std::vector<uint16_t>& mr) {
    if (d.conn_kind == kConnRaw) {  // raw_connector.rs:118-146
        permute_rows(d.raw.right_feats, d.raw.width, mr);
        permute_rows(d.raw.left_feats, d.raw.width, ml);
    } else if (d.conn_kind == kConnDual) {  // dual_connector.rs:211-264
        DualConnector& u = d.dual;
        permute_rows(u.right_feats, kSimd, mr);
        permute_rows(u.left_feats, kSimd, ml);
        permute_rows(u.right_map, 1, mr);
        permute_rows(u.left_map, 1, ml);
        // the small matrix is renumbered in order of first use by the new connection ids (l.237-262)
        auto renumber = [](std::vector<uint16_t>& id_map, uint32_t n_old) {
            std::vector<uint16_t> to_new(n_old, 0xFFFF);
            uint16_t next = 0;
            for (uint16_t& i : id_map) {
                if (to_new[i] == 0xFFFF) to_new[i] = next++;
                i = to_new[i];
            }
            return to_new;
        };
        const std::vector<uint16_t> lnew = renumber(u.left_map, u.m_num_left), rnew = renumber(u.right_map, u.m_num_right);
        std::vector<int16_t> mapped(u.matrix.size());  // matrix_connector.rs:99-116 (ids no connection id uses keep no cell)
        for (uint32_t l = 0; l < u.m_num_left; ++l)
            for (uint32_t r = 0; r < u.m_num_right; ++r)
                if (lnew[l] != 0xFFFF && rnew[r] != 0xFFFF) mapped[(size_t)lnew[l] * u.m_num_right + rnew[r]] = u.matrix[(size_t)l * u.m_num_right + r];
        u.matrix.swap(mapped);
    }
}

}  // namespace vbt
