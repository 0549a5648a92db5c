// Compact connectors on the host (product code): RawConnector / DualConnector of the reference
// (paths relative to /root/reference/vibrato/src/dictionary/connector):
//   raw_connector.rs:45-105, 200-330  bigram.right / bigram.left / bigram.cost parsing, feature matrices
//   raw_connector/scorer.rs:103-168   ScorerBuilder (two-level trie -> double-array hash)
//   raw_connector/scorer.rs:257-345   retrieve_cost / accumulate_cost (scalar path; the AVX2 path computes the same sum)
//   dual_connector.rs:267-279         cost = matrix over mapped ids + one 8-wide raw row per id
// A dual connector is BUILT like the reference builds it (greedy choice of the eight raw templates, matrix over the classes of
// the remaining ones, pruned scorer; dual_connector.rs:25-199), with one documented difference: ties of the greedy choice.
// The device never probes these structures per lattice pair: when a tokenizer is created, one kernel evaluates the cost
// function for every (left, right) id pair into the dense i16 matrix the sweep kernels read (engine.hip, expand_connector).
#include <algorithm>
#include <charconv>
#include <iterator>
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

// BufRead::lines(): split at '\n'; a '\r' is stripped only as part of a "\r\n" terminator (a last line without '\n' keeps
// it); no empty line behind a final newline
std::vector<std::string_view> lines_of(std::string_view buf) {
    std::vector<std::string_view> out;
    size_t pos = 0;
    while (pos < buf.size()) {
        size_t nl = buf.find('\n', pos);
        const bool terminated = nl != std::string_view::npos;
        if (!terminated) nl = buf.size();
        std::string_view line = buf.substr(pos, nl - pos);
        if (terminated && !line.empty() && line.back() == '\r') line.remove_suffix(1);
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
    std::string_view id_str = line.substr(0, tab);
    if (id_str.size() > 1 && id_str[0] == '+' && id_str[1] != '+' && id_str[1] != '-') id_str.remove_prefix(1);  // usize::from_str takes one '+'
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

namespace {

// U31x8::to_simd_vec (scorer.rs:24-44): chunks of 8, the last one padded with feature id 0 -- NOT with the invalid id, so a
// bigram.cost line for the pair of empty features ("/") is counted once per padded position, as in the reference
std::vector<uint32_t> simd_padded(const std::vector<uint32_t>& v) {
    std::vector<uint32_t> out(v);
    out.resize((v.size() + kSimd - 1) / kSimd * kSimd, 0u);
    return out;
}

uint32_t feat_or_invalid(const std::vector<uint32_t>& row, size_t idx) { return idx < row.size() ? row[idx] : kInvalidFeature; }

// DualConnector::remove_feature_templates_greedy (dual_connector.rs:25-71): eight times, drop the template whose removal leaves
// the fewest distinct (right rows) x (left rows).  The reference walks a hash set, so which of several equally good templates
// goes is not reproducible even between two runs of the reference; here candidates are tried in ascending order and, like its
// `<=`, the last of the best wins.  The cost function does not depend on the choice (only the split between matrix and scorer).
std::vector<size_t> choose_matrix_templates(const RawBuilder& b) {
    std::vector<size_t> keep(b.template_size);
    for (size_t i = 0; i < keep.size(); ++i) keep[i] = i;
    auto distinct_rows = [&](const std::vector<std::vector<uint32_t>>& rows, size_t without) {
        std::vector<std::vector<uint32_t>> proj;
        proj.reserve(rows.size());
        for (const auto& row : rows) {
            std::vector<uint32_t> p;
            for (size_t i : keep)
                if (i != without && i < row.size()) p.push_back(row[i]);  // (a template a short row does not have is skipped, l.46-50)
            proj.push_back(std::move(p));
        }
        std::sort(proj.begin(), proj.end());
        return (size_t)(std::unique(proj.begin(), proj.end()) - proj.begin());
    };
    for (uint32_t round = 0; round < kSimd; ++round) {
        size_t best = 0, best_size = b.right_rows.size() * b.left_rows.size();
        for (size_t trial : keep) {
            const size_t size = distinct_rows(b.right_rows, trial) * distinct_rows(b.left_rows, trial);
            if (size <= best_size) { best_size = size; best = trial; }
        }
        keep.erase(std::remove(keep.begin(), keep.end(), best), keep.end());
    }
    return keep;
}

// DualConnector::from_readers (dual_connector.rs:145-199)
void build_dual(Dictionary& d, RawBuilder& b) {
    if (b.template_size < kSimd) fail(VBT_ERR_INVALID_ARGUMENT, "bigram: a dual connector needs at least 8 feature templates");  // (the reference underflows, l.84)
    const Scorer full = build_scorer(b.trie);
    const std::vector<size_t> matrix_idx = choose_matrix_templates(b);
    std::vector<size_t> raw_idx;
    for (size_t i = 0; i < b.template_size; ++i)
        if (!std::binary_search(matrix_idx.begin(), matrix_idx.end(), i)) raw_idx.push_back(i);
    DualConnector& u = d.dual;
    // create_matrix_connector (l.73-112): connection ids with equal matrix-template features share a matrix row / column
    auto classes = [&](const std::vector<std::vector<uint32_t>>& rows, std::vector<uint16_t>& id_map) {
        std::map<std::vector<uint32_t>, uint32_t> cls;
        cls.emplace(std::vector<uint32_t>(b.template_size - kSimd, 0u), 0u);  // BOS / EOS
        id_map.assign(1, 0);
        for (const auto& row : rows) {
            std::vector<uint32_t> f;
            for (size_t i : matrix_idx) f.push_back(feat_or_invalid(row, i));
            const uint32_t id = cls.emplace(std::move(f), (uint32_t)cls.size()).first->second;
            if (id > 0xFFFF) fail(VBT_ERR_INVALID_ARGUMENT, "bigram: too many matrix classes");
            id_map.push_back((uint16_t)id);
        }
        std::vector<std::vector<uint32_t>> by_id(cls.size());
        for (auto& kv : cls) by_id[kv.second] = simd_padded(kv.first);
        return by_id;
    };
    const auto rclass = classes(b.right_rows, u.right_map), lclass = classes(b.left_rows, u.left_map);
    u.m_num_right = (uint32_t)rclass.size();
    u.m_num_left = (uint32_t)lclass.size();
    u.matrix.resize((size_t)u.m_num_right * u.m_num_left);
    for (uint32_t l = 0; l < u.m_num_left; ++l)
        for (uint32_t r = 0; r < u.m_num_right; ++r) {
            const int32_t c = scorer_accumulate(full, rclass[r].data(), lclass[l].data(), rclass[r].size());
            u.matrix[(size_t)l * u.m_num_right + r] = (int16_t)std::min(32767, std::max(-32768, c));  // clamped, l.101
        }
    // create_raw_connector (l.114-143): the eight removed templates per id; scorer entries nobody can reach are dropped
    auto raw_rows = [&](const std::vector<std::vector<uint32_t>>& rows) {
        std::vector<uint32_t> out(raw_idx.size(), 0u);  // id 0: zeros
        for (const auto& row : rows)
            for (size_t i : raw_idx) out.push_back(feat_or_invalid(row, i));
        return out;
    };
    u.right_feats = raw_rows(b.right_rows);
    u.left_feats = raw_rows(b.left_rows);
    std::vector<uint32_t> ru(u.right_feats), lu(u.left_feats);
    std::sort(ru.begin(), ru.end());
    std::sort(lu.begin(), lu.end());
    for (size_t k1 = 0; k1 < b.trie.size(); ++k1) {
        if (!std::binary_search(ru.begin(), ru.end(), (uint32_t)k1)) { b.trie[k1].clear(); continue; }
        for (auto it = b.trie[k1].begin(); it != b.trie[k1].end();)
            it = std::binary_search(lu.begin(), lu.end(), it->first) ? std::next(it) : b.trie[k1].erase(it);
    }
    u.scorer = build_scorer(b.trie);
    d.conn_kind = kConnDual;
    d.num_right = (uint32_t)u.right_map.size();
    d.num_left = (uint32_t)u.left_map.size();
}

}  // namespace

Dictionary* build_dictionary_bigram(std::string_view lex, std::string_view bigram_right, std::string_view bigram_left,
                                    std::string_view bigram_cost, std::string_view char_def, std::string_view unk_def, bool dual) {
    auto d = std::make_unique<Dictionary>();
    RawBuilder b = parse_bigram(bigram_right, bigram_left, bigram_cost);
    if (b.right_rows.size() + 1 > 0xFFFF || b.left_rows.size() + 1 > 0xFFFF) fail(VBT_ERR_INVALID_ARGUMENT, "bigram: too many connection ids");
    if (dual) {
        build_dual(*d, b);
        finish_dictionary(*d, lex, char_def, unk_def);
        return d.release();
    }
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
