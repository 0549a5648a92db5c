// Host-side dictionary construction (product code; see dict.hpp).
//
// Format semantics follow the reference parsers (paths relative to
// /root/reference/vibrato/src):
//   lex.csv / user.csv / unk.def : dictionary/lexicon.rs:111-200 (csv_core defaults)
//   matrix.def                   : dictionary/connector/matrix_connector.rs:27-77
//   char.def                     : dictionary/character.rs:140-281
//   unk.def grouping             : dictionary/unknown.rs:230-263
// The double-array layout is this project's own (one 16-byte node per transition,
// no END_CODE children); results depend only on the enumeration contract
// (increasing end_char, word ids ascending per surface; lexicon.rs:232-272).
#include "dict.hpp"

#include <algorithm>
#include <functional>
#include <charconv>
#include <cstring>
#include <memory>
#include <numeric>
#include <unordered_map>

namespace vbt {
namespace {

[[noreturn]] void fail(int code, const std::string& msg) { throw Error(code, msg); }

// ---- UTF-8 -------------------------------------------------------------------

// Strict decoder (Rust `str` validity). Returns bytes consumed, 0 if invalid.
int decode_utf8(const unsigned char* s, size_t n, uint32_t& cp) {
    if (n == 0) return 0;
    unsigned b0 = s[0];
    if (b0 < 0x80) { cp = b0; return 1; }
    auto cont = [&](size_t i) { return i < n && (s[i] & 0xC0) == 0x80; };
    if (b0 >= 0xC2 && b0 < 0xE0) {
        if (!cont(1)) return 0;
        cp = ((b0 & 0x1F) << 6) | (s[1] & 0x3F);
        return 2;
    }
    if (b0 >= 0xE0 && b0 < 0xF0) {
        if (!cont(1) || !cont(2)) return 0;
        cp = ((b0 & 0x0F) << 12) | ((s[1] & 0x3F) << 6) | (s[2] & 0x3F);
        if (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF)) return 0;
        return 3;
    }
    if (b0 >= 0xF0 && b0 < 0xF5) {
        if (!cont(1) || !cont(2) || !cont(3)) return 0;
        cp = ((b0 & 0x07) << 18) | ((s[1] & 0x3F) << 12) | ((s[2] & 0x3F) << 6) | (s[3] & 0x3F);
        if (cp < 0x10000 || cp > 0x10FFFF) return 0;
        return 4;
    }
    return 0;
}

}  // namespace

// Rust `str` validity of a byte range (what the reference's API takes by type, sentence.rs:28-32).
bool valid_utf8(const uint8_t* s, size_t n) {
    size_t i = 0;
    while (i < n) {
        if (i + 8 <= n) {  // ASCII fast path
            uint64_t w;
            std::memcpy(&w, s + i, 8);
            if (!(w & 0x8080808080808080ull)) { i += 8; continue; }
        }
        uint32_t cp;
        const int k = decode_utf8(s + i, n - i, cp);
        if (!k) return false;
        i += (size_t)k;
    }
    return true;
}

namespace {

std::u32string to_code_points(std::string_view s, const char* what) {
    std::u32string out;
    out.reserve(s.size());
    for (size_t i = 0; i < s.size();) {
        uint32_t cp;
        int k = decode_utf8(reinterpret_cast<const unsigned char*>(s.data()) + i, s.size() - i, cp);
        if (!k) fail(VBT_ERR_UTF8, std::string(what) + ": invalid UTF-8");
        out.push_back(cp);
        i += k;
    }
    return out;
}

// ---- integers (Rust str::parse) ------------------------------------------------

template <typename T>
bool parse_number(std::string_view s, T& out) {
    if (s.empty()) return false;
    if (s[0] == '+') {  // Rust accepts one leading '+'
        s.remove_prefix(1);
        if (s.empty() || s[0] == '+' || s[0] == '-') return false;
    }
    auto [p, ec] = std::from_chars(s.data(), s.data() + s.size(), out);
    return ec == std::errc() && p == s.data() + s.size();
}

// ---- CSV rows -------------------------------------------------------------------

struct LexRow {
    std::string surface;
    WordParam param;
    std::string_view feature;  // raw record tail, borrowed from the input buffer
};

// Splits MeCab lexicon CSV the way Lexicon::parse_csv does (lexicon.rs:111-200):
// first four fields unquoted by RFC-4180 rules, everything after the 4th delimiter is
// the raw feature string; blank lines skipped; \r, \n and \r\n all terminate a record.
std::vector<LexRow> parse_lexicon_csv(std::string_view buf, const char* name) {
    std::vector<LexRow> rows;
    size_t pos = 0;
    const size_t len = buf.size();
    std::string field;
    while (pos < len) {
        if (buf[pos] == '\n' || buf[pos] == '\r') { ++pos; continue; }
        const size_t rec_begin = pos;
        LexRow row;
        size_t feature_begin = 0;
        int idx = 0;
        for (;;) {
            field.clear();
            bool at_record_end = false;
            if (pos < len && buf[pos] == '"') {
                ++pos;
                while (pos < len) {
                    char c = buf[pos];
                    if (c == '"') {
                        if (pos + 1 < len && buf[pos + 1] == '"') { field.push_back('"'); pos += 2; continue; }
                        ++pos;
                        break;
                    }
                    field.push_back(c);
                    ++pos;
                }
            }
            for (;;) {
                if (pos >= len) { at_record_end = true; break; }
                char c = buf[pos];
                if (c == ',') { ++pos; break; }
                if (c == '\n' || c == '\r') { at_record_end = true; break; }
                field.push_back(c);
                ++pos;
            }
            auto bad_int = [&]() { fail(VBT_ERR_PARSE_INT, std::string(name) + ": invalid integer '" + field + "'"); };
            switch (idx) {
                case 0: row.surface = field; break;
                case 1: if (!parse_number(field, row.param.left_id)) bad_int(); break;
                case 2: if (!parse_number(field, row.param.right_id)) bad_int(); break;
                case 3:
                    if (!parse_number(field, row.param.word_cost)) bad_int();
                    feature_begin = pos;
                    break;
                default: break;
            }
            if (at_record_end) break;
            ++idx;
        }
        const size_t rec_end = pos;
        if (pos < len && buf[pos] == '\r') { ++pos; if (pos < len && buf[pos] == '\n') ++pos; }
        else if (pos < len && buf[pos] == '\n') ++pos;
        if (idx <= 3)
            fail(VBT_ERR_INVALID_FORMAT, std::string(name) + ": A csv row of lexicon must have five items at least, " +
                                             std::string(buf.substr(rec_begin, rec_end - rec_begin)));
        row.feature = buf.substr(feature_begin, rec_end - feature_begin);
        (void)to_code_points(row.surface, name);  // surface must be valid UTF-8
        if (row.surface.empty()) continue;        // "Skipped an empty surface": takes no word id
        rows.push_back(std::move(row));
    }
    return rows;
}

// ---- double-array construction --------------------------------------------------

struct Key {
    std::vector<uint16_t> codes;
    uint32_t val, cnt;
};

class DoubleArrayBuilder {
  public:
    DoubleArrayBuilder(uint32_t alphabet) {
        block_ = 256;
        while (block_ <= alphabet) block_ <<= 1;
    }

    std::vector<TrieNode> build(std::vector<Key>& keys) {
        std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.codes < b.codes; });
        grow();
        take(0);
        nodes_[0] = TrieNode{0, kUnused, 0, 0};
        struct Frame { uint32_t node, lo, hi, depth; };
        std::vector<Frame> stack;
        if (!keys.empty()) stack.push_back({0, 0, (uint32_t)keys.size(), 0});
        std::vector<uint32_t> codes, starts;
        while (!stack.empty()) {
            Frame f = stack.back();
            stack.pop_back();
            uint32_t i = f.lo;
            if (keys[i].codes.size() == f.depth) {  // a word ends at this node
                nodes_[f.node].val = keys[i].val;
                nodes_[f.node].cnt = keys[i].cnt;
                ++i;
            }
            codes.clear();
            starts.clear();
            while (i < f.hi) {
                uint32_t c = keys[i].codes[f.depth];
                codes.push_back(c);
                starts.push_back(i);
                while (i < f.hi && keys[i].codes[f.depth] == c) ++i;
            }
            if (codes.empty()) continue;
            starts.push_back(f.hi);
            uint32_t base = find_base(codes);
            nodes_[f.node].base = base;
            for (uint32_t c : codes) {
                uint32_t child = base ^ c;
                take(child);
                nodes_[child] = TrieNode{0, f.node, 0, 0};
            }
            for (size_t j = codes.size(); j-- > 0;)
                stack.push_back({base ^ codes[j], starts[j], starts[j + 1], f.depth + 1});
        }
        return std::move(nodes_);
    }

  private:
    static constexpr uint32_t kNone = 0xFFFFFFFFu;
    static constexpr uint32_t kUnused = 0xFFFFFFFFu;  // check value of free slots and of the root

    void grow() {
        uint32_t old = (uint32_t)nodes_.size(), cap = old + block_;
        nodes_.resize(cap, TrieNode{0, kUnused, 0, 0});
        next_.resize(cap);
        prev_.resize(cap);
        used_.resize(cap, 0);
        for (uint32_t i = old; i < cap; ++i) {
            next_[i] = i + 1 < cap ? i + 1 : kNone;
            prev_[i] = i > old ? i - 1 : tail_;
        }
        if (head_ == kNone) head_ = old; else next_[tail_] = old;
        tail_ = cap - 1;
    }
    void take(uint32_t i) {
        uint32_t n = next_[i], p = prev_[i];
        if (p != kNone) next_[p] = n; else head_ = n;
        if (n != kNone) prev_[n] = p; else tail_ = p;
        if (rover_ == i) rover_ = n;
        used_[i] = 1;
    }
    // Find base such that every slot base^code is free. Multi-child nodes search from a
    // roving pointer that skips regions where only isolated holes are left; single-child
    // nodes fill those holes from the true head of the free list.
    uint32_t find_base(const std::vector<uint32_t>& codes) {
        const size_t k = codes.size();
        for (;;) {
            uint32_t e = (k == 1 || rover_ == kNone) ? head_ : rover_;
            uint32_t tries = 0;
            for (; e != kNone; e = next_[e], ++tries) {
                uint32_t base = e ^ codes[0];
                size_t j = 1;
                while (j < k && !used_[base ^ codes[j]]) ++j;
                if (j == k) {
                    if (k > 1 && tries > 32) rover_ = e;
                    return base;
                }
            }
            uint32_t first_new = (uint32_t)nodes_.size();
            grow();
            if (k > 1) rover_ = first_new;
        }
    }

    uint32_t block_;
    std::vector<TrieNode> nodes_;
    std::vector<uint32_t> next_, prev_;
    std::vector<uint8_t> used_;
    uint32_t head_ = kNone, tail_ = kNone, rover_ = kNone;
};

// Lexicon::from_entries (lexicon.rs:85-96): word id = row index; WordMapBuilder
// (map.rs:45-73) groups equal surfaces, ids in insertion (= ascending) order.
void build_lexicon(Lexicon& lx, std::vector<LexRow>& rows, const char* name) {
    const uint32_t n = (uint32_t)rows.size();
    lx.params.resize(n);
    lx.features.resize(n);
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    for (uint32_t i = 0; i < n; ++i) {
        lx.params[i] = rows[i].param;
        lx.features[i].assign(rows[i].feature);
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rows[a].surface < rows[b].surface; });

    // distinct surfaces -> keys; entries grouped per surface in ascending word id
    lx.entries.reserve(n);
    std::vector<std::u32string> key_cps;
    std::vector<Key> keys;
    std::unordered_map<uint32_t, uint32_t> freq;
    uint32_t max_cp = 0;
    for (uint32_t i = 0; i < n;) {
        uint32_t j = i;
        while (j < n && rows[order[j]].surface == rows[order[i]].surface) ++j;
        Key k;
        k.val = (uint32_t)lx.entries.size();
        k.cnt = j - i;
        for (uint32_t t = i; t < j; ++t) {
            const WordParam& p = lx.params[order[t]];
            lx.entries.push_back(Entry{order[t], (uint32_t)p.left_id | ((uint32_t)p.right_id << 16), (uint32_t)(uint16_t)p.word_cost});
        }
        key_cps.push_back(to_code_points(rows[order[i]].surface, name));
        for (char32_t c : key_cps.back()) { ++freq[c]; max_cp = std::max<uint32_t>(max_cp, c); }
        lx.max_word_chars = std::max<uint32_t>(lx.max_word_chars, (uint32_t)key_cps.back().size());
        keys.push_back(std::move(k));
        i = j;
    }
    // code mapper: most frequent characters get the smallest codes (dense XOR targets)
    std::vector<std::pair<uint32_t, uint32_t>> by_freq(freq.begin(), freq.end());
    std::sort(by_freq.begin(), by_freq.end(), [](auto& a, auto& b) { return a.second != b.second ? a.second > b.second : a.first < b.first; });
    if (by_freq.size() >= 0xFFFF) fail(VBT_ERR_UNSUPPORTED, std::string(name) + ": more than 65534 distinct characters in surfaces");
    lx.alphabet = (uint32_t)by_freq.size();
    lx.mapper.assign(keys.empty() ? 1 : (size_t)max_cp + 1, 0);
    for (uint32_t i = 0; i < by_freq.size(); ++i) lx.mapper[by_freq[i].first] = (uint16_t)(i + 1);
    for (size_t i = 0; i < keys.size(); ++i) {
        keys[i].codes.reserve(key_cps[i].size());
        for (char32_t c : key_cps[i]) keys[i].codes.push_back(lx.mapper[c]);
    }
    DoubleArrayBuilder b(lx.alphabet);
    lx.nodes = b.build(keys);
}

bool verify_ids(const std::vector<WordParam>& params, uint32_t num_left, uint32_t num_right) {
    for (const WordParam& p : params)
        if (num_left <= p.left_id || num_right <= p.right_id) return false;
    return true;
}

// ---- line helpers -------------------------------------------------------------------

// BufRead::lines(): split on '\n'; a '\r' is stripped only as part of "\r\n" (an unterminated last line keeps it, and the
// reference's integer parse then fails on it).
bool next_line(std::string_view buf, size_t& pos, std::string_view& line) {
    if (pos >= buf.size()) return false;
    size_t e = buf.find('\n', pos);
    size_t end = e == std::string_view::npos ? buf.size() : e;
    line = buf.substr(pos, end - pos);
    if (e != std::string_view::npos && !line.empty() && line.back() == '\r') line.remove_suffix(1);
    pos = e == std::string_view::npos ? buf.size() : e + 1;
    return true;
}

std::vector<std::string_view> split_on_space(std::string_view s) {  // str::split(' '), keeps empties
    std::vector<std::string_view> out;
    size_t st = 0;
    for (size_t i = 0; i <= s.size(); ++i)
        if (i == s.size() || s[i] == ' ') { out.push_back(s.substr(st, i - st)); st = i + 1; }
    return out;
}

bool is_space_char(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

std::vector<std::string_view> split_whitespace(std::string_view s) {
    std::vector<std::string_view> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && is_space_char(s[i])) ++i;
        size_t st = i;
        while (i < s.size() && !is_space_char(s[i])) ++i;
        if (i > st) out.push_back(s.substr(st, i - st));
    }
    return out;
}

// ---- matrix.def ----------------------------------------------------------------------

void parse_matrix_def(Dictionary& d, std::string_view buf) {
    size_t pos = 0;
    std::string_view line;
    if (!next_line(buf, pos, line)) fail(VBT_ERR_INVALID_FORMAT, "matrix.def: missing header");
    auto cols = split_on_space(line);
    uint16_t nr, nl;
    if (cols.size() != 2)
        fail(VBT_ERR_INVALID_FORMAT, "matrix.def: The header must consists of two integers separated by spaces, " + std::string(line));
    if (!parse_number(cols[0], nr) || !parse_number(cols[1], nl)) fail(VBT_ERR_PARSE_INT, "matrix.def: invalid header " + std::string(line));
    d.num_right = nr;
    d.num_left = nl;
    d.matrix.assign((size_t)nr * nl, 0);
    while (next_line(buf, pos, line)) {
        if (line.empty()) continue;
        cols = split_on_space(line);
        if (cols.size() != 3)
            fail(VBT_ERR_INVALID_FORMAT,
                 "matrix.def: A row other than the header must consists of three integers separated by spaces, " + std::string(line));
        uint64_t r, l;
        int16_t c;
        if (!parse_number(cols[0], r) || !parse_number(cols[1], l) || !parse_number(cols[2], c))
            fail(VBT_ERR_PARSE_INT, "matrix.def: invalid row " + std::string(line));
        if (nr <= r || nl <= l) fail(VBT_ERR_INVALID_FORMAT, "matrix.def: left/right_id must be within num_left/right.");
        d.matrix[(size_t)l * nr + r] = c;
    }
}

// ---- char.def --------------------------------------------------------------------------

constexpr uint32_t kCateBits = 18;

uint32_t parse_hex(std::string_view s, std::string_view line) {
    while (s.size() >= 2 && s[0] == '0' && s[1] == 'x') s.remove_prefix(2);  // trim_start_matches("0x")
    uint32_t v = 0;
    auto [p, ec] = std::from_chars(s.data(), s.data() + s.size(), v, 16);
    if (s.empty() || ec != std::errc() || p != s.data() + s.size())
        fail(VBT_ERR_PARSE_INT, "char.def: invalid code point in " + std::string(line));
    return v;
}

void parse_char_def(Dictionary& d, std::string_view buf) {
    struct Range { uint32_t start, end; std::vector<std::string_view> cats; };
    std::vector<Range> ranges;
    std::vector<uint32_t> info;      // per category id: CharInfo::new(0, id, invoke, group, length)
    std::vector<uint8_t> defined;
    d.categories = {"DEFAULT"};      // cate_map.insert("DEFAULT", 0), character.rs:148
    info.push_back(0);
    defined.push_back(0);
    size_t pos = 0;
    std::string_view raw;
    while (next_line(buf, pos, raw)) {
        std::string_view line = raw;
        while (!line.empty() && is_space_char(line.front())) line.remove_prefix(1);
        while (!line.empty() && is_space_char(line.back())) line.remove_suffix(1);
        if (line.empty() || line.front() == '#') continue;
        auto cols = split_whitespace(line);
        if (line.substr(0, 2) != "0x") {
            if (cols.size() < 4)
                fail(VBT_ERR_INVALID_FORMAT, "char.def: A character category must consists of four items separated by spaces, " + std::string(line));
            if (cols[1] != "0" && cols[1] != "1") fail(VBT_ERR_INVALID_FORMAT, "char.def: INVOKE must be 1 or 0.");
            if (cols[2] != "0" && cols[2] != "1") fail(VBT_ERR_INVALID_FORMAT, "char.def: GROUP must be 1 or 0.");
            uint16_t length;
            if (!parse_number(cols[3], length)) fail(VBT_ERR_PARSE_INT, "char.def: invalid LENGTH in " + std::string(line));
            if (length >= 16) fail(VBT_ERR_INVALID_FORMAT, "char.def: LENGTH must be less than 16.");
            int id = d.cate_id(cols[0]);
            if (id < 0) {
                id = (int)d.categories.size();
                if (id >= (int)kCateBits) fail(VBT_ERR_UNSUPPORTED, "char.def: more than 18 categories");
                d.categories.emplace_back(cols[0]);
                info.push_back(0);
                defined.push_back(0);
            }
            info[id] = ((uint32_t)id << kCateBits) | ((uint32_t)(cols[1] == "1") << 26) | ((uint32_t)(cols[2] == "1") << 27) |
                       ((uint32_t)length << 28);
            defined[id] = 1;
        } else {
            if (cols.size() < 2) fail(VBT_ERR_INVALID_FORMAT, "char.def: A character range must have two items at least, " + std::string(line));
            Range r;
            size_t dd = cols[0].find("..");
            r.start = parse_hex(dd == std::string_view::npos ? cols[0] : cols[0].substr(0, dd), line);
            r.end = dd == std::string_view::npos ? r.start + 1 : parse_hex(cols[0].substr(dd + 2), line) + 1;
            if (r.start >= r.end)
                fail(VBT_ERR_INVALID_FORMAT, "char.def: The start of a character range must be no more than the end, " + std::string(line));
            if (r.start > 0xFFFF || r.end > 0x10000)
                fail(VBT_ERR_INVALID_FORMAT, "char.def: A character range must be no more 0xFFFF, " + std::string(line));
            for (size_t i = 1; i < cols.size() && cols[i].front() != '#'; ++i) r.cats.push_back(cols[i]);
            ranges.push_back(std::move(r));
        }
    }
    auto encode = [&](const std::vector<std::string_view>& cats) {  // encode_cate_info, character.rs:193-216
        uint32_t base = 0, set = 0;
        for (size_t i = 0; i < cats.size(); ++i) {
            int id = d.cate_id(cats[i]);
            if (id < 0 || !defined[id]) fail(VBT_ERR_INVALID_FORMAT, "char.def: Undefined category: " + std::string(cats[i]));
            if (i == 0) base = info[id];
            set |= 1u << id;
        }
        return (base & ~((1u << kCateBits) - 1)) | set;
    };
    if (!defined[0]) fail(VBT_ERR_INVALID_FORMAT, "char.def: Undefined category: DEFAULT");
    d.chr2inf.assign(1 << 16, encode({"DEFAULT"}));
    for (const Range& r : ranges) {
        if (r.cats.empty()) fail(VBT_ERR_INVALID_FORMAT, "char.def: A character range must name a category");
        uint32_t v = encode(r.cats);
        std::fill(d.chr2inf.begin() + r.start, d.chr2inf.begin() + r.end, v);
    }
}

// ---- unk.def ---------------------------------------------------------------------------

void parse_unk_def(Dictionary& d, std::string_view buf) {
    auto rows = parse_lexicon_csv(buf, "unk.def");
    std::vector<std::vector<size_t>> by_cat(d.categories.size());
    for (size_t i = 0; i < rows.size(); ++i) {
        int id = d.cate_id(rows[i].surface);
        if (id < 0) fail(VBT_ERR_INVALID_FORMAT, "unk.def: Undefined category: " + rows[i].surface);
        by_cat[id].push_back(i);
    }
    if (rows.size() > 0xFFFF) fail(VBT_ERR_UNSUPPORTED, "unk.def: more than 65535 rows");
    for (auto& v : by_cat) {
        d.unk_offsets.push_back((uint32_t)d.unk_entries.size());
        for (size_t i : v) {
            const WordParam& p = rows[i].param;
            d.unk_entries.push_back(Entry{(uint32_t)d.unk_entries.size(), (uint32_t)p.left_id | ((uint32_t)p.right_id << 16),
                                          (uint32_t)(uint16_t)p.word_cost});
            d.unk_features.emplace_back(rows[i].feature);
        }
    }
    d.unk_offsets.push_back((uint32_t)d.unk_entries.size());
}

}  // namespace

int Dictionary::cate_id(std::string_view name) const {
    for (size_t i = 0; i < categories.size(); ++i)
        if (categories[i] == name) return (int)i;
    return -1;
}

void Lexicon::common_prefix(const uint32_t* cps, size_t n, std::vector<std::pair<uint32_t, uint32_t>>& out) const {
    uint32_t cur = 0, base = nodes.empty() ? 0 : nodes[0].base;
    for (size_t i = 0; i < n; ++i) {
        uint32_t code = cps[i] < mapper.size() ? mapper[cps[i]] : 0;
        if (code == 0) break;
        uint32_t child = base ^ code;
        if (child >= nodes.size() || nodes[child].check != cur) break;
        cur = child;
        base = nodes[child].base;
        for (uint32_t k = 0; k < nodes[child].cnt; ++k) out.emplace_back(entries[nodes[child].val + k].word_id, (uint32_t)(i + 1));
    }
}

// A lexicon from (surface, param, feature) per word id -- the binary dictionary reader's way in (dictio.cpp)
void lexicon_from_words(Lexicon& lx, const std::function<const std::string&(uint32_t)>& surface, std::vector<WordParam> params,
                        std::vector<std::string> features, const char* name) {
    std::vector<LexRow> rows(params.size());
    for (uint32_t i = 0; i < rows.size(); ++i) {
        rows[i].surface = surface(i);
        rows[i].param = params[i];
        rows[i].feature = features[i];
    }
    build_lexicon(lx, rows, name);
}

// The keys of a lexicon's double array as code points, each with its span of `entries` -- the binary dictionary writer's way out
void lexicon_keys(const Lexicon& lx, std::vector<std::u32string>& keys, std::vector<std::pair<uint32_t, uint32_t>>& spans) {
    const uint32_t n = (uint32_t)lx.nodes.size();
    std::vector<uint32_t> code_to_cp(lx.alphabet + 1, 0);
    for (uint32_t cp = 0; cp < lx.mapper.size(); ++cp)
        if (lx.mapper[cp]) code_to_cp[lx.mapper[cp]] = cp;
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    std::vector<uint32_t> first(n, kNone), next(n, kNone), parent(n, kNone);
    for (uint32_t i = n; i-- > 1;) {
        const uint32_t p = lx.nodes[i].check;
        if (p >= n) continue;
        next[i] = first[p];
        first[p] = i;
        parent[i] = p;
    }
    std::vector<uint32_t> stack;
    if (n) stack.push_back(0);
    while (!stack.empty()) {
        const uint32_t v = stack.back();
        stack.pop_back();
        if (lx.nodes[v].cnt) {
            std::u32string key;
            for (uint32_t u = v; u != 0; u = parent[u]) key.push_back((char32_t)code_to_cp[lx.nodes[parent[u]].base ^ u]);
            std::reverse(key.begin(), key.end());
            keys.push_back(std::move(key));
            spans.push_back({lx.nodes[v].val, lx.nodes[v].cnt});
        }
        for (uint32_t c = first[v]; c != kNone; c = next[c]) stack.push_back(c);
    }
}

// the id checks of SystemDictionaryBuilder::build (builder.rs:24-35) and of reset_user_lexicon (dictionary.rs:218-223)
void verify_dictionary_ids(const Dictionary& d) {
    if (!verify_ids(d.system.params, d.num_left, d.num_right))
        fail(VBT_ERR_INVALID_ARGUMENT, "system_lexicon_rdr: system_lexicon_rdr includes invalid connection ids.");
    if (d.has_user && !verify_ids(d.user.params, d.num_left, d.num_right))
        fail(VBT_ERR_INVALID_ARGUMENT, "user_lexicon_rdr: includes invalid connection ids.");
    for (const Entry& e : d.unk_entries)
        if (d.num_left <= (e.left_right & 0xFFFF) || d.num_right <= (e.left_right >> 16))
            fail(VBT_ERR_INVALID_ARGUMENT, "unk_handler_rdr: unk_handler_rdr includes invalid connection ids.");
}

// lexicon, char.def, unk.def and the id checks of SystemDictionaryBuilder::build (builder.rs:16-47); the connector is set already
void finish_dictionary(Dictionary& d, std::string_view lex, std::string_view char_def, std::string_view unk_def) {
    auto rows = parse_lexicon_csv(lex, "lex.csv");
    parse_char_def(d, char_def);
    parse_unk_def(d, unk_def);
    build_lexicon(d.system, rows, "lex.csv");
    if (!verify_ids(d.system.params, d.num_left, d.num_right))  // builder.rs:24-29
        fail(VBT_ERR_INVALID_ARGUMENT, "system_lexicon_rdr: system_lexicon_rdr includes invalid connection ids.");
    for (const Entry& e : d.unk_entries)  // builder.rs:30-35
        if (d.num_left <= (e.left_right & 0xFFFF) || d.num_right <= (e.left_right >> 16))
            fail(VBT_ERR_INVALID_ARGUMENT, "unk_handler_rdr: unk_handler_rdr includes invalid connection ids.");
}

Dictionary* build_dictionary(std::string_view lex, std::string_view matrix_def, const int16_t* matrix_bin, uint32_t num_right,
                             uint32_t num_left, std::string_view char_def, std::string_view unk_def) {
    auto d = std::make_unique<Dictionary>();
    if (matrix_bin) {
        if (num_right > 0xFFFF || num_left > 0xFFFF) fail(VBT_ERR_INVALID_ARGUMENT, "matrix: num_right/num_left must fit in u16");
        d->num_right = num_right;
        d->num_left = num_left;
        d->matrix.assign(matrix_bin, matrix_bin + (size_t)num_right * num_left);
    } else {
        parse_matrix_def(*d, matrix_def);
    }
    finish_dictionary(*d, lex, char_def, unk_def);
    return d.release();
}

namespace {

// ConnIdMapper::parse (mapper.rs:49-80)
std::vector<uint16_t> parse_id_map(const uint16_t* map, size_t n) {
    std::vector<uint32_t> old_ids{0};
    for (size_t i = 0; i < n; ++i) {
        if (map[i] == 0) fail(VBT_ERR_INVALID_ARGUMENT, "map: Id 0 is reserved.");
        old_ids.push_back(map[i]);
    }
    if (old_ids.size() > 0x10000) fail(VBT_ERR_INVALID_ARGUMENT, "map: too many ids");
    std::vector<uint16_t> new_ids(old_ids.size(), 0xFFFF);
    new_ids[0] = 0;
    for (size_t new_id = 1; new_id < old_ids.size(); ++new_id) {
        const uint32_t old_id = old_ids[new_id];
        if (old_id >= new_ids.size()) fail(VBT_ERR_INVALID_ARGUMENT, "map: ids are out of range.");
        if (new_ids[old_id] != 0xFFFF) fail(VBT_ERR_INVALID_ARGUMENT, "map: ids are duplicate.");
        new_ids[old_id] = (uint16_t)new_id;
    }
    return new_ids;
}

void map_lexicon(Lexicon& lx, const std::vector<uint16_t>& ml, const std::vector<uint16_t>& mr) {  // param.rs:48-53
    for (WordParam& p : lx.params) { p.left_id = ml[p.left_id]; p.right_id = mr[p.right_id]; }
    for (Entry& e : lx.entries) e.left_right = (uint32_t)ml[e.left_right & 0xFFFF] | ((uint32_t)mr[e.left_right >> 16] << 16);
}

}  // namespace

void map_connection_ids(Dictionary& d, const uint16_t* lmap, size_t n_lmap, const uint16_t* rmap, size_t n_rmap) {
    std::vector<uint16_t> ml = parse_id_map(lmap, n_lmap), mr = parse_id_map(rmap, n_rmap);
    if (ml.size() != d.num_left || mr.size() != d.num_right)  // assert_eq! in matrix_connector.rs:100-101
        fail(VBT_ERR_INVALID_ARGUMENT, "map: the mappings must cover every connection id except 0");
    map_lexicon(d.system, ml, mr);
    if (d.has_user) map_lexicon(d.user, ml, mr);
    if (d.conn_kind == kConnMatrix) {
        std::vector<int16_t> mapped(d.matrix.size());  // matrix_connector.rs:99-116
        for (uint32_t l = 0; l < d.num_left; ++l) {
            const int16_t* src = d.matrix.data() + (size_t)l * d.num_right;
            int16_t* dst = mapped.data() + (size_t)ml[l] * d.num_right;
            for (uint32_t r = 0; r < d.num_right; ++r) dst[mr[r]] = src[r];
        }
        d.matrix.swap(mapped);
    } else {
        map_connector_ids(d, ml, mr);
    }
    for (Entry& e : d.unk_entries)  // unknown.rs:206-211
        e.left_right = (uint32_t)ml[e.left_right & 0xFFFF] | ((uint32_t)mr[e.left_right >> 16] << 16);
    d.mapper_left = std::move(ml);
    d.mapper_right = std::move(mr);
}

void set_user_lexicon(Dictionary& d, const char* csv, size_t len) {
    if (!csv) {
        d.has_user = false;
        d.user = Lexicon();
        return;
    }
    auto rows = parse_lexicon_csv(std::string_view(csv, len), "lex.csv");
    Lexicon lx;
    build_lexicon(lx, rows, "lex.csv");
    if (!d.mapper_left.empty()) {  // dictionary.rs:214-217: the stored mapper is applied before verification
        for (const WordParam& p : lx.params)
            if (p.left_id >= d.mapper_left.size() || p.right_id >= d.mapper_right.size())
                fail(VBT_ERR_INVALID_ARGUMENT, "user_lexicon_rdr: includes invalid connection ids.");
        map_lexicon(lx, d.mapper_left, d.mapper_right);
    }
    if (!verify_ids(lx.params, d.num_left, d.num_right))
        fail(VBT_ERR_INVALID_ARGUMENT, "user_lexicon_rdr: includes invalid connection ids.");
    d.user = std::move(lx);
    d.has_user = true;
}

}  // namespace vbt
