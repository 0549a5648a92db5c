// Dictionary::read / Dictionary::write (product code): vibrato's binary dictionary format, `system.dic`, optionally inside a
// zstd frame (`system.dic.zst`, as released and as the reference's CLIs read it: tokenize/src/main.rs:59-60,
// compile/src/main.rs:98).
//
// Layout (paths relative to /root/reference/vibrato/src):
//   dictionary.rs:27        MODEL_MAGIC = "VibratoTokenizer 0.5\n" (21 bytes), then one bincode value
//   common.rs:5-9           bincode 2 `standard().with_little_endian().with_fixed_int_encoding()`:
//                           integers fixed-width LE, usize/len = u64, Vec/String = len + items, Option = u8 tag,
//                           derived enum = u32 variant index, structs / tuples / arrays = fields back to back
//   dictionary.rs:43-51     DictionaryInner { system_lexicon, user_lexicon: Option, connector, mapper: Option, char_prop, unk_handler }
//   dictionary/lexicon.rs:23-29            Lexicon { map { trie: Vec<u8> blob, postings: Vec<u32> }, params: Vec<{u16,u16,i16}>,
//                                          features: Vec<String>, lex_type: enum }
//   dictionary/lexicon/map/trie.rs:14-19   the trie is `crawdad::Trie::serialize_to_vec()` wrapped in a Vec<u8>
//   dictionary/connector.rs:30-35          ConnectorWrapper { 0: Matrix, 1: Raw, 2: Dual }
//   connector/matrix_connector.rs:11-15    { data: Vec<i16>, num_right: usize, num_left: usize }
//   connector/raw_connector.rs:22-27       { right_feat_ids: Vec<U31x8>, left_feat_ids: Vec<U31x8>, feat_template_size: usize (in
//                                          8-wide blocks, l.95), scorer }; scorer.rs:230-237 { bases, checks: Vec<u32>, costs: Vec<i32> }
//   connector/dual_connector.rs:16-23      { matrix_connector, right_conn_id_map, left_conn_id_map: Vec<u16>, right_feat_ids,
//                                          left_feat_ids: Vec<U31x8>, raw_scorer }
//   dictionary/mapper.rs:9-12              ConnIdMapper { left: Vec<u16>, right: Vec<u16> }
//   dictionary/character.rs:105-108        CharProperty { chr2inf: Vec<u32>, categories: Vec<String> }
//   dictionary/unknown.rs:20-27,63-66      UnkHandler { offsets: Vec<usize>, entries: Vec<{cate_id, left_id, right_id: u16, word_cost: i16, feature}> }
//
// The trie blob is the one part whose format lives outside /root/reference (crate crawdad 0.3.0, vibrato/Cargo.toml:21; not
// vendored, no network).  Its layout is restated from the published crate and is NOT pinned by any
// fixture: the reader therefore checks the blob structurally (sizes, every word id reached exactly once through the trie +
// postings) and refuses anything that does not add up.  The device never sees this layout: the keys are re-enumerated and
// the GPU double array (16-byte nodes, dict.hpp) is rebuilt from them.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>

#include "dict.hpp"

namespace vbt {
namespace {

constexpr char kMagic[] = "VibratoTokenizer 0.5\n";  // dictionary.rs:27
constexpr size_t kMagicLen = sizeof(kMagic) - 1;
constexpr uint32_t kOffsetMask = 0x7FFFFFFFu;  // crawdad: low 31 bits = base / check, top bit = is_leaf / has_leaf
constexpr uint32_t kInvalidCode = 0xFFFFFFFFu; // crawdad CodeMapper: character not in any key

[[noreturn]] void bad(const std::string& m) { throw Error(VBT_ERR_INVALID_FORMAT, "dictionary: " + m); }

// ---------------------------------------------------------------- bincode (fixed-int, little endian)

struct Reader {
    const uint8_t* p;
    size_t n, pos = 0;
    void need(size_t k) const {
        if (k > n - pos) bad("unexpected end of data");
    }
    template <typename T>
    T num() {
        need(sizeof(T));
        T v;
        std::memcpy(&v, p + pos, sizeof(T));
        pos += sizeof(T);
        return v;
    }
    size_t len(size_t elem_bytes) {  // a Vec / String length that must fit in what is left
        const uint64_t v = num<uint64_t>();
        if (elem_bytes && v > (n - pos) / elem_bytes) bad("a length field exceeds the data");
        return (size_t)v;
    }
    template <typename T>
    std::vector<T> vec() {
        const size_t k = len(sizeof(T));
        std::vector<T> v(k);
        if (k) std::memcpy(v.data(), p + pos, k * sizeof(T));
        pos += k * sizeof(T);
        return v;
    }
    std::string str() {
        const size_t k = len(1);
        std::string s(reinterpret_cast<const char*>(p + pos), k);
        pos += k;
        if (!valid_utf8(reinterpret_cast<const uint8_t*>(s.data()), s.size())) bad("a string is not valid UTF-8");
        return s;
    }
    std::vector<std::string> strs() {
        const size_t k = len(8);
        std::vector<std::string> v;
        v.reserve(k);
        for (size_t i = 0; i < k; ++i) v.push_back(str());
        return v;
    }
    bool option() {
        const uint8_t t = num<uint8_t>();
        if (t > 1) bad("invalid Option tag");
        return t == 1;
    }
};

struct Writer {
    std::vector<uint8_t> out;
    template <typename T>
    void num(T v) {
        const size_t at = out.size();
        out.resize(at + sizeof(T));
        std::memcpy(out.data() + at, &v, sizeof(T));
    }
    template <typename T>
    void vec(const std::vector<T>& v) {
        num<uint64_t>(v.size());
        const size_t at = out.size();
        out.resize(at + v.size() * sizeof(T));
        if (!v.empty()) std::memcpy(out.data() + at, v.data(), v.size() * sizeof(T));
    }
    void str(std::string_view s) {
        num<uint64_t>(s.size());
        out.insert(out.end(), s.begin(), s.end());
    }
    void strs(const std::vector<std::string>& v) {
        num<uint64_t>(v.size());
        for (const std::string& s : v) str(s);
    }
};

void append_utf8(std::string& s, uint32_t cp) {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else { s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
}

// ---------------------------------------------------------------- crawdad 0.3 trie blob -> (surface, value) records
//
//   u32 table_len, table_len x u32 (code point -> code, 0xFFFFFFFF = unmapped), u32 alphabet_size,
//   u32 n_nodes, n_nodes x { u32 base, u32 check }                                    (all little endian)
//   child(n, code) = (base[n] & MASK) ^ code, valid iff (check[child] & MASK) == n and n is not a leaf;
//   leaf: base top bit set, value = base & MASK; a key that ends at an inner node n sets check[n]'s top bit and keeps its
//   value in the leaf child on code 0 (the end marker, never a character of a key); root = node 0.
struct CrawdadRecord {
    std::string surface;
    uint32_t value;
};

std::vector<CrawdadRecord> crawdad_records(const std::vector<uint8_t>& blob, const char* name) {
    const std::string where = std::string(name) + " trie: ";
    Reader r{blob.data(), blob.size()};
    auto u32s = [&](size_t k, size_t each) {
        if (k > (r.n - r.pos) / each) bad(where + "a length field exceeds the blob");
    };
    const uint32_t table_len = r.num<uint32_t>();
    u32s(table_len, 4);
    std::vector<uint32_t> table(table_len);
    if (table_len) std::memcpy(table.data(), r.p + r.pos, (size_t)table_len * 4);
    r.pos += (size_t)table_len * 4;
    const uint32_t alphabet = r.num<uint32_t>();
    const uint32_t n_nodes = r.num<uint32_t>();
    u32s(n_nodes, 8);
    if (r.pos + (size_t)n_nodes * 8 != r.n) bad(where + "blob length != 12 + 4*table_len + 8*n_nodes");
    if (table_len > 0x110000) bad(where + "code table larger than the Unicode range");
    // codes are < alphabet and distinct per code point, so a sane alphabet has at most one code per table entry plus the end
    // marker: an unchecked u32 here would size the inverse table below (16 GiB for 0xFFFFFFFF) before anything else is looked at
    if ((uint64_t)alphabet > (uint64_t)table_len + 1) bad(where + "alphabet size exceeds the code table");
    std::vector<uint32_t> base(n_nodes), check(n_nodes);
    for (uint32_t i = 0; i < n_nodes; ++i) {
        std::memcpy(&base[i], r.p + r.pos + (size_t)i * 8, 4);
        std::memcpy(&check[i], r.p + r.pos + (size_t)i * 8 + 4, 4);
    }
    std::vector<uint32_t> code_to_cp(alphabet, kInvalidCode);  // code 0 = end marker (code point 0)
    for (uint32_t cp = 0; cp < table_len; ++cp) {
        const uint32_t c = table[cp];
        if (c == kInvalidCode) continue;
        if (c >= alphabet || code_to_cp[c] != kInvalidCode) bad(where + "the code table is not a bijection onto the alphabet");
        if (c == 0 && cp != 0) bad(where + "a character shares the end code");
        code_to_cp[c] = cp;
    }
    // code 0 is the end marker whether or not the table maps U+0000 to it (crawdad's builder does; a table without that entry
    // describes the same trie)
    if (alphabet) code_to_cp[0] = 0;
    std::vector<CrawdadRecord> out;
    if (n_nodes == 0) return out;
    // children by inverting `check` (one pass), then a walk from the root; only nodes reached from the root count
    std::vector<uint32_t> first(n_nodes, kInvalidCode), next(n_nodes, kInvalidCode), code_of(n_nodes, 0), parent(n_nodes, kInvalidCode);
    for (uint32_t i = n_nodes; i-- > 1;) {
        const uint32_t p = check[i] & kOffsetMask;
        if (p >= n_nodes || p == i || (base[p] >> 31)) continue;  // vacant slot, or the "parent" is a leaf
        const uint32_t c = (base[p] & kOffsetMask) ^ i;
        if (c >= alphabet || code_to_cp[c] == kInvalidCode) continue;
        code_of[i] = c;
        next[i] = first[p];
        first[p] = i;
    }
    std::vector<uint32_t> stack{0};
    std::vector<uint8_t> seen(n_nodes, 0);
    seen[0] = 1;
    std::vector<uint32_t> cps;
    while (!stack.empty()) {
        const uint32_t n = stack.back();
        stack.pop_back();
        if (base[n] >> 31) {  // leaf: the key is the path, without the end marker
            cps.clear();
            for (uint32_t v = n; v != 0; v = parent[v])
                if (code_of[v] != 0) cps.push_back(code_to_cp[code_of[v]]);
            CrawdadRecord rec;
            for (size_t i = cps.size(); i-- > 0;) append_utf8(rec.surface, cps[i]);
            rec.value = base[n] & kOffsetMask;
            out.push_back(std::move(rec));
            continue;
        }
        for (uint32_t c = first[n]; c != kInvalidCode; c = next[c]) {
            if (seen[c]) bad(where + "a node is reachable twice");
            if (code_of[c] == 0 && !(base[c] >> 31)) bad(where + "an end-marker child is not a leaf");
            if (code_of[c] == 0 && !(check[n] >> 31)) bad(where + "an end-marker child below a node without the has-leaf flag");
            seen[c] = 1;
            parent[c] = n;
            stack.push_back(c);
        }
    }
    return out;
}

// (surface, value) records -> crawdad blob.  Same layout as above; node placement is this builder's own (first fit over a free
// list), so the bytes differ from what crawdad's builder would emit for the same keys while every documented invariant holds.
std::vector<uint8_t> crawdad_blob(const std::vector<std::u32string>& keys, const std::vector<uint32_t>& values) {
    // code mapper: end marker = 0, characters by descending frequency (ties: ascending code point)
    uint32_t max_cp = 0;
    std::vector<std::pair<uint32_t, uint32_t>> freq;  // (cp, count)
    {
        std::vector<uint32_t> all;
        for (const auto& k : keys) for (char32_t c : k) all.push_back(c);
        std::sort(all.begin(), all.end());
        for (size_t i = 0; i < all.size();) {
            size_t j = i;
            while (j < all.size() && all[j] == all[i]) ++j;
            freq.push_back({all[i], (uint32_t)(j - i)});
            i = j;
        }
        if (!all.empty()) max_cp = all.back();
    }
    std::sort(freq.begin(), freq.end(), [](auto& a, auto& b) { return a.second != b.second ? a.second > b.second : a.first < b.first; });
    std::vector<uint32_t> table((size_t)max_cp + 1, kInvalidCode);
    table[0] = 0;
    for (size_t i = 0; i < freq.size(); ++i) table[freq[i].first] = (uint32_t)i + 1;
    const uint32_t alphabet = (uint32_t)freq.size() + 1;
    uint32_t block = 256;
    while (block < alphabet) block <<= 1;

    // sorted keys as code strings with the end marker appended: a prefix-free set, every key ends in a leaf
    std::vector<uint32_t> order(keys.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::vector<std::vector<uint32_t>> codes(keys.size());
    for (size_t i = 0; i < keys.size(); ++i) {
        for (char32_t c : keys[i]) codes[i].push_back(table[c]);
        codes[i].push_back(0);
    }
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return codes[a] < codes[b]; });

    std::vector<uint32_t> base, check;
    std::vector<uint8_t> used;
    uint32_t hint = 0;  // every slot below is taken
    auto grow = [&] {
        base.resize(base.size() + block, kOffsetMask);  // vacant: base = check = MASK
        check.resize(check.size() + block, kOffsetMask);
        used.resize(used.size() + block, 0);
    };
    grow();
    used[0] = 1;
    base[0] = 0;
    check[0] = kOffsetMask;
    struct Frame { uint32_t node, lo, hi, depth; };
    std::vector<Frame> stack;
    if (!keys.empty()) stack.push_back({0, 0, (uint32_t)keys.size(), 0});
    std::vector<uint32_t> labels, starts;
    while (!stack.empty()) {
        const Frame f = stack.back();
        stack.pop_back();
        labels.clear();
        starts.clear();
        for (uint32_t i = f.lo; i < f.hi;) {
            const uint32_t c = codes[order[i]][f.depth];
            labels.push_back(c);
            starts.push_back(i);
            while (i < f.hi && codes[order[i]][f.depth] == c) ++i;
        }
        starts.push_back(f.hi);
        // first base (a multiple-free XOR base) whose child slots are all free
        while (hint < used.size() && used[hint]) ++hint;
        uint32_t b = 0;
        for (uint32_t e = hint;; ++e) {
            if (e >= used.size()) grow();
            if (used[e]) continue;
            b = e ^ labels[0];
            bool ok = true;
            for (uint32_t c : labels) {
                const uint32_t s = b ^ c;
                while (s >= used.size()) grow();
                if (used[s]) { ok = false; break; }
            }
            if (ok) break;
        }
        base[f.node] = b | (base[f.node] & ~kOffsetMask);
        for (size_t j = 0; j < labels.size(); ++j) {
            const uint32_t child = b ^ labels[j];
            used[child] = 1;
            check[child] = f.node;
            base[child] = 0;
            if (labels[j] == 0) {  // end marker: leaf with the value; the parent carries the has-leaf flag
                base[child] = values[order[starts[j]]] | ~kOffsetMask;
                check[f.node] |= ~kOffsetMask;
            }
        }
        for (size_t j = labels.size(); j-- > 0;)
            if (labels[j] != 0) stack.push_back({b ^ labels[j], starts[j], starts[j + 1], f.depth + 1});
    }
    Writer w;
    w.num<uint32_t>((uint32_t)table.size());
    for (uint32_t t : table) w.num<uint32_t>(t);
    w.num<uint32_t>(alphabet);
    w.num<uint32_t>((uint32_t)base.size());
    for (size_t i = 0; i < base.size(); ++i) { w.num<uint32_t>(base[i]); w.num<uint32_t>(check[i]); }
    return std::move(w.out);
}

// ---------------------------------------------------------------- Lexicon <-> its serialized parts

void read_lexicon(Reader& r, Lexicon& lx, uint32_t expect_type, const char* name) {
    const std::vector<uint8_t> blob = r.vec<uint8_t>();
    const std::vector<uint32_t> postings = r.vec<uint32_t>();
    const size_t n_params = r.len(6);
    std::vector<WordParam> params(n_params);
    for (WordParam& p : params) { p.left_id = r.num<uint16_t>(); p.right_id = r.num<uint16_t>(); p.word_cost = r.num<int16_t>(); }
    std::vector<std::string> features = r.strs();
    const uint32_t lex_type = r.num<uint32_t>();
    if (lex_type != expect_type) bad(std::string(name) + ": unexpected lex_type");
    if (features.size() != params.size()) bad(std::string(name) + ": params and features differ in length");
    // word id -> surface through trie values and postings (map/posting.rs:16-22: data[v] = count, then the ids)
    std::vector<CrawdadRecord> recs = crawdad_records(blob, name);
    std::vector<uint32_t> surface_of(params.size(), 0xFFFFFFFFu);
    for (uint32_t k = 0; k < recs.size(); ++k) {
        const uint32_t v = recs[k].value;
        if (v >= postings.size() || postings[v] == 0 || postings[v] > postings.size() - v - 1) bad(std::string(name) + ": a trie value does not point at a postings list");
        for (uint32_t j = 0; j < postings[v]; ++j) {
            const uint32_t id = postings[v + 1 + j];
            if (id >= params.size() || surface_of[id] != 0xFFFFFFFFu) bad(std::string(name) + ": a word id is missing from or repeated in the postings");
            if (j && id <= postings[v + j]) bad(std::string(name) + ": postings ids are not ascending");
            surface_of[id] = k;
        }
    }
    for (uint32_t s : surface_of)
        if (s == 0xFFFFFFFFu) bad(std::string(name) + ": the trie does not reach every word (crawdad layout mismatch?)");
    lexicon_from_words(lx, [&](uint32_t id) -> const std::string& { return recs[surface_of[id]].surface; }, std::move(params), std::move(features), name);
}

void write_lexicon(Writer& w, const Lexicon& lx, uint32_t lex_type) {
    // WordMapBuilder::build (map.rs:57-73): surfaces in BTreeMap order (byte order of UTF-8 = code point order), one postings
    // list per surface, the trie value is the list's offset
    std::vector<std::u32string> keys;
    std::vector<std::pair<uint32_t, uint32_t>> spans;  // entries[val .. val + cnt)
    lexicon_keys(lx, keys, spans);
    std::vector<uint32_t> order(keys.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
    std::vector<std::u32string> sorted_keys;
    std::vector<uint32_t> values, postings;
    for (uint32_t k : order) {
        values.push_back((uint32_t)postings.size());
        postings.push_back(spans[k].second);
        for (uint32_t j = 0; j < spans[k].second; ++j) postings.push_back(lx.entries[spans[k].first + j].word_id);
        sorted_keys.push_back(std::move(keys[k]));
    }
    w.vec(crawdad_blob(sorted_keys, values));
    w.vec(postings);
    w.num<uint64_t>(lx.params.size());
    for (const WordParam& p : lx.params) { w.num<uint16_t>(p.left_id); w.num<uint16_t>(p.right_id); w.num<int16_t>(p.word_cost); }
    w.strs(lx.features);
    w.num<uint32_t>(lex_type);
}

void read_scorer(Reader& r, Scorer& s) {
    s.bases = r.vec<uint32_t>();
    s.checks = r.vec<uint32_t>();
    s.costs = r.vec<int32_t>();
    if (s.checks.size() != s.costs.size()) bad("scorer: checks and costs differ in length");
}
void write_scorer(Writer& w, const Scorer& s) { w.vec(s.bases); w.vec(s.checks); w.vec(s.costs); }

std::vector<uint32_t> read_u31x8(Reader& r) {  // Vec<U31x8>: count of 8-wide blocks, 32 bytes each
    const size_t k = r.len(32);
    std::vector<uint32_t> v(k * 8);
    if (k) std::memcpy(v.data(), r.p + r.pos, k * 32);
    r.pos += k * 32;
    for (uint32_t x : v)
        if (x > 0x7FFFFFFFu) bad("connector: a feature id is not a U31");  // num.rs:44-52
    return v;
}
void write_u31x8(Writer& w, const std::vector<uint32_t>& v) {
    w.num<uint64_t>(v.size() / 8);
    const size_t at = w.out.size();
    w.out.resize(at + v.size() * 4);
    if (!v.empty()) std::memcpy(w.out.data() + at, v.data(), v.size() * 4);
}

void read_matrix(Reader& r, std::vector<int16_t>& data, uint32_t& num_right, uint32_t& num_left) {
    data = r.vec<int16_t>();
    const uint64_t nr = r.num<uint64_t>(), nl = r.num<uint64_t>();
    if (nr > 0xFFFF || nl > 0xFFFF || nr * nl != data.size()) bad("connector: matrix dimensions do not match its data");
    num_right = (uint32_t)nr;
    num_left = (uint32_t)nl;
}

// ---------------------------------------------------------------- zstd through the system's libzstd.so.1

struct Zstd {
    void* h = nullptr;
    size_t (*decompressStream)(void*, void*, void*) = nullptr;
    void* (*createDStream)() = nullptr;
    size_t (*freeDStream)(void*) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    const char* (*getErrorName)(size_t) = nullptr;
    size_t (*compressBound)(size_t) = nullptr;
    size_t (*compress)(void*, size_t, const void*, size_t, int) = nullptr;
};
struct ZBuf { const void* p; size_t size, pos; };  // ZSTD_inBuffer / ZSTD_outBuffer (same shape, the stable streaming ABI)

const Zstd& zstd() {
    static Zstd z;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"libzstd.so.1", "libzstd.so"}) {
            z.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (z.h) break;
        }
        if (!z.h) return;
        auto sym = [&](const char* s) { return dlsym(z.h, s); };
        z.decompressStream = reinterpret_cast<size_t (*)(void*, void*, void*)>(sym("ZSTD_decompressStream"));
        z.createDStream = reinterpret_cast<void* (*)()>(sym("ZSTD_createDStream"));
        z.freeDStream = reinterpret_cast<size_t (*)(void*)>(sym("ZSTD_freeDStream"));
        z.isError = reinterpret_cast<unsigned (*)(size_t)>(sym("ZSTD_isError"));
        z.getErrorName = reinterpret_cast<const char* (*)(size_t)>(sym("ZSTD_getErrorName"));
        z.compressBound = reinterpret_cast<size_t (*)(size_t)>(sym("ZSTD_compressBound"));
        z.compress = reinterpret_cast<size_t (*)(void*, size_t, const void*, size_t, int)>(sym("ZSTD_compress"));
        if (!z.decompressStream || !z.createDStream || !z.freeDStream || !z.isError || !z.getErrorName || !z.compressBound || !z.compress) z.h = nullptr;
    });
    if (!z.h) throw Error(VBT_ERR_UNSUPPORTED, "zstd: libzstd.so.1 is not available on this system (pass the decompressed system.dic)");
    return z;
}

bool is_zstd_frame(const uint8_t* p, size_t n) { return n >= 4 && p[0] == 0x28 && p[1] == 0xB5 && p[2] == 0x2F && p[3] == 0xFD; }

std::vector<uint8_t> zstd_decompress(const uint8_t* data, size_t len) {
    const Zstd& z = zstd();
    void* ds = z.createDStream();
    if (!ds) throw std::bad_alloc();
    std::vector<uint8_t> out(std::max<size_t>(len * 4, 1 << 20));
    ZBuf in{data, len, 0};
    size_t produced = 0;
    for (;;) {
        if (produced == out.size()) {
            if (out.size() >= (1ull << 34)) { z.freeDStream(ds); throw Error(VBT_ERR_INVALID_FORMAT, "zstd: more than 16 GiB of output (not a dictionary)"); }
            out.resize(out.size() * 2);
        }
        struct { void* p; size_t size, pos; } ob{out.data(), out.size(), produced};
        const size_t rc = z.decompressStream(ds, &ob, &in);
        produced = ob.pos;
        if (z.isError(rc)) {
            const std::string msg = z.getErrorName(rc);
            z.freeDStream(ds);
            throw Error(VBT_ERR_INVALID_FORMAT, "zstd: " + msg);
        }
        if (rc == 0 && in.pos == in.size) break;       // end of the last frame
        if (in.pos == in.size && produced < out.size()) {  // input exhausted inside a frame
            z.freeDStream(ds);
            throw Error(VBT_ERR_INVALID_FORMAT, "zstd: truncated frame");
        }
    }
    z.freeDStream(ds);
    out.resize(produced);
    return out;
}

}  // namespace

std::vector<uint8_t> zstd_compress(const uint8_t* data, size_t len, int level) {
    const Zstd& z = zstd();
    std::vector<uint8_t> out(z.compressBound(len));
    const size_t rc = z.compress(out.data(), out.size(), data, len, level);
    if (z.isError(rc)) throw Error(VBT_ERR_INVALID_FORMAT, std::string("zstd: ") + z.getErrorName(rc));
    out.resize(rc);
    return out;
}

// Dictionary::read (dictionary.rs:173-197); a zstd frame around it is unwrapped first (tokenize/src/main.rs:59-60)
Dictionary* read_dictionary(const uint8_t* data, size_t len) {
    std::vector<uint8_t> inflated;
    if (is_zstd_frame(data, len)) {
        inflated = zstd_decompress(data, len);
        data = inflated.data();
        len = inflated.size();
    }
    if (len < kMagicLen || std::memcmp(data, kMagic, kMagicLen) != 0)
        throw Error(VBT_ERR_INVALID_ARGUMENT, "rdr: The magic number of the input model mismatches.");  // dictionary.rs:188-193
    Reader r{data + kMagicLen, len - kMagicLen};
    auto d = std::make_unique<Dictionary>();
    read_lexicon(r, d->system, 0, "system_lexicon");
    if (r.option()) {
        read_lexicon(r, d->user, 1, "user_lexicon");
        d->has_user = true;
    }
    const uint32_t kind = r.num<uint32_t>();
    if (kind == kConnMatrix) {
        read_matrix(r, d->matrix, d->num_right, d->num_left);
    } else if (kind == kConnRaw) {
        d->raw.right_feats = read_u31x8(r);
        d->raw.left_feats = read_u31x8(r);
        const uint64_t blocks = r.num<uint64_t>();  // feat_template_size / SIMD_SIZE, raw_connector.rs:95
        if (blocks == 0 || blocks > 0x1000 || d->raw.right_feats.size() % (blocks * 8) || d->raw.left_feats.size() % (blocks * 8))
            bad("connector: feature rows do not match the template size");
        d->raw.width = (uint32_t)blocks * 8;
        read_scorer(r, d->raw.scorer);
        const size_t nr = d->raw.right_feats.size() / d->raw.width, nl = d->raw.left_feats.size() / d->raw.width;
        if (nr > 0xFFFF || nl > 0xFFFF || nr == 0 || nl == 0) bad("connector: invalid number of connection ids");
        d->num_right = (uint32_t)nr;
        d->num_left = (uint32_t)nl;
    } else if (kind == kConnDual) {
        read_matrix(r, d->dual.matrix, d->dual.m_num_right, d->dual.m_num_left);
        d->dual.right_map = r.vec<uint16_t>();
        d->dual.left_map = r.vec<uint16_t>();
        d->dual.right_feats = read_u31x8(r);
        d->dual.left_feats = read_u31x8(r);
        read_scorer(r, d->dual.scorer);
        if (d->dual.right_feats.size() != d->dual.right_map.size() * 8 || d->dual.left_feats.size() != d->dual.left_map.size() * 8 ||
            d->dual.right_map.empty() || d->dual.left_map.empty() || d->dual.right_map.size() > 0xFFFF || d->dual.left_map.size() > 0xFFFF)
            bad("connector: dual connector arrays differ in length");
        for (uint16_t m : d->dual.right_map) if (m >= d->dual.m_num_right) bad("connector: dual id map out of range");
        for (uint16_t m : d->dual.left_map) if (m >= d->dual.m_num_left) bad("connector: dual id map out of range");
        d->num_right = (uint32_t)d->dual.right_map.size();
        d->num_left = (uint32_t)d->dual.left_map.size();
    } else {
        bad("unknown connector variant");
    }
    d->conn_kind = (int)kind;
    if (r.option()) {
        d->mapper_left = r.vec<uint16_t>();
        d->mapper_right = r.vec<uint16_t>();
        if (d->mapper_left.size() != d->num_left || d->mapper_right.size() != d->num_right) bad("mapper: size differs from the connector's");
    }
    d->chr2inf = r.vec<uint32_t>();
    d->categories = r.strs();
    if (d->chr2inf.size() != 65536) bad("char_prop: chr2inf must have 65536 entries");
    if (d->categories.empty() || d->categories.size() > 18) bad("char_prop: invalid number of categories");
    for (uint32_t v : d->chr2inf) {  // CharInfo (character.rs:10-24): the kernels index the unknown-word table with base_id
        const uint32_t idset = v & 0x3FFFFu, base_id = (v >> 18) & 0xFFu;
        if (base_id >= d->categories.size() || (idset >> d->categories.size()) != 0) bad("char_prop: a CharInfo names a category that does not exist");
    }
    {
        const std::vector<uint64_t> offsets = r.vec<uint64_t>();
        const size_t n_entries = r.len(16);
        if (offsets.size() != d->categories.size() + 1 || offsets.back() != n_entries || n_entries > 0xFFFF) bad("unk_handler: offsets do not match the categories / entries");
        size_t cat = 0;
        for (size_t i = 0; i < n_entries; ++i) {
            const uint16_t cate = r.num<uint16_t>(), left = r.num<uint16_t>(), right = r.num<uint16_t>();
            const int16_t cost = r.num<int16_t>();
            while (cat + 1 < offsets.size() && offsets[cat + 1] <= i) ++cat;
            if (cate != cat || offsets[cat] > i) bad("unk_handler: entries are not grouped by category");
            d->unk_entries.push_back(Entry{(uint32_t)i, (uint32_t)left | ((uint32_t)right << 16), (uint32_t)(uint16_t)cost});
            d->unk_features.push_back(r.str());
        }
        for (size_t i = 0; i + 1 < offsets.size(); ++i)
            if (offsets[i] > offsets[i + 1]) bad("unk_handler: offsets are not ascending");
        for (uint64_t o : offsets) d->unk_offsets.push_back((uint32_t)o);
    }
    if (r.pos != r.n) bad("trailing bytes after the dictionary");
    // the checks of the builder (builder.rs:24-35) and of the device image (category bits of CharInfo vs categories)
    verify_dictionary_ids(*d);
    return d.release();
}

// Dictionary::write (dictionary.rs:142-150)
std::vector<uint8_t> write_dictionary(const Dictionary& d) {
    Writer w;
    w.out.assign(kMagic, kMagic + kMagicLen);
    write_lexicon(w, d.system, 0);
    w.num<uint8_t>(d.has_user ? 1 : 0);
    if (d.has_user) write_lexicon(w, d.user, 1);
    w.num<uint32_t>((uint32_t)d.conn_kind);
    if (d.conn_kind == kConnMatrix) {
        w.vec(d.matrix);
        w.num<uint64_t>(d.num_right);
        w.num<uint64_t>(d.num_left);
    } else if (d.conn_kind == kConnRaw) {
        write_u31x8(w, d.raw.right_feats);
        write_u31x8(w, d.raw.left_feats);
        w.num<uint64_t>(d.raw.width / 8);
        write_scorer(w, d.raw.scorer);
    } else {
        w.vec(d.dual.matrix);
        w.num<uint64_t>(d.dual.m_num_right);
        w.num<uint64_t>(d.dual.m_num_left);
        w.vec(d.dual.right_map);
        w.vec(d.dual.left_map);
        write_u31x8(w, d.dual.right_feats);
        write_u31x8(w, d.dual.left_feats);
        write_scorer(w, d.dual.scorer);
    }
    w.num<uint8_t>(d.mapper_left.empty() ? 0 : 1);
    if (!d.mapper_left.empty()) { w.vec(d.mapper_left); w.vec(d.mapper_right); }
    w.vec(d.chr2inf);
    w.strs(d.categories);
    w.num<uint64_t>(d.unk_offsets.size());
    for (uint32_t o : d.unk_offsets) w.num<uint64_t>(o);
    w.num<uint64_t>(d.unk_entries.size());
    size_t cat = 0;
    for (size_t i = 0; i < d.unk_entries.size(); ++i) {
        while (cat + 1 < d.unk_offsets.size() && d.unk_offsets[cat + 1] <= i) ++cat;
        const Entry& e = d.unk_entries[i];
        w.num<uint16_t>((uint16_t)cat);
        w.num<uint16_t>((uint16_t)(e.left_right & 0xFFFF));
        w.num<uint16_t>((uint16_t)(e.left_right >> 16));
        w.num<int16_t>((int16_t)(uint16_t)e.cost);
        w.str(d.unk_features[i]);
    }
    return std::move(w.out);
}

}  // namespace vbt
