// Host-side dictionary model of the MI355X tokenizer (product code).
//
// Parses MeCab-format sources exactly as vibrato's SystemDictionaryBuilder does
// (reference: vibrato/src/dictionary/builder.rs:64-89) and lays the result out in
// the flat arrays the HIP kernels read ("device image").
#pragma once
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "../../include/vibrato_hip.h"

namespace vbt {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// One double-array node as the GPU reads it: a single 16-byte load per transition.
//   child(n, c) = nodes[n].base ^ c, valid iff nodes[child].check == n
//   nodes[n].cnt > 0  <=> some word ends at n; its entries are entries[val .. val+cnt)
struct alignas(16) TrieNode {
    uint32_t base;
    uint32_t check;
    uint32_t val;
    uint32_t cnt;
};
static_assert(sizeof(TrieNode) == 16, "TrieNode must be 16 bytes");

// One lexicon / unknown-word entry: what Lattice::insert_node needs, 12 bytes.
//   w0 = word_id, w1 = left_id | right_id << 16, w2 = (uint16) word_cost
struct Entry {
    uint32_t word_id;
    uint32_t left_right;
    uint32_t cost;
};
static_assert(sizeof(Entry) == 12, "Entry must be 12 bytes");

struct WordParam {
    uint16_t left_id, right_id;
    int16_t word_cost;
};

// A lexicon (system or user): trie + entries + per-word parameters and features.
// Mirrors vibrato::dictionary::lexicon::Lexicon (lexicon.rs:23-29).
struct Lexicon {
    std::vector<uint16_t> mapper;   // code point -> trie code (0 = not in any key)
    uint32_t alphabet = 0;          // codes are 1..alphabet
    std::vector<TrieNode> nodes;    // double array, root = 0
    std::vector<Entry> entries;     // grouped by surface, word ids ascending
    std::vector<WordParam> params;  // indexed by word id
    std::vector<std::string> features;
    uint32_t max_word_chars = 0;

    // Host mirror of the device walk; used by tests (Lexicon::common_prefix_iterator,
    // lexicon.rs:33-46). Appends (word_id, end_char) pairs in enumeration order.
    void common_prefix(const uint32_t* cps, size_t n, std::vector<std::pair<uint32_t, uint32_t>>& out) const;
};

// Scorer of the compact connectors (connector/raw_connector/scorer.rs:170-178): a double-array hash from a pair of
// feature ids to a cost: pos = bases[key1] ^ key2, valid iff checks[pos] == key1.
struct Scorer {
    std::vector<uint32_t> bases, checks;
    std::vector<int32_t> costs;
};

// RawConnector (connector/raw_connector.rs:22-27): per connection id one row of `width` feature ids (a multiple of 8, the
// reference's U31x8 vectors flattened; 0x7FFFFFFF = no feature; row 0 = BOS/EOS, all zero).
struct RawConnector {
    std::vector<uint32_t> right_feats, left_feats;
    uint32_t width = 0;
    Scorer scorer;
};

// DualConnector (connector/dual_connector.rs:16-23): a small matrix over mapped ids plus one 8-wide raw row per id.
struct DualConnector {
    std::vector<int16_t> matrix;  // data[left * m_num_right + right] over the mapped ids
    uint32_t m_num_right = 0, m_num_left = 0;
    std::vector<uint16_t> right_map, left_map;
    std::vector<uint32_t> right_feats, left_feats;  // 8 per id
    Scorer scorer;
};

enum ConnKind { kConnMatrix = 0, kConnRaw = 1, kConnDual = 2 };  // ConnectorWrapper, connector.rs:30-35 (bincode variant order)

struct Dictionary {
    int conn_kind = kConnMatrix;
    RawConnector raw;
    DualConnector dual;
    Lexicon system;
    bool has_user = false;
    Lexicon user;
    std::vector<int16_t> matrix;  // data[left * num_right + right], matrix_connector.rs:47
    uint32_t num_right = 0, num_left = 0;
    std::vector<uint32_t> chr2inf;  // 65536 packed CharInfo, character.rs:10-24
    std::vector<std::string> categories;
    std::vector<uint32_t> unk_offsets;  // per category, unknown.rs:63-66
    std::vector<Entry> unk_entries;     // word_id = row index
    std::vector<std::string> unk_features;
    // ConnIdMapper of the last map_connection_ids call (mapper.rs:9-12); empty = none.
    // new id = mapper_left[old id]; applied to user lexicons attached later (dictionary.rs:214-217).
    std::vector<uint16_t> mapper_left, mapper_right;

    int cate_id(std::string_view name) const;
};

// SystemDictionaryBuilder::from_readers (builder.rs:64-89). If matrix_bin != nullptr the
// connection matrix comes from a binary i16 array (data[left*num_right+right]).
Dictionary* build_dictionary(std::string_view lex, std::string_view matrix_def, const int16_t* matrix_bin,
                             uint32_t num_right, uint32_t num_left, std::string_view char_def,
                             std::string_view unk_def);

// SystemDictionaryBuilder::from_readers_with_bigram_info (builder.rs:111-160): the connector comes from bigram.right /
// bigram.left / bigram.cost: a RawConnector, or (dual) a DualConnector built like the reference's (connector.cpp).  The device
// materialises either as a dense matrix when the tokenizer is created.
Dictionary* build_dictionary_bigram(std::string_view lex, std::string_view bigram_right, std::string_view bigram_left,
                                    std::string_view bigram_cost, std::string_view char_def, std::string_view unk_def, bool dual);

// ConnectorCost::cost(right_id, left_id) (connector.rs:25-28) for any connector kind, on the host.
int32_t conn_cost(const Dictionary& d, uint32_t right_id, uint32_t left_id);
// Scorer::accumulate_cost (scorer.rs:327-345, the scalar path): sum of the costs of the (key1[i], key2[i]) pairs present.
int32_t scorer_accumulate(const Scorer& s, const uint32_t* keys1, const uint32_t* keys2, size_t n);
// utils::parse_csv_row (utils.rs:40-61)
std::vector<std::string> parse_csv_row(std::string_view row);
// helpers shared by dict.cpp and connector.cpp
void finish_dictionary(Dictionary& d, std::string_view lex, std::string_view char_def, std::string_view unk_def);
void map_connector_ids(Dictionary& d, const std::vector<uint16_t>& ml, const std::vector<uint16_t>& mr);

// Dictionary::map_connection_ids_from_iter (dictionary.rs:245-259): the i-th item (1-origin) of lmap / rmap
// is the OLD id that becomes new id i (ConnIdMapper::parse, mapper.rs:49-80).
void map_connection_ids(Dictionary& d, const uint16_t* lmap, size_t n_lmap, const uint16_t* rmap, size_t n_rmap);

// Dictionary::read / Dictionary::write (dictionary.rs:142-197): vibrato's binary `system.dic`, a zstd frame around it is
// unwrapped on reading (dictio.cpp).
Dictionary* read_dictionary(const uint8_t* data, size_t len);
std::vector<uint8_t> write_dictionary(const Dictionary& d);
std::vector<uint8_t> zstd_compress(const uint8_t* data, size_t len, int level);
// helpers of the reader / writer that live next to the double-array builder (dict.cpp)
void lexicon_from_words(Lexicon& lx, const std::function<const std::string&(uint32_t)>& surface, std::vector<WordParam> params,
                        std::vector<std::string> features, const char* name);
void lexicon_keys(const Lexicon& lx, std::vector<std::u32string>& keys, std::vector<std::pair<uint32_t, uint32_t>>& spans);
void verify_dictionary_ids(const Dictionary& d);

// Rust `str` validity (strict UTF-8: no overlongs, surrogates or code points above U+10FFFF).
bool valid_utf8(const uint8_t* s, size_t n);

// Dictionary::reset_user_lexicon_from_reader (dictionary.rs:209-229).
void set_user_lexicon(Dictionary& d, const char* csv, size_t len);

}  // namespace vbt
