// Device engine of the MI355X tokenizer (product code): device dictionary image,
// batch workspace and the kernel launch sequence.  See DESIGN.md for the layout.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "dict.hpp"

namespace vbt {

// Device view of one lexicon (trie + entries); all pointers are device pointers.
struct DevLexicon {
    const uint16_t* mapper;
    uint32_t mapper_len;
    const TrieNode* nodes;
    uint32_t root_base;
    const Entry* entries;
};

// Device view of the dictionary + tokenizer options (passed to kernels by value).
struct DevDict {
    DevLexicon sys, user;
    int has_user;
    const int16_t* matrix;  // [num_left][num_right]
    uint32_t num_right;
    const uint32_t* chr2inf;      // 65536 packed CharInfo
    const uint32_t* unk_off;      // n_categories + 1
    const Entry* unk_entries;
    uint32_t space_cateset;       // 0 when ignore_space is off (tokenizer.rs:16,50)
    uint32_t max_grouping_len;    // 0xFFFFFFFF = unlimited (tokenizer.rs:67-74)
};

// Per-call arguments of the tokenize kernels.
struct BatchArgs {
    const uint8_t* text;
    const uint64_t* offsets;
    uint32_t n;
    // outputs
    vbt_token_rec* tokens;
    uint32_t tok_cap;
    uint32_t* tok_off;
    uint32_t* tok_cnt;
    // control block (device): [0]=total tokens [1]=overflow0 count [2]=overflow1 count
    // [3]=cursor tier1 [4]=cursor tier2 [5]=error flags [6..7]=scratch bump (u64)
    uint32_t* ctrl;
    uint32_t* overflow0;  // sentences that did not fit tier 0
    uint32_t* overflow1;  // sentences that did not fit tier 1
    // global scratch arena for tier 2
    char* scratch;
    uint64_t scratch_bytes;
};

enum CtrlSlot { kTotal = 0, kOver0 = 1, kOver1 = 2, kCursor1 = 3, kCursor2 = 4, kError = 5, kBump = 6, kCtrlWords = 8 };
enum DevError { kErrTokCap = 1, kErrScratch = 2, kErrTooLong = 4 };

class Tokenizer {
  public:
    // Builds the device image of `dict` (borrowed until adopt() hands over ownership).
    Tokenizer(const Dictionary* dict, bool ignore_space, uint32_t max_grouping_len, int device);
    ~Tokenizer();
    void adopt(std::unique_ptr<Dictionary> d) { owned_ = std::move(d); }
    const Dictionary& dict() const { return *dict_; }
    const DevDict& dev() const { return dev_; }
    int device() const { return device_; }

  private:
    void upload_lexicon(const Lexicon& lx, DevLexicon& out);
    const Dictionary* dict_;
    std::unique_ptr<Dictionary> owned_;
    DevDict dev_{};
    int device_ = 0;
    std::vector<void*> allocs_;
};

class Workspace {
  public:
    Workspace(const Tokenizer& tok, uint64_t max_sentences, uint64_t max_bytes);
    ~Workspace();
    void run(const uint8_t* d_text, const uint64_t* d_offsets, uint64_t n, uint64_t total_bytes, void* stream);
    void stats(vbt_call_stats* out);  // synchronizes the last stream used

    const Tokenizer& tok;
    uint64_t max_sentences, max_bytes;
    vbt_token_rec* d_tokens = nullptr;
    uint32_t *d_tok_off = nullptr, *d_tok_cnt = nullptr, *d_ctrl = nullptr, *d_over0 = nullptr, *d_over1 = nullptr;
    char* d_scratch = nullptr;
    uint64_t scratch_bytes = 0;
    uint32_t lds0 = 0, lds1 = 0;
    bool timing = false;
    uint64_t last_n = 0;
    void* last_stream = nullptr;
    void* ev[4] = {nullptr, nullptr, nullptr, nullptr};
};

}  // namespace vbt
