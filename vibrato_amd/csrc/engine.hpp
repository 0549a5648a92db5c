// Device engine of the MI355X tokenizer (product code): device dictionary image,
// batch workspace and the kernel launch sequence.  See DESIGN.md for the layout.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <thread>
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
    const int16_t* matrix;  // [num_left][num_right]; i32 cells when matrix_wide (a compact connector whose costs leave i16)
    uint32_t num_right;
    uint32_t matrix_bytes;        // bytes of the matrix (< 2^32: checked at upload)
    uint32_t matrix_wide;
    const uint32_t* chr2inf;      // 65536 packed CharInfo
    // what the generators read per character, in ONE 8-byte load: {chr2inf[cp] (character.rs:112-116: cp >= 65536 -> entry 0), code of the
    // system trie | code of the user trie << 16}; cpinfo_len entries for the code points below it + one for everything beyond (round 5:
    // two gathers per character -- 94 of a sentence's ~520 line requests in a kernel that is bound by their number)
    const uint2* cpinfo;
    uint32_t cpinfo_len;
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
    vbt_token_rec* tokens;     // compact result (sentence order)
    vbt_token_rec* tok_stage;  // staging: sentence s writes its tokens at its own slot (sentence_slot)
    uint32_t tok_cap;
    uint32_t* tok_off;
    uint32_t* tok_cnt;
    uint32_t* out_header;  // optional: {n_sentences (u64), n_tokens (u32), 0 ...} of a packed result slot (Workspace::set_packed_output), written with the total
    uint32_t* tile_sums;  // tokens per tile of kScanTile sentences, added up by the kernels that write tok_cnt (zeroed by validate_batch)
    // control block (device, kCtrlWords u32): see CtrlSlot
    uint32_t* ctrl;
    // global scratch arena for the last tier
    char* scratch;
    uint64_t scratch_bytes;
    // optional per-phase cycle counters (kProfSlots x kProfWords u64), nullptr = off
    unsigned long long* prof;
    // ---- two-kernel pipeline: gen_candidates -> lattice_lds (per LDS tier) ----
    // per sentence: ONE 16-byte header written by the generator {characters | bytes << 16, lattice candidates | LDS tier << 16,
    // upper bound of the lattice passes, byte offset relative to the batch}: everything lattice_lds needs to find the sentence's
    // regions, in one load.  Bit 31 of the third word: the generator laid out the sweep's pass records as well (a sentence that is
    // swept whole; the word is then the exact pass count, the records sit at the start of the sentence's hit-staging region)
    uint4* s_hdr;
    // per character slot (sentence s, char i -> slot offsets[s] - offsets[0] + kSentenceSlack * s + i; nb + kSentenceSlack slots per sentence)
    uint16_t* g_c2b;
    uint4* g_pc;        // {cand_off | end-list offset << 16, pass bound | window end << 14 | is_space << 31, lens lo, lens hi}
    // per candidate (insertion order, contiguous per sentence; sentence s owns the node slots
    // [node_factor * slot(s), node_factor * slot(s + 1)): no allocation atomics).
    //   g_cand: 16 bytes, one scattered store by the generator: {first cell of the left id's matrix row, (u16) word_cost |
    //   end-list slot << 16, word_idx, end_char | right_id << 16}
    uint4* g_cand;
    uint4* g_hits;      // staging of the trie hits of gen_candidates, same per-sentence regions as g_cand
    uint32_t node_factor;
    uint8_t* s_tier;    // LDS tier chosen by gen_candidates (0xFF = none / empty sentence)
    // work lists: list t (t < n_tiers) feeds LDS tier t, list n_tiers the global-memory fallback (fused kernel),
    // lists n_tiers + 1 .. n_tiers + kGenLevels the large-LDS instances of gen_candidates
    uint32_t* lists;
    uint32_t list_stride;
    uint32_t n_tiers;
    uint32_t seg_tier;     // LDS tier that sweeps longer sentences in segments (>= n_tiers: none)
    uint32_t n_lean;       // the first n_lean LDS tiers are swept by the lean instance (lattice_lean: whole sentences with the generator's pass records ONLY)
    unsigned long long* density_out;  // pinned host memory (device address), {candidates, bytes} of this batch's bulk-generated sentences: written once by build_lists; nullptr = not wanted
    uint32_t inline_lean;  // gen_sweep: the generator's wave sweeps its sentence itself when it routed it to a lean tier (nothing is filed for it)
    uint32_t direct_push;  // gen_one appends to the work lists directly instead of routing through s_tier (no build_lists behind it)
    uint32_t tier_prio;  // the top `tier_prio` LDS tiers run at raised wave priority (0 = off)
    // a launch covers sentences [sid0, sid0 + n); cctrl = the list counters it works with (cctrl[2t] = entries of
    // list t, cctrl[2t+1] = its work cursor); list t starts at lists[t * list_stride + list_off]
    uint32_t sid0;
    uint32_t* cctrl;
    uint32_t list_off;  // offset of this launch's entries inside every list region
    // optional connection-id usage counters (Worker::update_connid_counts, worker.rs:77-93): nullptr = off
    unsigned long long* lid_count;
    unsigned long long* rid_count;
    uint32_t* s_counted;  // per sentence: the steps of positions below this are already counted (a retry must not count them again)
    uint32_t tier_bytes[8];
    uint32_t gen_level_bytes[3];  // LDS of the levels of gen_long (gen_one files what outgrows it at the smallest level that holds it)
};

// ctrl[kTotal] total tokens; ctrl[kError] DevError flags; ctrl[kBump..+1] u64 scratch bump pointer;
// ctrl[kTierCtrl + 2t] = number of sentences tier t passed on, ctrl[kTierCtrl + 2t + 1] = work cursor of tier t
// ctrl[kNodeCursor] bump pointer of the candidate arrays
// ctrl[kDensCand..+1] / ctrl[kDensBytes..+1] u64: candidates / bytes of the sentences the bulk generator took, ctrl[kDensDone] workgroups of
// build_lists that have added theirs (the last one reports the two sums to the tokenizer's pinned slot: BatchArgs::density_out)
enum CtrlSlot { kTotal = 0, kError = 1, kBump = 2, kNodeCursor = 4, kTierCtrl = 6, kDensCand = 8, kDensBytes = 10, kDensDone = 12, kCtrlWords = 32 };
constexpr int kMaxTiers = 8;
constexpr uint32_t kSegTierBytes = 10240;  // default LDS of the sweep's one tier: 16 wavefronts per CU (4 per SIMD, 128 VGPRs each) x 10 KiB = the CU's 160 KiB
constexpr uint32_t kScanBlock = 256, kScanTile = 256;  // token packing: sentences per tile (small tiles: the copy needs the parallelism)
constexpr uint32_t kSentenceSlack = 24;  // character slots per sentence on top of its bytes (see sentence_slot in engine.hip)
constexpr int kGenLevels = 3;  // large-LDS instances of the generator behind the bulk one (32 KiB, 64 KiB, whole CU)
constexpr int kListsBehindTiers = 1 + kGenLevels;  // fallback, generator levels
constexpr int kBlockCtrlWords = 2 * (kMaxTiers + kListsBehindTiers);
constexpr int kProfSlots = 256;  // the counters are spread over this many copies (hot-word atomics serialise)
constexpr int kProfWords = 12;   // kProfPhases cycle totals, sentences, lattice steps, lattice passes, candidates
constexpr int kProfPhases = 8;  // decode, count, fill, end lists, pre-pass, gather, recurrence, emit
// kErrOffsets / kErrUtf8 are set by validate_batch; the batch is then skipped (no kernel touches the per-sentence regions)
enum DevError { kErrTokCap = 1, kErrScratch = 2, kErrTooLong = 4, kErrOffsets = 8, kErrUtf8 = 16, kErrFatal = kErrOffsets | kErrUtf8 };

// What a kernel launch reads of the dictionary.  The connection ids inside it -- the rows and columns of the matrix, the id pairs of
// the entry arrays -- are DEVICE ids: image 0 numbers them as the dictionary does; the image a tokenizer builds from the usage
// counts of its first large batch (Tokenizer::maybe_calibrate) numbers them by descending frequency, the reference's own locality
// lever (map/src/reorder.rs:34-63, matrix_connector.rs:99-116, dictionary.rs:245-259) applied internally.  Nothing a caller sees is
// in device ids: token records carry word ids, the connection-id counters are translated back on their way out.
struct DevImage {
    DevDict dev{};
    uint32_t epoch = 0;
    std::vector<uint16_t> perm_left, perm_right;  // dictionary id -> device id (empty: identity)
    std::vector<void*> allocs;                    // what this image owns on the device
};

// vbt_tokenizer_connid_reorder_info
struct ConnidReorderInfo {
    uint32_t epoch = 0;              // 0: the dictionary's own numbering, 1: renumbered by measured usage
    uint32_t state = 0;              // 0 waiting for a batch of >= min_sentences, 1 running, 2 done, 3 off (VBT_CONNID_REORDER=0 or a failed attempt)
    uint64_t sample_sentences = 0;   // sentences the usage counts were taken on
    uint64_t min_sentences = 0;
    double ms = 0;                   // wall time of the calibration (counting run, sort, new image)
    uint32_t moved_left = 0, moved_right = 0;  // ids whose device id differs from their dictionary id
};

class Workspace;

class Tokenizer {
  public:
    // Builds the device image of `dict` (borrowed until adopt() hands over ownership).
    Tokenizer(const Dictionary* dict, bool ignore_space, uint32_t max_grouping_len, int device);
    ~Tokenizer();
    void adopt(std::unique_ptr<Dictionary> d) { owned_ = std::move(d); }
    const Dictionary& dict() const { return *dict_; }
    // the image new launches use (an image lives as long as the tokenizer: launches in flight and resident Worker kernels keep reading theirs)
    const DevImage& image() const { return *cur_.load(std::memory_order_acquire); }
    const DevImage& image_of(uint32_t epoch) const;
    const DevDict& dev() const { return image().dev; }
    int device() const { return device_; }
    // Called by Workspace::run in front of every batch.  The first batch of >= min_sentences triggers the renumbering of the device
    // image's connection ids by measured usage -- WITHOUT blocking, allocating or synchronising anything on the caller's side: two small
    // kernels on the caller's stream copy a sample (at most VBT_CONNID_SAMPLE = 16384 sentences, spread evenly over the batch) into
    // buffers the tokenizer owns; a background host thread then sweeps the sample once more on a stream of its own with the
    // connection-id counters on (the reorder tool's statistics, worker.rs:77-93, lattice.rs:170-183), sorts the ids by count
    // (mapper.rs:108-146), builds the renumbered image -- permuted matrix, permuted id pairs in the entries -- and publishes it.
    // Launches enqueued before that moment read the old image, later ones the new one; results do not depend on the image.
    void maybe_calibrate(const uint8_t* d_text, const uint64_t* d_offsets, uint64_t n, uint64_t total_bytes, void* stream) const;
    // The same from host buffers, synchronously (vbt_tokenizer_calibrate): for callers who want it done up front.  No-op when a
    // calibration has already happened or is switched off; waits for one that is running.
    void calibrate_host(const uint8_t* text, const uint64_t* offsets, uint64_t n) const;
    // Returns once no calibration is running (state != 1), or after timeout_ms (< 0: no limit); true = not running.
    bool wait_calibration(int64_t timeout_ms) const;
    ConnidReorderInfo reorder_info() const;
    // Lattice density of the text this tokenizer has been seeing: candidates per input byte of the bulk-generated sentences of the last
    // batch that reported (a device kernel stores the two sums into pinned host memory; nothing is ever waited for).  0 = nothing yet.
    // Workspace::run picks the sweep's tier set by it: dense lattices want 10 KiB segments, everything else 8 KiB ones and a fifth wave.
    double candidates_per_byte() const;
    unsigned long long* density_slot_dev() const { return density_dev_; }

  private:
    void upload_lexicon(const Lexicon& lx, DevLexicon& out);
    bool calibrate_sample(uint64_t ns, uint64_t bytes) const;  // counting sweep over the sample buffers + new image; false: sample rejected
    void finish_calibration(bool done, bool give_up) const;
    void background_calibration() const;
    std::unique_ptr<DevImage> renumbered_image(const std::vector<uint16_t>& perm_left, const std::vector<uint16_t>& perm_right, void* stream) const;
    const Dictionary* dict_;
    std::unique_ptr<Dictionary> owned_;
    int device_ = 0;
    std::vector<void*> allocs_;  // what every image shares: tries, code mappers, character classes, unknown-word offsets
    mutable std::mutex img_mu_;
    mutable std::vector<std::unique_ptr<DevImage>> images_;  // [0] = the dictionary's numbering
    mutable std::atomic<const DevImage*> cur_{nullptr};
    mutable std::atomic<int> calib_state_{0};
    mutable ConnidReorderInfo info_;
    uint64_t calib_min_ = 2048, calib_sample_ = 16384;
    // the calibration sample (device memory owned by the tokenizer, allocated with the image when the renumbering is on), the event
    // behind the kernels that fill it, the stream the counting sweep runs on, the thread that does it
    uint8_t* s_text_ = nullptr;
    uint64_t* s_offs_ = nullptr;
    uint32_t *s_src_ = nullptr, *s_info_ = nullptr;
    uint64_t s_cap_bytes_ = 0;
    void *calib_event_ = nullptr, *calib_stream_ = nullptr;
    mutable std::thread calib_thread_;
    mutable std::mutex calib_mu_;
    mutable std::condition_variable calib_cv_;
    mutable int calib_attempts_ = 0;
    unsigned long long* density_host_ = nullptr;  // {candidates, bytes}, pinned
    unsigned long long* density_dev_ = nullptr;   // the same, as the device sees it
};

class Workspace {
  public:
    Workspace(const Tokenizer& tok, uint64_t max_sentences, uint64_t max_bytes);
    ~Workspace();
    // defer_pack: stop behind the token-offset scan (the total is in d_ctrl[kTotal]); pack_to() then writes tok_off, tok_cnt and
    // the packed records wherever the caller wants them (vbt_tokenize_batch: straight into its pinned host block)
    void run(const uint8_t* d_text, const uint64_t* d_offsets, uint64_t n, uint64_t total_bytes, void* stream, bool defer_pack = false);
    void pack_to(vbt_token_rec* out_tokens, uint32_t* out_off, uint32_t* out_cnt, void* stream);
    // The results of every later run() go straight into ONE caller-owned device buffer, laid out as a rank's slot of the final gather
    // (vibrato_amd/sharding.py): [32-byte header {n_sentences u64, n_tokens u32, 0}] [tok_off u32 x max_sentences] [tok_cnt u32 x
    // max_sentences] [24-byte token records] -- tok_cnt by the sweep, tok_off and the records by compact_tokens, the header with its
    // total: no copy kernels between the tokenizer and the collective.  slot == nullptr: back to the workspace's own buffers.
    void set_packed_output(void* slot, uint64_t slot_bytes, uint64_t max_sentences);
    void* packed_slot = nullptr;
    uint64_t packed_bytes = 0, packed_max_s = 0;
    void stats(vbt_call_stats* out);  // synchronizes the last stream used
    // Worker::tokenize() latency path: starts the resident kernel that serves one Worker out of its pinned host block (`h_text_dev`:
    // device address of the block's text area, padded to 16 bytes; `ctl`: device address of its control words, see tokenize_serve in
    // engine.hip; `tokens_out`: device address of its token records).  `d_text` (>= capacity + 16 bytes) / `d_offsets` (2 words) are
    // device scratch of the caller.  `last_seq`: the sequence number already served; `idle_polls`: polls without work after which the
    // kernel leaves (0: one sentence, then out).  Asynchronous on `stream`, which the kernel occupies until it leaves.
    void serve(const uint8_t* h_text_dev, uint8_t* d_text, uint64_t* d_offsets, vbt_token_rec* tokens_out, uint32_t* ctl, uint32_t last_seq,
               uint32_t idle_polls, void* stream);

    const Tokenizer& tok;
    uint64_t max_sentences, max_bytes;
    vbt_token_rec* d_tokens = nullptr;
    vbt_token_rec* d_tok_stage = nullptr;
    uint32_t* d_tile_sums = nullptr;
    uint32_t *d_tok_off = nullptr, *d_tok_cnt = nullptr, *d_ctrl = nullptr, *d_over = nullptr, *d_cctrl = nullptr;
    std::vector<void*> pipe_allocs;  // buffers of the two-kernel pipeline
    bool lean_tier_default = false;  // the default tier set with the lean tier in front (run() may shrink that tier for a batch of short sentences)
    uint32_t last_T = 0;             // LDS tiers of the last run()
    bool last_inline = false;        // the last run used gen_sweep (stats: its sentences are counted in the first tier's cursor word)
    BatchArgs pipe{};                // device pointers of those buffers
    std::vector<void*> streams;      // one side stream per LDS tier
    std::vector<void*> tier_events;
    void* ev_fork2 = nullptr;
    void* ev_fork_early = nullptr;  // behind build_lists: where the lean tiers' sweeps fork off (created at first use)
    bool fused = false;              // VBT_FUSED=1: the single fused kernel per sentence (A/B reference)
    unsigned long long* d_connid = nullptr;  // [num_left + num_right] usage counters (DEVICE ids of image `count_epoch`), allocated on first use
    uint32_t* d_counted = nullptr;           // per-sentence watermark of the counted steps
    bool count_connids = false;
    uint32_t count_epoch = 0;                // the image whose numbering d_connid is in
    std::vector<uint64_t> acc_lid, acc_rid;  // counts of earlier images, already in dictionary ids (the image changed while counting)
    void enable_connid_counts(bool on);
    void read_connid_counts(uint64_t* lid, uint64_t* rid, bool reset);  // dictionary ids
    void reset_connid_counts();
    void fold_connid_counts();               // device counters -> acc_* (dictionary ids), device counters zeroed
    unsigned long long* d_prof = nullptr;
    char* d_scratch = nullptr;
    uint64_t scratch_bytes = 0;
    std::vector<uint32_t> tiers;  // LDS bytes per wave of each LDS tier
    uint32_t last_seg_tier = 0xFFFFFFFFu;  // of the last run(): what stats() counts as "routed up front" (tiers <= it) and as escape tiers
    std::vector<uint32_t> tier_waves;  // lattice grid per tier (empty = fill the machine per tier)
    bool timing = false, profile = false;
    void read_profile(uint64_t* out, bool reset);  // kProfWords values
    uint64_t last_n = 0;
    BatchArgs last_args{};  // of the last run(): what pack_to() packs
    void* last_stream = nullptr;
    bool has_run = false;  // (last_stream == nullptr is the null stream, not "never ran")
    void* ev[4] = {nullptr, nullptr, nullptr, nullptr};

  private:
    void release();  // frees every device allocation, stream and event (also on a constructor failure)
};

}  // namespace vbt
