// HIP kernels + launch sequence of the MI355X tokenizer (gfx950, wave64).  DESIGN.md section 3 is the map.
//
// Pipeline (default): validate_batch (device-side input contract) -> gen_candidates (one wavefront per sentence: UTF-8 decode, char categories and groupable
// runs -- Sentence::compile, sentence.rs:34-71 -- and candidate generation by double-array common-prefix search +
// unknown-word rules -- Tokenizer::add_lattice_edges tokenizer.rs:141-199, UnkHandler::gen_unk_words
// unknown.rs:69-137) -> build_lists -> gen_candidates_large (gen_long: one WORKGROUP per sentence that outgrew the bulk generator's LDS) ->
// lattice_lds (by default ONE 10 KiB tier that sweeps longer sentences in segments; one wavefront per sentence: the position sweep with per-node min-cost
// search over the connection matrix -- build_lattice_inner tokenizer.rs:94-139, Lattice::insert_node /
// search_min_node lattice.rs:103-151, insert_eos 85-101 -- and the back-trace, append_top_nodes
// lattice.rs:159-168; the lattice lives in LDS, longer sentences are swept in segments between clean cuts; what the
// generator found too dense for that is swept by a 48 KiB launch next to the tiers) -> tokenize_global for whatever is
// left -> compact_tokens (tokens from per-sentence staging into sentence order; tok_tile_scan in front of it for huge batches).  The fused single-kernel design (process_sentence: tokenize_lds /
// tokenize_global) is the fallback with a global-memory lattice and, with VBT_FUSED=1, an A/B reference.  Worker::tokenize is
// tokenize_one: generator + sweep of one sentence in ONE launch, text and token records through pinned host memory.
//
// Bit-exactness notes: end lists are built with LDS atomics, so their order is arbitrary;
// every node carries its insertion sequence number and ties are broken towards the LARGEST sequence number,
// which is exactly what `<=` does in search_min_node (lattice.rs:141-146).  Candidates of positions the sweep
// never visits (unreachable or inside a skipped space run) stay "dead" and are ignored as predecessors.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "engine.hpp"

#ifndef VBT_GEN_OCC
#define VBT_GEN_OCC 8
#endif
#if VBT_GEN_OCC
#define VBT_GEN_OCC_ATTR __attribute__((amdgpu_waves_per_eu(VBT_GEN_OCC, VBT_GEN_OCC)))
#else
#define VBT_GEN_OCC_ATTR
#endif

namespace vbt {
namespace {

constexpr uint64_t kNoFit = ~0ull;
constexpr uint8_t kRouteDone = 0xFC;  // s_tier value of a sentence that already sits in a work list

// Cache policy of the three random-access streams (A/B knobs): non-temporal loads
// do not allocate in the per-CU vector L1, whose in-order tag pipeline stalls on hit-under-miss.
#ifndef VBT_NT_MATRIX
#define VBT_NT_MATRIX 0
#endif
#ifndef VBT_NT_TRIE
#define VBT_NT_TRIE 0
#endif
template <bool kNt, typename T>
__device__ __forceinline__ T load_policy(const T* p) {
    if constexpr (kNt) return __builtin_nontemporal_load(p);
    else return *p;
}

#define HIP_CHECK(expr)                                                                               \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            throw Error(VBT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

// ---------------------------------------------------------------- wave helpers

// Inclusive prefix sum over the 64 lanes of a wavefront in registers: four row shifts inside the 16-lane rows, then the last lane of
// a row broadcast into the next one (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- the gfx9 DPP scan; lanes
// without a source add 0.  (Was six __shfl_up = six trips through the LDS crossbar.)  Every lane of the wave must be active.
__device__ __forceinline__ uint32_t wave_inscan_dpp(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31
    return x;
}
__device__ __forceinline__ uint32_t wave_exscan(uint32_t v, uint32_t& total) {
    const uint32_t x = wave_inscan_dpp(v);
    total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);  // an SGPR: what depends on it stays wave-uniform for the compiler
    return x - v;
}
__device__ __forceinline__ uint32_t wave_exscan_any(uint32_t v, uint32_t& total) { return wave_exscan(v, total); }  // (workgroups of several waves)
// the same ladder with max (unsigned: identity 0): inclusive prefix maximum over the lanes
__device__ __forceinline__ uint32_t wave_inscan_max_dpp(uint32_t x) {
#define VBT_MAX_DPP(ctrl, rows) { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rows, 0xF, false); x = o_ > x ? o_ : x; }
    VBT_MAX_DPP(0x111, 0xF) VBT_MAX_DPP(0x112, 0xF) VBT_MAX_DPP(0x114, 0xF) VBT_MAX_DPP(0x118, 0xF) VBT_MAX_DPP(0x142, 0xA) VBT_MAX_DPP(0x143, 0xC)
#undef VBT_MAX_DPP
    return x;
}
// sum / maximum over the wave, wave-uniform (lane 63 of the inclusive scans)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_inscan_dpp(v), 63); }
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_inscan_max_dpp(v), 63); }

// Packed lattice key: high word = min_cost biased to unsigned order (cost ^ 0x80000000), low word =
// 0xFFFFFFFE - insertion sequence number.  Unsigned-minimum over keys = minimum cost with ties broken
// towards the LAST inserted node, the `<=` rule of search_min_node (lattice.rs:141-146).  Adding a
// connection cost is a wrapping add on the high word.  Low word 0xFFFFFFFF marks "never inserted".
constexpr uint64_t kDeadKey = ~0ull;
__device__ __forceinline__ uint64_t make_key(uint32_t cost, uint32_t seq) {
    return ((uint64_t)(cost ^ 0x80000000u) << 32) | (0xFFFFFFFEu - seq);
}
__device__ __forceinline__ uint32_t key_cost(uint64_t k) { return (uint32_t)(k >> 32) ^ 0x80000000u; }
__device__ __forceinline__ uint32_t key_seq(uint64_t k) { return 0xFFFFFFFEu - (uint32_t)k; }

// Minimum of a 32-bit value over aligned groups of 2^kLevels lanes, left in every lane of the group.  Each level is one
// v_min_u32 with a DPP source operand (quad_perm / row_half_mirror / row_mirror: the mirrors are fine because the
// sub-blocks are already uniform); the 32- and 64-lane levels use the gfx950 row / half-wave swaps
// (v_permlane16_swap / v_permlane32_swap) instead of the LDS crossbar.
template <int kCtrl>
__device__ __forceinline__ uint32_t dpp_min_u32(uint32_t x) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, kCtrl, 0xF, 0xF, true);
    return o < x ? o : x;
}
template <int kLevels>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t x) {
    if constexpr (kLevels >= 1) x = dpp_min_u32<0xB1>(x);
    if constexpr (kLevels >= 2) x = dpp_min_u32<0x4E>(x);
    if constexpr (kLevels >= 3) x = dpp_min_u32<0x141>(x);
    if constexpr (kLevels >= 4) x = dpp_min_u32<0x140>(x);
    if constexpr (kLevels >= 5) { const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false); x = r[0] < r[1] ? r[0] : r[1]; }
    if constexpr (kLevels >= 6) { const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false); x = r[0] < r[1] ? r[0] : r[1]; }
    return x;
}
// 128-bit window helpers (shift distances 0..64), by value so everything stays in registers
struct U128 { uint64_t lo, hi; };
__device__ __forceinline__ U128 shr128(U128 w, uint32_t d) {
    const uint64_t lo_s = d >= 64 ? w.hi : (d ? (w.lo >> d) | (w.hi << (64 - d)) : w.lo);
    const uint64_t hi_s = d >= 64 ? 0ull : (w.hi >> d);
    return U128{lo_s, hi_s};
}
__device__ __forceinline__ U128 or_shl128(U128 w, uint64_t m, uint32_t d) {
    const uint64_t lo_m = d >= 64 ? 0ull : (m << d);
    const uint64_t hi_m = d >= 64 ? m : (d ? m >> (64 - d) : 0ull);
    return U128{w.lo | lo_m, w.hi | hi_m};
}
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    // (the builtin returns int: without the casts the low half would sign-extend into the high half)
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
}

// Per-sentence regions of the workspace (per-character records, candidates, staged hits) are addressed by the
// sentence's byte offset RELATIVE to the batch (offsets[0] may be anything: a window into a larger text buffer)
// plus kSentenceSlack slots per sentence before it: sentence s owns the character slots [off(s) + K s, off(s + 1) + K (s + 1)) -- its
// bytes + 1 are what the per-character arrays need, the rest is head room for its node region (node_factor slots per character slot):
// a short sentence of a dense lexicon has more than node_factor nodes per byte (0.3 % of the dense law's sentences took the
// global-memory fallback for that, 0.6 of its 5.3 ms per step).
__device__ __forceinline__ size_t sentence_slot(const BatchArgs& A, uint64_t b0, uint32_t sid) {
    return (size_t)(b0 - uniform64(A.offsets[0])) + (size_t)kSentenceSlack * sid;
}
__device__ __forceinline__ bool batch_rejected(const BatchArgs& A) {
    return (__builtin_amdgcn_readfirstlane(A.ctrl[kError]) & (uint32_t)kErrFatal) != 0;
}

// One step of the position sweep: candidates [cbeg, cbeg+nc) connect to end-list slots [pbeg, pbeg+np).
template <typename IdxT>
struct StepRec { IdxT cbeg, nc, pbeg, np; };

// Bump allocator over the per-sentence arena (LDS or a global slab).
struct Arena {
    char* base;
    uint64_t cap, used;
    bool ok;
    template <typename T>
    __device__ __forceinline__ T* take(uint64_t count) {
        uint64_t off = (used + alignof(T) - 1) & ~(uint64_t)(alignof(T) - 1);
        used = off + count * sizeof(T);
        if (used > cap) { ok = false; return reinterpret_cast<T*>(base); }
        return reinterpret_cast<T*>(base + off);
    }
};

// Counter loads in the global tier must not be served from a stale L1 line after L2
// atomics (the vector L1 is not updated by atomics executed in L2).
template <bool kGlobal>
__device__ __forceinline__ uint32_t load_counter(const uint32_t* p) {
    if constexpr (kGlobal) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

// ------------------------------------------------- candidate generation pieces

// Common-prefix search over the 16-byte-node double array (one load per transition).
// Reference: Lexicon::common_prefix_iterator lexicon.rs:33-46 / crawdad CPS (trie.rs:49-57).
template <typename F>
__device__ __forceinline__ bool walk_trie(const DevLexicon& L, const uint16_t* code, uint32_t i, uint32_t n, F&& on_hit) {
    bool matched = false;
    uint32_t cur = 0, base = L.root_base;
    for (uint32_t j = i; j < n; ++j) {
        const uint32_t c = code[j];
        if (c == 0) break;
        const uint32_t child = base ^ c;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 ndv = load_policy<VBT_NT_TRIE != 0>(reinterpret_cast<const u32x4*>(&L.nodes[child]));
        const uint4 nd = make_uint4(ndv.x, ndv.y, ndv.z, ndv.w);
        if (nd.y != cur) break;
        cur = child;
        base = nd.x;
        if (nd.w) { matched = true; on_hit(nd.z, nd.w, j + 1); }
    }
    return matched;
}

// UnkHandler::gen_unk_words unknown.rs:69-116: calls emit(end_char) once per generated span.
template <typename F>
__device__ __forceinline__ void unk_spans(uint32_t cinfo, uint32_t g, uint32_t i, bool matched, uint32_t maxg, F&& emit) {
    const bool invoke = (cinfo >> 26) & 1u, group = (cinfo >> 27) & 1u;
    const uint32_t length = cinfo >> 28;
    if (matched && !invoke) return;
    bool grouped = false;
    if (group) {
        grouped = true;
        if (g - 1 <= maxg) { emit(i + g); matched = true; }
    }
    const uint32_t lim = length < g ? length : g;
    for (uint32_t k = 1; k <= lim; ++k) {
        if (grouped && k == g) continue;
        emit(i + k);
        matched = true;
    }
    if (!matched) emit(i + 1);
}

// ------------------------------------------------------------ the fused kernel body

// Tokenizes sentence `sid` inside the arena [abase, abase+acap). Returns 0 when done,
// otherwise the number of arena bytes it would need (kNoFit: can never fit this tier).
// kWide: the connection matrix holds i32 cells (a compact connector whose costs leave i16, raw_connector.rs:153-161).
template <typename IdxT, bool kGlobal, bool kWide>
__device__ __forceinline__ uint64_t process_sentence(const DevDict& D, const BatchArgs& A, uint32_t sid, char* abase,
                                                     uint64_t acap) {
    typedef typename std::conditional<kWide, int32_t, int16_t>::type ConnT;
    const uint32_t ln = threadIdx.x;
    const uint64_t lt_mask = (1ull << ln) - 1ull;
    // optional per-phase cycle accounting (A.prof != nullptr): s_memtime deltas summed per launch
    uint64_t prof_t = A.prof ? clock64() : 0, prof_acc[kProfPhases] = {};
#define PROF_MARK(i)                                  \
    do {                                              \
        if (A.prof) {                                 \
            const uint64_t t_ = clock64();            \
            prof_acc[i] += t_ - prof_t;               \
            prof_t = t_;                              \
        }                                             \
    } while (0)
    constexpr uint64_t kIdxMax = (uint64_t)(IdxT) ~(IdxT)0;
    const uint64_t b0 = A.offsets[sid], nb64 = A.offsets[sid + 1] - b0;
    if (nb64 == 0) {
        if (ln == 0) A.tok_cnt[sid] = 0;
        return 0;
    }
    if (nb64 >= kIdxMax) return kNoFit;
    const uint32_t nb = (uint32_t)nb64;
    const uint8_t* __restrict__ txt = A.text + b0;

    // ---- P0a: count characters (UTF-8 lead bytes) ---------------------------------
    uint32_t n = 0;
    for (uint32_t c0 = 0; c0 < nb; c0 += 64) {
        const uint32_t bi = c0 + ln;
        const bool lead = bi < nb && (txt[bi] & 0xC0) != 0x80;
        n += (uint32_t)__popcll(__ballot(lead));
    }
    if (n == 0) {
        if (ln == 0) A.tok_cnt[sid] = 0;
        return 0;
    }

    Arena ar{abase, acap, 0, true};
    uint32_t* ci = ar.take<uint32_t>(n);            // CharInfo per char
    uint32_t* end_off = ar.take<uint32_t>(n + 2);   // end-list offsets (u32: LDS atomics)
    uint16_t* code = ar.take<uint16_t>(n);          // system-trie code per char
    uint16_t* ucode = D.has_user ? ar.take<uint16_t>(n) : code;
    IdxT* c2b = ar.take<IdxT>(n + 1);               // char -> byte offset
    IdxT* grp = ar.take<IdxT>(n);                   // groupable run length
    IdxT* cand_off = ar.take<IdxT>(n + 1);          // candidates by start position (CSR)
    uint8_t* reach = ar.take<uint8_t>(n + 1);       // has_previous_node
    if (!ar.ok) return ar.used + (uint64_t)n * 128;  // lower bound; exact size follows after counting

    // ---- P0b: decode, CharInfo, trie codes (Sentence::compute_basic/categories) ------
    {
        uint32_t cb = 0;
        for (uint32_t c0 = 0; c0 < nb; c0 += 64) {
            const uint32_t bi = c0 + ln;
            const uint32_t b = bi < nb ? txt[bi] : 0x80u;
            const bool lead = (b & 0xC0) != 0x80;
            const uint64_t m = __ballot(lead);
            if (lead) {
                const uint32_t idx = cb + (uint32_t)__popcll(m & lt_mask);
                const uint32_t t1 = bi + 1 < nb ? txt[bi + 1] & 0x3Fu : 0u;
                const uint32_t t2 = bi + 2 < nb ? txt[bi + 2] & 0x3Fu : 0u;
                const uint32_t t3 = bi + 3 < nb ? txt[bi + 3] & 0x3Fu : 0u;
                uint32_t cp;
                if (b < 0x80) cp = b;
                else if (b < 0xE0) cp = ((b & 0x1F) << 6) | t1;
                else if (b < 0xF0) cp = ((b & 0x0F) << 12) | (t1 << 6) | t2;
                else cp = ((b & 0x07) << 18) | (t1 << 12) | (t2 << 6) | t3;
                ci[idx] = D.chr2inf[cp < 65536u ? cp : 0u];  // character.rs:112-116
                code[idx] = cp < D.sys.mapper_len ? D.sys.mapper[cp] : (uint16_t)0;
                if (D.has_user) ucode[idx] = cp < D.user.mapper_len ? D.user.mapper[cp] : (uint16_t)0;
                c2b[idx] = (IdxT)bi;
            }
            cb += (uint32_t)__popcll(m);
        }
        if (ln == 0) c2b[n] = (IdxT)nb;
    }
    __syncthreads();

    // ---- groupable (Sentence::compute_groupable sentence.rs:57-71) -------------------
    {
        uint32_t carry = 0;
        for (int ch = (int)((n - 1) / 64); ch >= 0; --ch) {
            const uint32_t i = (uint32_t)ch * 64 + ln;
            const bool valid = i < n;
            bool link = false;
            if (valid && i + 1 < n) link = ((ci[i] & ci[i + 1]) & 0x3FFFFu) != 0;
            const uint64_t brk = __ballot(valid && !link);
            const uint64_t m = brk >> ln;
            const uint32_t g = m ? (uint32_t)__builtin_ctzll(m) + 1 : (64 - ln) + carry;
            if (valid) grp[i] = (IdxT)g;
            carry = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
        }
    }
    __syncthreads();
    PROF_MARK(0);

    // ---- P1a: count candidates per start position --------------------------------------
    uint32_t C = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        const uint32_t i = c0 + ln;
        uint32_t cnt = 0;
        if (i < n) {
            bool matched = false;
            if (D.has_user) matched |= walk_trie(D.user, ucode, i, n, [&](uint32_t, uint32_t c, uint32_t) { cnt += c; });
            matched |= walk_trie(D.sys, code, i, n, [&](uint32_t, uint32_t c, uint32_t) { cnt += c; });
            const uint32_t cinfo = ci[i], cate = (cinfo >> 18) & 0xFFu;
            const uint32_t nunk = D.unk_off[cate + 1] - D.unk_off[cate];
            unk_spans(cinfo, grp[i], i, matched, D.max_grouping_len, [&](uint32_t) { cnt += nunk; });
        }
        uint32_t tot;
        const uint32_t ex = wave_exscan(cnt, tot);
        if (i < n) cand_off[i] = (IdxT)(C + ex);
        C += tot;
    }
    if ((uint64_t)C + 3 >= kIdxMax) return kNoFit;
    if (ln == 0) cand_off[n] = (IdxT)C;
    PROF_MARK(1);

    // start-major node arrays; index C is the EOS pseudo node (left_id 0), C+1 stands for BOS
    uint64_t* e_key = ar.take<uint64_t>(C + 2);  // end-major: packed (cost, sequence) key, see make_key
    uint64_t* lens = ar.take<uint64_t>(n + 1);   // per start position: bit L-1 set <=> a candidate of length L
    StepRec<IdxT>* st;                           // sweep steps (one per visited start position, + EOS)
    if constexpr (sizeof(StepRec<IdxT>) == 8) st = reinterpret_cast<StepRec<IdxT>*>(lens);  // step S <= its start_word: safe alias
    else st = ar.take<StepRec<IdxT>>(n + 1);
    uint32_t* nd_word = ar.take<uint32_t>(C);
    uint16_t* nd_left = ar.take<uint16_t>(C + 1);
    int16_t* nd_wcost = ar.take<int16_t>(C + 1);
    uint16_t* e_right = ar.take<uint16_t>(C + 2);
    IdxT* nd_end = ar.take<IdxT>(C);
    IdxT* nd_eslot = ar.take<IdxT>(C + 2);
    IdxT* e_back = ar.take<IdxT>(C + 2);  // end-major: sequence number of the best predecessor
    if (!ar.ok) return ar.used;
    uint16_t* tmp_right = reinterpret_cast<uint16_t*>(e_back);  // right ids until the end lists exist

    // zero the end counters while the fill pass runs
    for (uint32_t p = ln; p < n + 2; p += 64) end_off[p] = 0;
    for (uint32_t p = ln; p < n + 1; p += 64) reach[p] = 0;
    __syncthreads();

    // ---- P1b: fill candidates in reference insertion order (tokenizer.rs:155-198) --------
    bool any_long = false;  // a word longer than 64 chars: the windowed pre-pass cannot represent it
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        const uint32_t i = c0 + ln;
        bool is_long = false;
        if (i < n) {
            uint32_t k = cand_off[i];
            uint64_t lmask = 0;
            bool matched = false;
            auto put = [&](const Entry* ent, uint32_t v, uint32_t c, uint32_t end, uint32_t lex) {
                const uint32_t len = end - i;
                if (len <= 64) lmask |= 1ull << (len - 1); else is_long = true;
                for (uint32_t t = 0; t < c; ++t, ++k) {
                    const Entry e = ent[v + t];
                    nd_word[k] = (lex << 30) | e.word_id;
                    nd_left[k] = (uint16_t)(e.left_right & 0xFFFFu);
                    tmp_right[k] = (uint16_t)(e.left_right >> 16);
                    nd_wcost[k] = (int16_t)(uint16_t)e.cost;
                    nd_end[k] = (IdxT)end;
                    // end-list slot within its end position (order irrelevant, see header)
                    nd_eslot[k] = (IdxT)atomicAdd(&end_off[end], 1u);
                }
            };
            if (D.has_user)
                matched |= walk_trie(D.user, ucode, i, n, [&](uint32_t v, uint32_t c, uint32_t e) { put(D.user.entries, v, c, e, 1u); });
            matched |= walk_trie(D.sys, code, i, n, [&](uint32_t v, uint32_t c, uint32_t e) { put(D.sys.entries, v, c, e, 0u); });
            const uint32_t cinfo = ci[i], cate = (cinfo >> 18) & 0xFFu;
            const uint32_t u0 = D.unk_off[cate], nunk = D.unk_off[cate + 1] - u0;
            unk_spans(cinfo, grp[i], i, matched, D.max_grouping_len, [&](uint32_t e) { put(D.unk_entries, u0, nunk, e, 2u); });
            lens[i] = lmask;
        }
        any_long |= __ballot(is_long) != 0;
    }
    __syncthreads();
    PROF_MARK(2);

    // ---- P2: end lists: exclusive scan of per-end counts; slot 0 is BOS --------------------
    {
        uint32_t running = 0;
        for (uint32_t c0 = 0; c0 < n + 1; c0 += 64) {
            const uint32_t p = c0 + ln;
            uint32_t cnt = 0;
            if (p < n + 1) cnt = load_counter<kGlobal>(&end_off[p]) + (p == 0 ? 1u : 0u);  // BOS in ends[0], lattice.rs:72-83
            uint32_t tot;
            const uint32_t ex = wave_exscan(cnt, tot);
            if (p < n + 1) end_off[p] = running + ex;
            running += tot;
        }
        if (ln == 0) end_off[n + 1] = running;
    }
    __syncthreads();
    for (uint32_t c = ln; c < C; c += 64) {
        const uint32_t es = end_off[nd_end[c]] + nd_eslot[c];
        const uint16_t r = tmp_right[c];
        nd_eslot[c] = (IdxT)es;
        e_right[es] = r;
        e_key[es] = kDeadKey;  // never inserted until a sweep step reaches its start position
    }
    __syncthreads();  // all tmp_right reads done before e_back is written
    const uint32_t kBosSeq = C + 1;
    if (ln == 0) {
        e_right[0] = 0;  // BOS: right_id = BOS_EOS_CONNECTION_ID, min_cost = 0 (lattice.rs:72-83)
        e_key[0] = make_key(0u, kBosSeq);
        nd_eslot[kBosSeq] = 0;
        e_back[0] = (IdxT)kBosSeq;
        nd_left[C] = 0;  // EOS: left_id = BOS_EOS_CONNECTION_ID (lattice.rs:85-101), no word cost
        nd_wcost[C] = 0;
        nd_eslot[C] = (IdxT)(C + 1);
        e_key[C + 1] = kDeadKey;
    }
    __syncthreads();
    PROF_MARK(3);

    // ---- P3a: structural pre-pass of build_lattice_inner (tokenizer.rs:106-138): which
    // (start_node, start_word) steps the sweep takes depends only on which positions have a
    // word ending there, never on costs.  Records one step per visited start position + EOS.
    uint32_t S = 0, sn_eos = 0;
    uint64_t total_pairs = 0, max_pairs = 0;
    bool windowed = !any_long;
    if (windowed) {
        // Reachability as a 128-bit sliding window: bit b <=> a word ends at position p + b.
        U128 w{1, 0};  // BOS ends at position 0
        uint32_t p = 0;
        while (p < n) {
            w.lo = uniform64(w.lo);  // wave-uniform by construction: keep the state machine on the scalar unit
            w.hi = uniform64(w.hi);
            p = __builtin_amdgcn_readfirstlane(p);
            if (!(w.lo & 1)) {  // has_previous_node(p) is false: skip to the next reachable position
                uint32_t z = w.lo ? (uint32_t)__builtin_ctzll(w.lo) : 64u;
                if (z > n - p) z = n - p;
                w = shr128(w, z);
                p += z;
                continue;
            }
            uint32_t sw = p;
            if (D.space_cateset) {  // tokenizer.rs:117-125
                const uint32_t cs = __builtin_amdgcn_readfirstlane(ci[p]);
                if (cs & D.space_cateset) sw += __builtin_amdgcn_readfirstlane((uint32_t)grp[p]);
            }
            if (sw >= n) break;  // input ends with spaces, tokenizer.rs:128-130
            const uint32_t d = sw - p + 1;
            if (d > 64) { windowed = false; break; }  // a space run too long for the window: generic path
            const uint64_t lm = uniform64(lens[sw]);
            const uint32_t c_beg = __builtin_amdgcn_readfirstlane((uint32_t)cand_off[sw]);
            const uint32_t c_end = __builtin_amdgcn_readfirstlane((uint32_t)cand_off[sw + 1]);
            const uint32_t p_beg = __builtin_amdgcn_readfirstlane(end_off[p]);
            const uint32_t p_end = __builtin_amdgcn_readfirstlane(end_off[p + 1]);
            if (ln == 0) st[S] = StepRec<IdxT>{(IdxT)c_beg, (IdxT)(c_end - c_beg), (IdxT)p_beg, (IdxT)(p_end - p_beg)};
            const uint64_t pairs = (uint64_t)(c_end - c_beg) * (p_end - p_beg);
            total_pairs += pairs;
            max_pairs = pairs > max_pairs ? pairs : max_pairs;
            ++S;
            // words starting at sw end at sw + L: bit (L - 1) of lm -> window bit (L - 1) + d; then advance to sw + 1
            w = shr128(or_shl128(w, lm, d), d);
            p = sw + 1;
        }
        sn_eos = p < n ? p : n;
        if (!windowed) { S = 0; total_pairs = 0; max_pairs = 0; }
    }
    if (!windowed) {  // generic path: byte-per-position reachability in LDS
        __syncthreads();
        if (ln == 0) reach[0] = 1;
        __syncthreads();
        uint32_t sn = 0, sw = 0;
        while (sw < n) {
            if (!__builtin_amdgcn_readfirstlane(reach[sn])) {  // has_previous_node, lattice.rs:155-157
                sw += 1;
                sn = sw;
                continue;
            }
            if (D.space_cateset) {
                const uint32_t cs = __builtin_amdgcn_readfirstlane(ci[sn]);
                if (cs & D.space_cateset) sw += __builtin_amdgcn_readfirstlane((uint32_t)grp[sn]);
            }
            if (sw == n) break;
            const uint32_t c_beg = __builtin_amdgcn_readfirstlane((uint32_t)cand_off[sw]);
            const uint32_t c_end = __builtin_amdgcn_readfirstlane((uint32_t)cand_off[sw + 1]);
            const uint32_t p_beg = __builtin_amdgcn_readfirstlane(end_off[sn]);
            const uint32_t p_end = __builtin_amdgcn_readfirstlane(end_off[sn + 1]);
            for (uint32_t c = c_beg + ln; c < c_end; c += 64) reach[nd_end[c]] = 1;
            __syncthreads();  // (also orders the lens[] reads of other lanes before st[] overwrites them)
            if (ln == 0) st[S] = StepRec<IdxT>{(IdxT)c_beg, (IdxT)(c_end - c_beg), (IdxT)p_beg, (IdxT)(p_end - p_beg)};
            const uint64_t pairs = (uint64_t)(c_end - c_beg) * (p_end - p_beg);
            total_pairs += pairs;
            max_pairs = pairs > max_pairs ? pairs : max_pairs;
            ++S;
            sw += 1;
            sn = sw;
        }
        sn_eos = sn;
    }
    {   // EOS step (insert_eos(start_node), tokenizer.rs:138): one candidate (node C), preds = ends[sn]
        const uint32_t p_beg = __builtin_amdgcn_readfirstlane(end_off[sn_eos]);
        const uint32_t p_end = __builtin_amdgcn_readfirstlane(end_off[sn_eos + 1]);
        if (ln == 0) st[S] = StepRec<IdxT>{(IdxT)C, (IdxT)1, (IdxT)p_beg, (IdxT)(p_end - p_beg)};
        total_pairs += p_end - p_beg;
        max_pairs = (uint64_t)(p_end - p_beg) > max_pairs ? (uint64_t)(p_end - p_beg) : max_pairs;
        ++S;
    }
    // connection-cost staging buffer: all pairs if they fit, else as many whole steps as fit
    uint64_t q_cap;
    ConnT* conn;
    {
        const uint64_t off = (ar.used + sizeof(ConnT) - 1) & ~(uint64_t)(sizeof(ConnT) - 1);
        const uint64_t room = ar.cap > off ? (ar.cap - off) / sizeof(ConnT) : 0;
        if (room < max_pairs) return off + sizeof(ConnT) * max_pairs;
        q_cap = room < total_pairs ? room : total_pairs;
        conn = reinterpret_cast<ConnT*>(ar.base + off);
    }
    __syncthreads();
    PROF_MARK(4);

    // ---- P3b/P4: per block of steps: gather the connection costs of every (candidate,
    // predecessor) pair into `conn` with many loads in flight (addresses depend on ids only),
    // then run the cost recurrence of search_min_node/insert_node (lattice.rs:103-151) from LDS.
    const ConnT* __restrict__ matrix = reinterpret_cast<const ConnT*>(D.matrix);
    const uint32_t NR = D.num_right;
    for (uint32_t k = 0; k < S;) {
        uint32_t kend = k;
        {
            uint64_t q = 0;
            while (kend < S) {
                const StepRec<IdxT> r = st[kend];
                const uint64_t pairs = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)r.nc) *
                                       __builtin_amdgcn_readfirstlane((uint32_t)r.np);
                if (q + pairs > q_cap) break;
                q += pairs;
                ++kend;
            }
        }
        // gather: layout conn[soff + j * nc + ci] (pred-major: consecutive lanes = consecutive candidates).
        // A slot is 64 consecutive pairs of one step; the (j, ci) of a lane advances incrementally by
        // (64 / nc, 64 % nc) from slot to slot, so there is one division per step, none per pair.
        {
            constexpr int U = 32;
            uint32_t kk = k, q0 = 0, soff = 0;
            uint32_t c_beg = 0, nc = 1, p_beg = 0, pairs = 0, dq = 0, dr = 0;
            uint32_t pj = 0, pr = 0;  // this lane's pred / candidate offset within the current slot
            auto load_step = [&]() {
                const StepRec<IdxT> r = st[kk];
                c_beg = __builtin_amdgcn_readfirstlane((uint32_t)r.cbeg);
                nc = __builtin_amdgcn_readfirstlane((uint32_t)r.nc);
                p_beg = __builtin_amdgcn_readfirstlane((uint32_t)r.pbeg);
                pairs = nc * __builtin_amdgcn_readfirstlane((uint32_t)r.np);
                dq = 64u / nc;
                dr = 64u - dq * nc;
                pj = ln / nc;
                pr = ln - pj * nc;
            };
            load_step();
            while (kk < kend) {
                // Issue U independent gathers before the first use: loads are unconditional (inactive
                // slots re-read a valid cell) and kept in 32-bit registers, so the compiler places one
                // counted s_waitcnt per consumer instead of one vmcnt(0) per load.
                int32_t val[U];
                uint32_t idx[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    kk = __builtin_amdgcn_readfirstlane(kk);
                    q0 = __builtin_amdgcn_readfirstlane(q0);
                    soff = __builtin_amdgcn_readfirstlane(soff);
                    const bool live = kk < kend;
                    const uint32_t ql = q0 + ln;
                    const bool valid = live && ql < pairs;
                    const uint32_t left = nd_left[c_beg + (valid ? pr : 0u)];
                    const uint32_t right = e_right[p_beg + (valid ? pj : 0u)];
                    val[u] = load_policy<VBT_NT_MATRIX != 0>(&matrix[(size_t)left * NR + right]);  // matrix_connector.rs:79-85
                    idx[u] = valid ? soff + ql : 0xFFFFFFFFu;
                    if (live) {
                        q0 += 64;
                        if (q0 >= pairs) {
                            soff += pairs;
                            q0 = 0;
                            ++kk;
                            if (kk < kend) load_step();
                        } else {
                            pj += dq;
                            pr += dr;
                            if (pr >= nc) { pr -= nc; ++pj; }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (idx[u] != 0xFFFFFFFFu) conn[idx[u]] = (ConnT)val[u];
            }
        }
        __syncthreads();
        PROF_MARK(5);
        // cost recurrence, one step per visited start position
        {
            uint32_t soff = 0;
            for (uint32_t kk = k; kk < kend; ++kk) {
                const StepRec<IdxT> sr = st[kk];
                const uint32_t c_beg = __builtin_amdgcn_readfirstlane((uint32_t)sr.cbeg);
                const uint32_t nc = __builtin_amdgcn_readfirstlane((uint32_t)sr.nc);
                const uint32_t p_beg = __builtin_amdgcn_readfirstlane((uint32_t)sr.pbeg);
                const uint32_t np = __builtin_amdgcn_readfirstlane((uint32_t)sr.np);
                for (uint32_t cb = 0; cb < nc; cb += 64) {
                    const uint32_t ci_ = cb + ln;
                    if (ci_ < nc) {
                        const uint32_t c = c_beg + ci_;
                        const uint32_t es = nd_eslot[c];
                        const uint32_t wcost = (uint32_t)(int32_t)nd_wcost[c];
                        // argmin over packed keys: minimum key = minimum cost, ties -> largest insertion
                        // sequence number, i.e. the `<=` of search_min_node (lattice.rs:141-146)
                        uint64_t best = kDeadKey;
                        const ConnT* col = conn + soff + ci_;
                        const uint64_t* pk = e_key + p_beg;
                        for (uint32_t j = 0; j < np; j += 8) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const uint32_t jj = j + u < np ? j + u : np - 1;  // tail re-reads the last pred (idempotent)
                                const uint64_t kb = pk[jj];
                                const uint32_t cv = (uint32_t)(int32_t)col[(size_t)jj * nc];
                                uint64_t key = kb + ((uint64_t)cv << 32);  // wrapping i32 add of the connection cost
                                key = (uint32_t)kb == 0xFFFFFFFFu ? kDeadKey : key;
                                best = key < best ? key : best;
                            }
                        }
                        const uint32_t bseq = key_seq(best);
                        e_key[es] = make_key(key_cost(best) + wcost, c);  // lattice.rs:125
                        e_back[es] = (IdxT)bseq;
                    }
                }
                soff += nc * np;
                __syncthreads();
            }
        }
        PROF_MARK(6);
        k = kend;
    }

    if (A.lid_count) {
        // Lattice::add_connid_counts (lattice.rs:170-183), see lattice_lds; steps of positions below the sentence's
        // watermark were counted by the LDS pipeline before it passed the sentence on (candidates are in start order)
        const uint32_t counted = __builtin_amdgcn_readfirstlane(A.s_counted[sid]);
        const uint32_t c_skip = counted >= n ? C : (uint32_t)cand_off[counted];
        for (uint32_t k = 0; k < S && counted <= n; ++k) {
            const StepRec<IdxT> r = st[k];
            const bool eos_step = k + 1 == S;
            uint32_t c_beg = __builtin_amdgcn_readfirstlane((uint32_t)r.cbeg), nc = __builtin_amdgcn_readfirstlane((uint32_t)r.nc);
            uint32_t p_beg = __builtin_amdgcn_readfirstlane((uint32_t)r.pbeg), p_end = p_beg + __builtin_amdgcn_readfirstlane((uint32_t)r.np);
            if (!eos_step && c_beg < c_skip) continue;
            if (eos_step) { p_beg = __builtin_amdgcn_readfirstlane(end_off[n]); p_end = __builtin_amdgcn_readfirstlane(end_off[n + 1]); }  // EOS pairs with ends[len_char]
            uint32_t live = 0;
            for (uint32_t j0 = p_beg; j0 < p_end; j0 += 64) {
                const uint32_t j = j0 + ln;
                const bool alive = j < p_end && (uint32_t)e_key[j] != 0xFFFFFFFFu;
                live += (uint32_t)__popcll(__ballot(alive));
                if (alive) atomicAdd(&A.rid_count[e_right[j]], (unsigned long long)nc);
            }
            for (uint32_t c = c_beg + ln; c < c_beg + nc; c += 64) atomicAdd(&A.lid_count[nd_left[c]], (unsigned long long)live);
        }
        if (ln == 0) A.s_counted[sid] = n + 1;
        __syncthreads();
    }

    // ---- P5: back-trace (append_top_nodes lattice.rs:159-168) + token records ------------------
    IdxT* path = grp;  // groupable is dead after the sweep; tokens <= chars
    uint32_t T = 0;
    if (ln == 0) {
        uint32_t seq = e_back[C + 1];
        while (seq != kBosSeq && T < n) {  // tokens <= chars; the bound also keeps a corrupted chain finite
            path[T++] = (IdxT)seq;
            seq = e_back[nd_eslot[seq]];
        }
    }
    T = (uint32_t)__builtin_amdgcn_readfirstlane((int)T);
    // tokens go to the sentence's own region of the staging buffer (tokens <= characters <= bytes: it cannot overflow);
    // compact_tokens packs them in sentence order afterwards -- no allocation atomic on a hot counter
    const size_t out_base = sentence_slot(A, b0, sid);
    __syncthreads();
    if (ln == 0) { A.tok_cnt[sid] = T; if (T) atomicAdd(&A.tile_sums[sid / kScanTile], T); }
    for (uint32_t t = ln; t < T; t += 64) {
        const uint32_t c = path[T - 1 - t];  // Worker::token: index = n-1-i (worker.rs:65-68)
        // start_word = the position whose candidate range contains c: upper_bound(cand_off, c) - 1
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint32_t)cand_off[mid + 1] <= c) lo = mid + 1; else hi = mid;
        }
        const uint32_t stp = lo, en = nd_end[c];
        vbt_token_rec r;
        r.start_char = stp; r.end_char = en;
        r.start_byte = c2b[stp]; r.end_byte = c2b[en];
        r.word_idx = nd_word[c];
        r.total_cost = (int32_t)key_cost(e_key[nd_eslot[c]]);
        A.tok_stage[out_base + t] = r;
    }
    PROF_MARK(7);
    if (A.prof && ln == 0) {
        unsigned long long* pr_ = A.prof + (size_t)(sid & (kProfSlots - 1)) * kProfWords;
#pragma unroll
        for (int i = 0; i < kProfPhases; ++i) atomicAdd(&pr_[i], (unsigned long long)prof_acc[i]);
        atomicAdd(&pr_[kProfPhases], 1ull);
    }
#undef PROF_MARK
    return 0;
}

// Geometry of the sweep kernel (lattice_lds).  A sweep step -- all candidates of one start word x all nodes ending at its start node --
// is cut into PASSES of up to kRoundCands candidates x up to kRoundPreds predecessors.  In a pass lane = (candidate cl = lane >> 2,
// phase k = lane & 3): the lane walks the predecessors j = 4 i + k, i = 0 .. 3 ("units": one instruction stream per unit covers
// 4 predecessors x 16 candidates), keeps the minimum of its own phase in registers, and the four phases of a candidate are
// combined with two quad-permute levels at the end of the step.  VBT_DEPTH = passes whose connection costs are in flight.
#ifndef VBT_DEPTH
#define VBT_DEPTH 3
#endif
#ifndef VBT_LAT_WAVES
#define VBT_LAT_WAVES 4
#endif
#ifndef VBT_ROUND_PREDS
#define VBT_ROUND_PREDS 16
#endif
#ifndef VBT_DUMMY_EXEC0
#define VBT_DUMMY_EXEC0 1
#endif
constexpr uint32_t kRoundPreds = VBT_ROUND_PREDS, kRoundCands = 16;  // (build knob: 8 or 16 predecessors = 2 or 4 units, i.e. gathers, per pass)
static_assert(kRoundPreds == 8 || kRoundPreds == 16, "a pass walks 2 or 4 units of 4 predecessors");
constexpr uint32_t kUnits = kRoundPreds / 4;
// 64-byte pass records, in the sentence's own region of GLOBAL memory (the dead upper half of its hit-staging region), read back by
// the sweep loop with ONE scalar load per pass: everything that steers an iteration arrives in SGPRs, lane masks included, without
// a VALU or SALU instruction spent on it.  The loop's software pipeline is baked into the data: iteration i issues the gathers of
// pass i + VBT_DEPTH and consumes pass i, so record r holds the ISSUE half of pass r and the CONSUME half of pass r - VBT_DEPTH -- no
// register rings for what an iteration needs of an older record.  Built once per pass by the lane that owns the step (two records
// touched per pass).
//   issue half:   w0 / w1 = LDS address of the slot record of the pass's first predecessor / of its first candidate's record;
//                 m[i] = EXEC of unit i's gather: the lanes (4 per candidate) of the candidates that exist while a later unit follows,
//                 the lanes that hold a pair as the last unit, 0 behind it
//   consume half: w0c = the predecessor address of pass r - VBT_DEPTH;  w1c = its candidate address | its units (1..4; 0: an empty
//                 pass) << 20 | first round of its candidates << 23 | last round << 24;  lm = the lanes of its LAST unit that hold a
//                 pair (every unit before the last is full);  vm = the lanes that hold a pair in ANY unit of the step (phase <
//                 predecessors, candidate exists): what the combine at the end of the step looks at
// (what only the connection-id counting needs of a pass -- predecessors | first pass of the step << 15 | candidates << 16 of the
// whole step -- sits in a u32 array behind the records)
struct alignas(64) LPass { uint32_t w0, w1, w0c, w1c; uint64_t m[4]; uint64_t lm, vm; };
static_assert(sizeof(LPass) == 64, "one s_load_dwordx16 per pass");
__host__ __device__ __forceinline__ uint32_t step_passes(uint32_t nc, uint32_t np) {
    return ((np + kRoundPreds - 1) / kRoundPreds) * ((nc + kRoundCands - 1) / kRoundCands);
}
// LDS bytes of the lattice arrays of lattice_lds for a (segment of a) sentence of n positions with C candidates and a window of
// E end-list slots: 8 bytes per slot, 8 per candidate, 2 per position (the token path).  Must over-estimate the Arena carve
// there; gen_candidates routes sentences to LDS tiers with it.
__host__ __device__ __forceinline__ uint64_t lattice_fixed_bytes(uint32_t C, uint32_t n, uint32_t E) {
    return 8ull * (E + 2ull) + 8ull * (C + 2ull) + 2ull * (n + 4ull) + 48;
}
// Cost word of a slot whose node was never inserted (its start position is never visited).  Biased cost 0xC0000000 = +2^30: with
// 16-bit connection and word costs a sentence of < 8000 characters keeps every live cost inside +-2^29, so such a predecessor
// loses every minimum without being tested for; lattice_sentence tests the slot's own field instead where that bound does not
// hold (i32 matrix cells, longer sentences).
constexpr uint32_t kDeadHi = 0xC0000000u;

// =====================================================================================
// Two-kernel pipeline (default).  Candidate generation is memory-latency bound (dependent
// double-array loads), the lattice sweep is LDS bound: splitting them lets the first run at
// high occupancy with a small LDS footprint and lets the second be launched per exact LDS tier,
// all tiers concurrently on side streams.
// =====================================================================================

extern __shared__ __attribute__((aligned(16))) char g_smem[];

// First kernel of every batch: the device-side input contract.  Offsets must not decrease and must span at most
// `total_bytes` (what the caller declared, <= the workspace capacity); the text must be valid UTF-8 (Rust `str`
// validity: the reference takes `&str`, sentence.rs:28-32) with every sentence starting on a character boundary.
// A violation sets kErrOffsets / kErrUtf8 and the batch is skipped: no later kernel touches a per-sentence region.
__global__ void __launch_bounds__(256) validate_batch(BatchArgs A, uint64_t total_bytes) {
    const uint64_t o0 = A.offsets[0], oN = A.offsets[A.n];
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (uint64_t)gridDim.x * 256;
    uint32_t bad = 0;
    if (oN < o0 || oN - o0 > total_bytes) bad |= kErrOffsets;
    for (uint64_t s = tid; s < A.n; s += nthreads) {
        const uint64_t a = A.offsets[s], b = A.offsets[s + 1];
        if (A.s_tier) A.s_tier[A.sid0 + s] = 0xFF;  // nothing routed yet
        if ((s & (kScanTile - 1)) == 0) A.tile_sums[s / kScanTile] = 0;  // token totals per packing tile: added up by the kernels that emit
        if (b < a || a < o0 || b > oN) bad |= kErrOffsets;
        else if (a < oN && (A.text[a] & 0xC0) == 0x80) bad |= kErrUtf8;  // a sentence starts inside a character
    }
    if (!(bad & kErrOffsets) && oN - o0 <= total_bytes) {
        // Eight bytes per thread and round, read as the aligned 8-byte word they sit in plus the word behind it (4 bytes of
        // look-ahead): two loads instead of twelve.  Bytes of those words outside the text count as 0 (an aligned word that holds
        // a byte of the text lies in the text's page).
        const uint8_t* __restrict__ t = A.text + o0;
        const uint64_t nb = oN - o0;
        const uint64_t head = reinterpret_cast<uintptr_t>(t) & 7u;  // bytes of the first word in front of the text
        const uint64_t* __restrict__ tw = reinterpret_cast<const uint64_t*>(t - head);
        const uint64_t nwords = (head + nb + 7) >> 3;
        for (uint64_t w = tid; w < nwords; w += nthreads) {
            const uint64_t w0 = tw[w], w1 = w + 1 < nwords ? tw[w + 1] : 0ull;
            const int64_t i0 = (int64_t)(w << 3) - (int64_t)head;  // text index of the word's first byte (negative inside the head)
            uint32_t b[12];  // 8 lead positions + 4 bytes of look-ahead; outside the text = 0 (not a continuation byte)
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const uint32_t v = (uint32_t)((k < 8 ? w0 >> (8 * k) : w1 >> (8 * (k - 8))) & 0xFFu);
                b[k] = (i0 + k >= 0 && (uint64_t)(i0 + k) < nb) ? v : 0u;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t c = b[k];
                if (i0 + k < 0 || (uint64_t)(i0 + k) >= nb || (c & 0xC0) == 0x80) continue;  // continuation bytes are checked from their lead byte
                const uint32_t len = c < 0x80 ? 1u : c < 0xE0 ? 2u : c < 0xF0 ? 3u : 4u;
                bool ok = c < 0x80 || (c >= 0xC2 && c < 0xF5);
#pragma unroll
                for (uint32_t q = 1; q <= 4; ++q) {
                    const bool cont = (b[k + q] & 0xC0) == 0x80;
                    if (q < len) ok &= cont;
                    if (q == len) ok &= !cont;  // a stray continuation byte behind a complete character
                }
                const uint32_t b1 = b[k + 1];
                if (c == 0xE0) ok &= b1 >= 0xA0;  // overlong 3-byte form
                if (c == 0xED) ok &= b1 < 0xA0;   // surrogates
                if (c == 0xF0) ok &= b1 >= 0x90;  // overlong 4-byte form
                if (c == 0xF4) ok &= b1 < 0x90;   // above U+10FFFF
                if (!ok) bad |= kErrUtf8;
            }
        }
    }
    if (__ballot(bad != 0)) {  // rare: one atomic per offending lane
        if (bad) atomicOr(&A.ctrl[kError], bad);
    }
}

__device__ __forceinline__ void list_push(const BatchArgs& A, uint32_t t, uint32_t sid) {
    if (threadIdx.x == 0) A.lists[(size_t)t * A.list_stride + A.list_off + atomicAdd(&A.cctrl[2 * t], 1u)] = sid;
}

__device__ __forceinline__ void list_push_fb(const BatchArgs& A, uint32_t sid) { list_push(A, A.n_tiers, sid); }  // the fallback list (fused kernel)

// LDS bytes gen_long needs for a sentence of n characters / nb bytes (an over-estimate of its Arena carve: gen_one files a
// sentence that outgrows it at the smallest level that holds it).
__host__ __device__ __forceinline__ uint64_t gen_long_bytes(uint32_t n, uint32_t nb, bool has_user) {
    return 64 + 2 * ((uint64_t)(nb >> 6) + 4) + (uint64_t)(n + 2) * (has_user ? 16u : 14u) + 64;  // ci 4, code 2 (+ user 2), grp 2, co 2, endc 4 per character
}

// characters of a sentence (lead bytes), counted by one wavefront
__device__ __forceinline__ uint32_t count_chars(const uint8_t* __restrict__ txt, uint32_t nb) {
    uint32_t n = 0;
    for (uint32_t c0 = 0; c0 < nb; c0 += 64) {
        const uint32_t bi = c0 + threadIdx.x;
        const bool lead = bi < nb && (txt[bi] & 0xC0) != 0x80;
        n += (uint32_t)__popcll(__ballot(lead));
    }
    return n;
}

// gen_one's per-character working arrays in its wavefront's LDS (~26 bytes per character).  `ok` = they fit: the test by which
// gen_one files what does not fit for gen_long.
struct GenOneLds {
    uint64_t* lens;
    uint32_t *ci, *cand_off;
    uint16_t *code, *ucode, *grp;
    uint32_t *endc, *hcount;
    bool ok;
};
__device__ __forceinline__ GenOneLds carve_gen_one(char* base, uint32_t lds_bytes, uint32_t n, bool has_user) {
    Arena ar{base, lds_bytes, 0, true};
    GenOneLds L;
    L.lens = ar.take<uint64_t>(n);
    L.ci = ar.take<uint32_t>(n);
    L.cand_off = ar.take<uint32_t>(n + 1);
    L.code = ar.take<uint16_t>(n);
    L.ucode = has_user ? ar.take<uint16_t>(n) : L.code;
    L.grp = ar.take<uint16_t>(n);
    L.endc = ar.take<uint32_t>(n + 1);
    L.hcount = ar.take<uint32_t>(1);
    L.ok = ar.ok;
    return L;
}
// the smallest level of gen_long whose LDS holds the sentence
__device__ __forceinline__ uint32_t gen_long_level(const BatchArgs& A, uint32_t n, uint32_t nb, bool has_user) {
    const uint64_t need = gen_long_bytes(n, nb, has_user);
    uint32_t lv = 0;
    while (lv + 1 < (uint32_t)kGenLevels && need > A.gen_level_bytes[lv]) ++lv;
    return lv;
}

// Kernel 1 body: Sentence::compile + candidate enumeration of one sentence by one wavefront.
// Per-character working arrays live in LDS (the vector L1 stalls on hit-under-miss, so nothing
// is re-read from global while in flight); outputs: per-char records, byte offsets and the
// candidates in reference insertion order, each tagged with its (start position, left_id) group:
// search_min_node's result depends only on that pair (lattice.rs:129-151), so the lattice kernel
// evaluates one row per group instead of one per candidate.
// (Sentences that outgrow this wavefront's LDS -- ~26 bytes per character -- are handed to gen_long: one workgroup per sentence.)
__device__ __forceinline__ void gen_one(const DevDict& D, const BatchArgs& A, uint32_t sid, uint32_t lds_bytes) {
    const uint32_t ln = threadIdx.x;
    const uint64_t lt_mask = (1ull << ln) - 1ull;
    uint64_t prof_t = A.prof ? clock64() : 0, prof_acc[3] = {};
#define PROF_MARK(i) do { if (A.prof) { const uint64_t t_ = clock64(); prof_acc[i] += t_ - prof_t; prof_t = t_; } } while (0)
    const uint64_t b0 = A.offsets[sid], nb64 = A.offsets[sid + 1] - b0;
    const uint32_t fallback = A.n_tiers;
    // gen routes a sentence by writing its list index; build_lists turns that into work lists with
    // wave-aggregated atomics (a per-sentence atomic on a hot word caps the kernel at ~88 M/s)
    auto route = [&](uint32_t t) {
        if (A.direct_push) list_push(A, t, sid);  // (Worker's single launch: no build_lists behind it)
        else if (ln == 0) A.s_tier[sid] = (uint8_t)t;
    };
    auto init = [&]() { if (ln == 0) { A.s_n[sid] = 0; A.s_C[sid] = 0; A.s_tier[sid] = 0xFF; } };
    if (nb64 == 0) {
        init();
        if (ln == 0) A.tok_cnt[sid] = 0;
        return;
    }
    if (nb64 >= 65535) { init(); route(fallback); return; }  // positions are u16 in the LDS lattice
    const uint32_t nb = (uint32_t)nb64;
    const uint8_t* __restrict__ txt = A.text + b0;
    const size_t slot0 = sentence_slot(A, b0, sid);

    const uint32_t n = count_chars(txt, nb);
    if (n == 0) {
        init();
        if (ln == 0) A.tok_cnt[sid] = 0;
        return;
    }
    const GenOneLds L = carve_gen_one(g_smem, lds_bytes, n, D.has_user != 0);
    if (!L.ok) {
        // Outgrows this wavefront: gen_long, one workgroup per sentence.
        init();
        route(A.n_tiers + 1 + gen_long_level(A, n, nb, D.has_user != 0));
        return;
    }
    init();
    uint64_t* const lens = L.lens;
    uint32_t* const ci = L.ci;
    uint32_t* const cand_off = L.cand_off;
    auto set_lens = [&](uint32_t i, uint64_t v) { lens[i] = v; };
    auto get_lens = [&](uint32_t i) -> uint64_t { return lens[i]; };
    auto set_co = [&](uint32_t i, uint32_t v) { cand_off[i] = v; };
    auto get_co = [&](uint32_t i) -> uint32_t { return cand_off[i]; };
    uint16_t* const code = L.code;
    uint16_t* const ucode = L.ucode;
    uint16_t* const grp = L.grp;
    uint32_t* const endc = L.endc;      // candidates ending at each position (bounds the pass count)
    uint32_t* const hcount = L.hcount;  // hits staged so far
    for (uint32_t i = ln; i < n + 1; i += 64) endc[i] = i == 0 ? 1u : 0u;  // BOS ends at 0

    // decode (Sentence::compute_basic / compute_categories, sentence.rs:40-55); the 3 bytes after a
    // lead byte come from neighbouring lanes (or the look-ahead chunk), not from memory again
    {
        uint16_t* c2b = A.g_c2b + slot0;
        uint32_t cb = 0;
        uint32_t cur = ln < nb ? txt[ln] : 0x80u;
        for (uint32_t c0 = 0; c0 < nb; c0 += 64) {
            const uint32_t bi = c0 + ln;
            const uint32_t nxt = bi + 64 < nb ? txt[bi + 64] : 0x80u;
            const uint32_t b = cur;
            uint32_t t[3];
#pragma unroll
            for (int k = 1; k <= 3; ++k) {
                const uint32_t src = (ln + k) & 63u;
                const uint32_t a = __shfl(cur, src), c = __shfl(nxt, src);
                t[k - 1] = ((ln + k < 64) ? a : c) & 0x3Fu;
            }
            const bool lead = bi < nb && (b & 0xC0) != 0x80;
            const uint64_t m = __ballot(lead);
            if (lead) {
                const uint32_t idx = cb + (uint32_t)__popcll(m & lt_mask);
                uint32_t cp;
                if (b < 0x80) cp = b;
                else if (b < 0xE0) cp = ((b & 0x1F) << 6) | t[0];
                else if (b < 0xF0) cp = ((b & 0x0F) << 12) | (t[0] << 6) | t[1];
                else cp = ((b & 0x07) << 18) | (t[0] << 12) | (t[1] << 6) | t[2];
                ci[idx] = D.chr2inf[cp < 65536u ? cp : 0u];  // character.rs:112-116
                code[idx] = cp < D.sys.mapper_len ? D.sys.mapper[cp] : (uint16_t)0;
                if (D.has_user) ucode[idx] = cp < D.user.mapper_len ? D.user.mapper[cp] : (uint16_t)0;
                c2b[idx] = (uint16_t)bi;
            }
            cb += (uint32_t)__popcll(m);
            cur = nxt;
        }
        if (ln == 0) c2b[n] = (uint16_t)nb;
    }
    __syncthreads();
    {   // groupable (sentence.rs:57-71)
        uint32_t carry = 0;
        for (int ch = (int)((n - 1) / 64); ch >= 0; --ch) {
            const uint32_t i = (uint32_t)ch * 64 + ln;
            const bool valid = i < n;
            bool link = false;
            if (valid && i + 1 < n) link = ((ci[i] & ci[i + 1]) & 0x3FFFFu) != 0;
            const uint64_t brk = __ballot(valid && !link);
            const uint64_t m = brk >> ln;
            const uint32_t g = m ? (uint32_t)__builtin_ctzll(m) + 1 : (64 - ln) + carry;
            if (valid) grp[i] = (uint16_t)g;
            carry = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
        }
    }
    __syncthreads();
    PROF_MARK(0);

    // One trie walk per start position (tokenizer.rs:155-198, unknown.rs:69-116).  The walk is a chain of
    // dependent loads, so nothing else hangs on it: every hit -- a run of `c` dictionary entries ending at
    // `end` -- is appended to a staging list in global memory as {first entry, c | lexicon << 16,
    // end | start << 16, candidates of this start position before the hit} and expanded afterwards by
    // independent lanes.
    const uint64_t base = (uint64_t)A.node_factor * slot0;  // this sentence's node region (no allocation atomic)
    const uint64_t region = (uint64_t)A.node_factor * (nb + kSentenceSlack);
    uint4* __restrict__ hits = A.g_hits + base;
    if (ln == 0) *hcount = 0;
    __syncthreads();
    uint32_t C = 0;
    bool any_long = false;
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        const uint32_t i = c0 + ln;
        uint32_t cnt = 0;
        uint64_t lmask = 0;
        bool is_long = false;
        if (i < n) {
            auto seen = [&](uint32_t v, uint32_t c, uint32_t end, uint32_t lex) {
                if (c == 0) return;  // a category without unknown-word entries contributes nothing (unknown.rs:118-130)
                const uint32_t h = atomicAdd(hcount, 1u);
                if (h < region) hits[h] = make_uint4(v, c | (lex << 16), end | (i << 16), cnt);
                cnt += c;
                const uint32_t len = end - i;
                if (len <= 64) lmask |= 1ull << (len - 1); else is_long = true;
                atomicAdd(&endc[end], c);
            };
            bool matched = false;
            if (D.has_user) matched |= walk_trie(D.user, ucode, i, n, [&](uint32_t v, uint32_t c, uint32_t e) { seen(v, c, e, 1u); });
            matched |= walk_trie(D.sys, code, i, n, [&](uint32_t v, uint32_t c, uint32_t e) { seen(v, c, e, 0u); });
            const uint32_t cinfo = ci[i], cate = (cinfo >> 18) & 0xFFu;
            const uint32_t u0 = D.unk_off[cate], nunk = D.unk_off[cate + 1] - u0;
            unk_spans(cinfo, grp[i], i, matched, D.max_grouping_len, [&](uint32_t e) { seen(u0, nunk, e, 2u); });
            set_lens(i, lmask);
        }
        uint32_t tot;
        const uint32_t ex = wave_exscan(cnt, tot);
        if (i < n) set_co(i, C + ex);
        C += tot;
        any_long |= __ballot(is_long) != 0;
    }
    // words > 64 chars need the generic pre-pass, > 65531 nodes need u32 indices: fused kernel
    if (C >= 65532 || any_long) { route(fallback); return; }
    if (ln == 0) set_co(n, C);
    // End lists (`ends[e]` of lattice.rs:39-43) are laid out here once and for all: node slots are numbered by end
    // position (BOS is slot 0, the only node ending at 0), so the lattice kernel reads every candidate with its slot
    // attached and builds no lists.  endc[] turns from counts into running cursors: exclusive prefix now, after the
    // expansion below the inclusive one (eo() recovers the exclusive offsets).  Order inside a list is arbitrary.
    __syncthreads();
    {
        uint32_t running = 0;
        for (uint32_t c0 = 0; c0 < n + 1; c0 += 64) {
            const uint32_t p = c0 + ln;
            const uint32_t cnt = p < n + 1 ? endc[p] : 0u;
            uint32_t tot;
            const uint32_t ex = wave_exscan(cnt, tot);
            if (p < n + 1) endc[p] = running + ex;
            running += tot;
        }
    }
    __syncthreads();
    if (C > region) { route(fallback); return; }  // denser than the region: fused path
    // The staged hits are read back by this wave only: its stores have to be complete (workgroup scope:
    // s_waitcnt vmcnt(0); the vector L1 is write-through and never held these lines).  An agent-scope
    // release would write the whole L2 back (buffer_wbl2) once per sentence.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    PROF_MARK(1);

    // expand the hits: lanes = hits, every entry load independent of every other; a candidate becomes ONE 16-byte record --
    // {first cell of its left id's matrix row, (u16) word_cost | end-list slot << 16, word_idx, end_char | right_id << 16} -- at its
    // place in the reference's insertion order (cand_off[start] + candidates of that start before the hit).  One scattered store
    // per candidate: the generator is bound by the number of its scattered store requests (two 8-byte stores into separate arrays
    // cost 5 % more; the sweep's load phase does not notice the wider record)
    const uint32_t H = *hcount;  // <= C <= region
    const uint32_t row_cells = D.num_right;  // a left id's row of the connection matrix starts at cell left_id * num_right
    for (uint32_t h0 = 0; h0 < H; h0 += 64) {
        const uint32_t h = h0 + ln;
        const uint4 hr = h < H ? hits[h] : make_uint4(0, 0, 0, 0);
        if (h < H) {
            const uint32_t c = hr.y & 0xFFFFu, lex = hr.y >> 16, end = hr.z & 0xFFFFu, pos = hr.z >> 16;
            const Entry* __restrict__ ent = lex == 0 ? D.sys.entries : lex == 1 ? D.user.entries : D.unk_entries;
            const uint32_t dest = get_co(pos) + hr.w;
            const uint32_t slot0 = atomicAdd(&endc[end], c);  // the hit's run of slots in ends[end]
            for (uint32_t t0 = 0; t0 < c; t0 += 4) {
                Entry e[4];
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) e[q] = ent[hr.x + (t0 + q < c ? t0 + q : t0)];
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
                    if (t0 + q < c) {
                        const uint32_t k = dest + t0 + q;
                        A.g_cand[base + k] = make_uint4((e[q].left_right & 0xFFFFu) * row_cells, (e[q].cost & 0xFFFFu) | ((slot0 + t0 + q) << 16),
                                                        (lex << 30) | e[q].word_id, end | (e[q].left_right & 0xFFFF0000u));
                    }
                }
            }
        }
    }
    __syncthreads();
    // exclusive end-list offset of position p (0 .. n + 1): the cursors now hold the inclusive prefix
    auto eo = [&](uint32_t p) { return p == 0 ? 0u : p == 1 ? 1u : endc[p - 1]; };
    uint32_t passes = 0, maxcnt = 1;
    {   // per-character records for the lattice kernel:
        // {cand_off | end-list offset << 16, pass bound of the position's step | window end << 14 | space << 31,
        //  length mask (64 bits; for a space position of ignore_space mode its groupable run instead: the sweep never
        //  starts a word there, tokenizer.rs:113-125)}
        // Window end of position i: the end-list offset behind the furthest end of any candidate of the positions <= i (with
        // ignore_space, a visited space run hands its visit to the position behind the run, so such a run counts as spanning up
        // to the furthest end of that position's candidates) -- the slots a sweep segment that ends behind i has to hold
        // (lattice_lds may cut the sweep of a long sentence anywhere).
        uint4* pc = A.g_pc + slot0;
        uint32_t far = 0;  // furthest end of any candidate of the positions before this chunk
        for (uint32_t c0 = 0; c0 < n; c0 += 64) {
            const uint32_t i = c0 + ln;
            uint32_t e = 0, space = 0, nsl = 0, cnt = 0, co_i = 0;
            uint64_t lm = 0;
            if (i < n) {
                const uint32_t cinfo = ci[i];
                space = (D.space_cateset && (cinfo & D.space_cateset)) ? 0x80000000u : 0u;
                lm = get_lens(i);
                e = lm ? i + 64u - (uint32_t)__builtin_clzll(lm) : i + 1;
                co_i = get_co(i);
                uint32_t nc = get_co(i + 1) - co_i;
                if (space) {
                    const uint32_t sw = i + grp[i];
                    const uint64_t lw = sw < n ? get_lens(sw) : 0ull;
                    const uint32_t e2 = sw < n ? (lw ? sw + 64u - (uint32_t)__builtin_clzll(lw) : sw + 1) : n;
                    e = e2 > e ? e2 : e;
                    nc = sw < n ? get_co(sw + 1) - get_co(sw) : 0u;  // the step taken from a space position starts its words behind the run
                }
                cnt = eo(i + 1) - eo(i);
                nsl = step_passes(nc, cnt);
            }
            uint32_t m = e;  // inclusive prefix maximum over the lanes
            m = wave_inscan_max_dpp(m);
            if (i < n) {
                const uint32_t upto = m > far ? m : far;  // furthest end of any candidate of the positions <= i (<= n)
                const uint64_t third = space ? (uint64_t)grp[i] : lm;
                const uint32_t yw = (nsl < 0x3FFFu ? nsl : 0x3FFFu) | (eo(upto + 1) << 14) | space;
                pc[i] = make_uint4(co_i | (eo(i) << 16), yw, (uint32_t)third, (uint32_t)(third >> 32));
            }
            const uint32_t top = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
            far = top > far ? top : far;
            uint32_t mc = cnt;
            nsl = wave_sum(nsl);
            mc = wave_umax(mc);
            passes += nsl;
            maxcnt = mc > maxcnt ? mc : maxcnt;
        }
        {   // EOS connects to the end list of the last visited position: bounded by the longest list
            const uint32_t last = eo(n + 1) - eo(n);
            maxcnt = last > maxcnt ? last : maxcnt;
            passes += step_passes(1u, maxcnt);
        }
        // terminator: totals (candidates, end-list slots)
        if (ln == 0) pc[n] = make_uint4(C | (eo(n) << 16), 0, eo(n + 1), 0);
    }
    if (ln == 0) {
        A.s_n[sid] = n; A.s_C[sid] = C; A.s_passes[sid] = passes;
    }
    // smallest tier whose LDS holds the lattice arrays (connection costs are never staged)
    const uint64_t fixed = lattice_fixed_bytes(C, n, eo(n + 1));
    uint32_t tier = fallback;
    for (uint32_t t = 0; t < A.n_tiers; ++t)
        if (fixed <= A.tier_bytes[t]) { tier = t; break; }
    // longer sentences are swept in segments inside the segment tier instead of one huge LDS block (lattice_lds cuts anywhere;
    // what it cannot sweep there -- a window of end lists wider than the tier -- it hands to the escape tiers itself)
    if (A.seg_tier < A.n_tiers && tier > A.seg_tier) tier = A.seg_tier;
    route(tier);
    PROF_MARK(2);
    if (A.prof && ln == 0) {
        unsigned long long* pr_ = A.prof + (size_t)(sid & (kProfSlots - 1)) * kProfWords;
        for (int i = 0; i < 3; ++i) atomicAdd(&pr_[i], (unsigned long long)prof_acc[i]);
        atomicAdd(&pr_[kProfPhases], 1ull);
    }
#undef PROF_MARK
}

// The generator for sentences that outgrew the bulk generator's LDS: ONE WORKGROUP (several wavefronts) per sentence.  Long
// sentences hold most of the characters of a mixed-length batch (BASELINE config 5: 5 % of the sentences, 55 % of the characters);
// with one wavefront each their LDS footprint (14-16 bytes per character) left 5-10 waves on a CU.  Same phases and the same
// outputs as gen_one (per-character records, candidates in the reference's insertion order with their end-list slots,
// routing); the 64-position chunks of every phase are dealt round-robin to the workgroup's waves, and what gen_one carries
// from chunk to chunk in registers becomes a small scan between two barriers:
//   characters before a byte chunk (decode)            -> lead bytes per chunk, exclusive prefix
//   candidates before a position (insertion order)     -> counts per position in LDS (u16), exclusive prefix
//   furthest end of any earlier candidate (clean cuts) -> maximum per chunk, exclusive prefix maximum
// The groupable runs (a right-to-left carry) and the prefixes are done by wave 0 in LDS: n / 64 short iterations.
__device__ __forceinline__ void gen_long(const DevDict& D, const BatchArgs& A, uint32_t sid, uint32_t lds_bytes, uint32_t level) {
    const uint32_t tid = threadIdx.x, ln = tid & 63u, wv = tid >> 6, nw = blockDim.x >> 6, nthreads = blockDim.x;
    const uint64_t lt_mask = (1ull << ln) - 1ull;
    const uint64_t b0 = uniform64(A.offsets[sid]), nb64 = uniform64(A.offsets[sid + 1]) - b0;
    // (gen_one filed the sentence at a level whose LDS holds it -- gen_long_bytes over-estimates the carve below -- so "does not
    // fit after all" only happens beyond the last level: the fused kernel takes it)
    const uint32_t fallback = A.n_tiers, next_level = fallback;
    (void)level;
    // (thread 0 appends: no build_lists behind these launches.  The fallback list is the batch's, everything else this launch's own.)
    auto route = [&](uint32_t t) { if (t == fallback) list_push_fb(A, sid); else list_push(A, t, sid); };
    if (tid == 0) { A.s_n[sid] = 0; A.s_C[sid] = 0; }
    if (nb64 == 0) {
        if (tid == 0) A.tok_cnt[sid] = 0;
        return;
    }
    if (nb64 >= 65535) { route(fallback); return; }  // positions are u16 in the LDS lattice
    const uint32_t nb = (uint32_t)nb64;
    const uint8_t* __restrict__ txt = A.text + b0;
    const size_t slot0 = sentence_slot(A, b0, sid);
    enum { kN = 0, kHits = 1, kLong = 2, kPasses = 3, kMaxCnt = 4, kC = 5 };  // red[]: block-wide scalars
    Arena ar{g_smem, lds_bytes, 0, true};
    uint32_t* red = ar.take<uint32_t>(8);
    const uint32_t nbc = (nb + 63) >> 6;
    uint16_t* chunk = ar.take<uint16_t>(nbc + 1);  // per byte chunk: characters before it; later per position chunk: furthest end before it
    if (!ar.ok) { route(next_level); return; }
    for (uint32_t ch = wv; ch < nbc; ch += nw) {
        const uint32_t bi = ch * 64 + ln;
        const bool lead = bi < nb && (txt[bi] & 0xC0) != 0x80;
        const uint32_t c = (uint32_t)__popcll(__ballot(lead));
        if (ln == 0) chunk[ch] = (uint16_t)c;
    }
    __syncthreads();
    if (wv == 0) {
        uint32_t running = 0;
        for (uint32_t c0 = 0; c0 < nbc; c0 += 64) {
            const uint32_t i = c0 + ln;
            const uint32_t v = i < nbc ? chunk[i] : 0u;
            uint32_t tot;
            const uint32_t ex = wave_exscan(v, tot);
            if (i < nbc) chunk[i] = (uint16_t)(running + ex);
            running += tot;
        }
        if (ln == 0) { red[kN] = running; red[kHits] = 0; red[kLong] = 0; red[kPasses] = 0; red[kMaxCnt] = 1; }
    }
    __syncthreads();
    const uint32_t n = __builtin_amdgcn_readfirstlane(red[kN]);
    if (n == 0) {
        if (tid == 0) A.tok_cnt[sid] = 0;
        return;
    }
    uint32_t* ci = ar.take<uint32_t>(n);
    uint16_t* code = ar.take<uint16_t>(n);
    uint16_t* ucode = D.has_user ? ar.take<uint16_t>(n) : code;
    uint16_t* grp = ar.take<uint16_t>(n);
    uint16_t* co = ar.take<uint16_t>(n + 1);    // candidates of a position, then candidates before it (insertion order: CSR offsets)
    uint32_t* endc = ar.take<uint32_t>(n + 1);  // candidates ending at each position: counts, then running cursors (see gen_one)
    if (!ar.ok) { route(next_level); return; }
    uint4* const pcw = A.g_pc + slot0;  // .z/.w = length mask until the records are finalised (as gen_one<kLarge>)
    for (uint32_t i = tid; i < n + 1; i += nthreads) endc[i] = i == 0 ? 1u : 0u;  // BOS ends at 0

    // decode (sentence.rs:40-55): every chunk loads its 64 bytes and the 64 behind them (the 3 bytes after a lead byte)
    {
        uint16_t* c2b = A.g_c2b + slot0;
        for (uint32_t ch = wv; ch < nbc; ch += nw) {
            const uint32_t bi = ch * 64 + ln;
            const uint32_t cur = bi < nb ? txt[bi] : 0x80u, nxt = bi + 64 < nb ? txt[bi + 64] : 0x80u;
            const uint32_t b = cur;
            uint32_t t[3];
#pragma unroll
            for (int k = 1; k <= 3; ++k) {
                const uint32_t src = (ln + k) & 63u;
                const uint32_t a = __shfl(cur, src), c = __shfl(nxt, src);
                t[k - 1] = ((ln + k < 64) ? a : c) & 0x3Fu;
            }
            const bool lead = bi < nb && (b & 0xC0) != 0x80;
            const uint64_t m = __ballot(lead);
            if (lead) {
                const uint32_t idx = chunk[ch] + (uint32_t)__popcll(m & lt_mask);
                uint32_t cp;
                if (b < 0x80) cp = b;
                else if (b < 0xE0) cp = ((b & 0x1F) << 6) | t[0];
                else if (b < 0xF0) cp = ((b & 0x0F) << 12) | (t[0] << 6) | t[1];
                else cp = ((b & 0x07) << 18) | (t[0] << 12) | (t[1] << 6) | t[2];
                ci[idx] = D.chr2inf[cp < 65536u ? cp : 0u];  // character.rs:112-116
                code[idx] = cp < D.sys.mapper_len ? D.sys.mapper[cp] : (uint16_t)0;
                if (D.has_user) ucode[idx] = cp < D.user.mapper_len ? D.user.mapper[cp] : (uint16_t)0;
                c2b[idx] = (uint16_t)bi;
            }
        }
        if (tid == 0) c2b[n] = (uint16_t)nb;
    }
    __syncthreads();
    if (wv == 0) {  // groupable (sentence.rs:57-71): right to left, the run length carried across chunks
        uint32_t carry = 0;
        for (int ch = (int)((n - 1) / 64); ch >= 0; --ch) {
            const uint32_t i = (uint32_t)ch * 64 + ln;
            const bool valid = i < n;
            bool link = false;
            if (valid && i + 1 < n) link = ((ci[i] & ci[i + 1]) & 0x3FFFFu) != 0;
            const uint64_t brk = __ballot(valid && !link);
            const uint64_t m = brk >> ln;
            const uint32_t g = m ? (uint32_t)__builtin_ctzll(m) + 1 : (64 - ln) + carry;
            if (valid) grp[i] = (uint16_t)g;
            carry = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
        }
    }
    __syncthreads();

    // one trie walk per start position (tokenizer.rs:155-198, unknown.rs:69-116): hits staged in global memory exactly as in gen_one
    const uint64_t base = (uint64_t)A.node_factor * slot0;
    const uint64_t region = (uint64_t)A.node_factor * (nb + kSentenceSlack);
    uint4* __restrict__ hits = A.g_hits + base;
    for (uint32_t c0 = wv * 64; c0 < n; c0 += nw * 64) {
        const uint32_t i = c0 + ln;
        bool is_long = false;
        if (i < n) {
            uint32_t cnt = 0;
            uint64_t lmask = 0;
            auto seen = [&](uint32_t v, uint32_t c, uint32_t end, uint32_t lex) {
                if (c == 0) return;  // a category without unknown-word entries contributes nothing (unknown.rs:118-130)
                const uint32_t h = atomicAdd(&red[kHits], 1u);
                if (h < region) hits[h] = make_uint4(v, c | (lex << 16), end | (i << 16), cnt);
                cnt += c;
                const uint32_t len = end - i;
                if (len <= 64) lmask |= 1ull << (len - 1); else is_long = true;
                atomicAdd(&endc[end], c);
            };
            bool matched = false;
            if (D.has_user) matched |= walk_trie(D.user, ucode, i, n, [&](uint32_t v, uint32_t c, uint32_t e) { seen(v, c, e, 1u); });
            matched |= walk_trie(D.sys, code, i, n, [&](uint32_t v, uint32_t c, uint32_t e) { seen(v, c, e, 0u); });
            const uint32_t cinfo = ci[i], cate = (cinfo >> 18) & 0xFFu;
            const uint32_t u0 = D.unk_off[cate], nunk = D.unk_off[cate + 1] - u0;
            unk_spans(cinfo, grp[i], i, matched, D.max_grouping_len, [&](uint32_t e) { seen(u0, nunk, e, 2u); });
            pcw[i].z = (uint32_t)lmask; pcw[i].w = (uint32_t)(lmask >> 32);
            co[i] = (uint16_t)(cnt < 0xFFFFu ? cnt : 0xFFFFu);
            if (cnt >= 0xFFFFu) is_long = true;  // (more candidates at one position than the u16 arrays hold: fused kernel)
        }
        if (__ballot(is_long) != 0 && ln == 0) atomicOr(&red[kLong], 1u);
    }
    __syncthreads();
    if (wv == 0) {  // candidates before a position (CSR offsets), then the end-list offsets
        uint32_t running = 0;
        for (uint32_t c0 = 0; c0 < n; c0 += 64) {
            const uint32_t i = c0 + ln;
            const uint32_t v = i < n ? co[i] : 0u;
            uint32_t tot;
            const uint32_t ex = wave_exscan(v, tot);
            if (i < n && running + ex < 0xFFFFu) co[i] = (uint16_t)(running + ex);
            running += tot;
            if (running >= 65532u) { running = 65532u; break; }  // (wave-uniform) too many nodes for u16 indices: fused kernel, see below
        }
        if (ln == 0) { red[kC] = running; if (running < 65532u) co[n] = (uint16_t)running; }
        uint32_t run2 = 0;
        for (uint32_t c0 = 0; c0 < n + 1; c0 += 64) {
            const uint32_t p = c0 + ln;
            const uint32_t cnt = p < n + 1 ? endc[p] : 0u;
            uint32_t tot;
            const uint32_t ex = wave_exscan(cnt, tot);
            if (p < n + 1) endc[p] = run2 + ex;
            run2 += tot;
        }
    }
    __syncthreads();
    const uint32_t C = __builtin_amdgcn_readfirstlane(red[kC]), H = __builtin_amdgcn_readfirstlane(red[kHits]);
    // words > 64 chars need the generic pre-pass, > 65531 nodes need u32 indices, denser than the region: fused kernel
    if (C >= 65532 || __builtin_amdgcn_readfirstlane(red[kLong]) || C > region) { route(fallback); return; }
    // the staged hits (and the length masks) are read back by other waves of this workgroup: stores complete, workgroup scope
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // expand the hits: threads = hits (see gen_one)
    // (the next round's hit records are requested before this round's entries: one round trip per round instead of two; in
    // gen_one, with three rounds per sentence, the same was measured slightly slower)
    uint4 hr_next = tid < H ? hits[tid] : make_uint4(0, 0, 0, 0);
    const uint32_t row_cells = D.num_right;  // (see gen_one)
    for (uint32_t h = tid; h < H; h += nthreads) {
        const uint4 hr = hr_next;
        if (h + nthreads < H) hr_next = hits[h + nthreads];
        const uint32_t c = hr.y & 0xFFFFu, lex = hr.y >> 16, end = hr.z & 0xFFFFu, pos = hr.z >> 16;
        const Entry* __restrict__ ent = lex == 0 ? D.sys.entries : lex == 1 ? D.user.entries : D.unk_entries;
        const uint32_t dest = (uint32_t)co[pos] + hr.w;
        const uint32_t es0 = atomicAdd(&endc[end], c);  // the hit's run of slots in ends[end]
        for (uint32_t t0 = 0; t0 < c; t0 += 4) {
            Entry e[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) e[q] = ent[hr.x + (t0 + q < c ? t0 + q : t0)];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                if (t0 + q < c) {
                    const uint32_t k = dest + t0 + q;
                    A.g_cand[base + k] = make_uint4((e[q].left_right & 0xFFFFu) * row_cells, (e[q].cost & 0xFFFFu) | ((es0 + t0 + q) << 16),
                                                    (lex << 30) | e[q].word_id, end | (e[q].left_right & 0xFFFF0000u));
                }
            }
        }
    }
    __syncthreads();
    auto eo = [&](uint32_t p) { return p == 0 ? 0u : p == 1 ? 1u : endc[p - 1]; };  // exclusive end-list offset: the cursors hold the inclusive prefix now
    auto get_lens = [&](uint32_t i) -> uint64_t { const uint4 r = pcw[i]; return ((uint64_t)r.w << 32) | r.z; };
    // per-character records (layout and meaning: gen_one).  Per position: e = the furthest end of its candidates (for a space
    // position of ignore_space mode: of the position behind the run); first the maximum per chunk, then its exclusive prefix
    // maximum (`far` of gen_one), then the records.
    struct PosInfo { uint32_t e, space, nsl, cnt, co_i; uint64_t lm; };
    auto pos_info = [&](uint32_t i) {
        PosInfo r{0, 0, 0, 0, 0, 0};
        if (i < n) {
            const uint32_t cinfo = ci[i];
            r.space = (D.space_cateset && (cinfo & D.space_cateset)) ? 0x80000000u : 0u;
            r.lm = get_lens(i);
            r.e = r.lm ? i + 64u - (uint32_t)__builtin_clzll(r.lm) : i + 1;
            r.co_i = co[i];
            uint32_t nc = (uint32_t)co[i + 1] - r.co_i;
            if (r.space) {
                const uint32_t sw = i + grp[i];
                const uint64_t lw = sw < n ? get_lens(sw) : 0ull;
                const uint32_t e2 = sw < n ? (lw ? sw + 64u - (uint32_t)__builtin_clzll(lw) : sw + 1) : n;
                r.e = e2 > r.e ? e2 : r.e;
                nc = sw < n ? (uint32_t)co[sw + 1] - co[sw] : 0u;  // the step taken from a space position starts its words behind the run
            }
            r.cnt = eo(i + 1) - eo(i);
            r.nsl = step_passes(nc, r.cnt);
        }
        return r;
    };
    auto wave_max = [&](uint32_t v) {
        v = wave_umax(v);
        return v;
    };
    // (the furthest ends are kept per position over the dead trie codes: the second pass must not read another position's length
    // mask again -- a wave may have finalised that record already, and a finalised space position keeps its run length there)
    uint16_t* const far_end = code;
    const uint32_t npc = (n + 63) >> 6;
    for (uint32_t ch = wv; ch < npc; ch += nw) {
        const uint32_t i = ch * 64 + ln;
        const uint32_t e = pos_info(i).e;
        if (i < n) far_end[i] = (uint16_t)e;  // (<= n < 65535)
        const uint32_t top = wave_max(e);
        if (ln == 0) chunk[ch] = (uint16_t)top;
    }
    __syncthreads();
    if (wv == 0) {
        uint32_t far = 0;
        for (uint32_t c0 = 0; c0 < npc; c0 += 64) {
            const uint32_t i = c0 + ln;
            uint32_t m = i < npc ? chunk[i] : 0u;
            m = wave_inscan_max_dpp(m);
            uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x138, 0xF, 0xF, false);  // wave_shr:1 (lane 0: 0)
            before = ln == 0 ? far : (before > far ? before : far);
            const uint32_t top = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
            if (i < npc) chunk[i] = (uint16_t)before;
            far = top > far ? top : far;
        }
    }
    __syncthreads();
    {
        uint4* pc = A.g_pc + slot0;
        for (uint32_t ch = wv; ch < npc; ch += nw) {
            const uint32_t i = ch * 64 + ln;
            PosInfo r = pos_info(i);
            r.e = i < n ? (uint32_t)far_end[i] : 0u;
            const uint32_t far = chunk[ch];
            uint32_t m = r.e;  // inclusive prefix maximum over the lanes
            m = wave_inscan_max_dpp(m);
            if (i < n) {
                const uint32_t upto = m > far ? m : far;  // furthest end of any candidate of the positions <= i
                const uint64_t third = r.space ? (uint64_t)grp[i] : r.lm;
                const uint32_t yw = (r.nsl < 0x3FFFu ? r.nsl : 0x3FFFu) | (eo(upto + 1) << 14) | r.space;
                pc[i] = make_uint4(r.co_i | (eo(i) << 16), yw, (uint32_t)third, (uint32_t)(third >> 32));
            }
            uint32_t nsl = r.nsl;
            nsl = wave_sum(nsl);
            const uint32_t mc = wave_max(r.cnt);
            if (ln == 0) { atomicAdd(&red[kPasses], nsl); atomicMax(&red[kMaxCnt], mc); }
        }
    }
    __syncthreads();
    uint32_t passes = __builtin_amdgcn_readfirstlane(red[kPasses]), maxcnt = __builtin_amdgcn_readfirstlane(red[kMaxCnt]);
    {   // EOS connects to the end list of the last visited position: bounded by the longest list
        const uint32_t last = eo(n + 1) - eo(n);
        maxcnt = last > maxcnt ? last : maxcnt;
        passes += step_passes(1u, maxcnt);
    }
    if (tid == 0) {
        A.g_pc[slot0 + n] = make_uint4(C | (eo(n) << 16), 0, eo(n + 1), 0);  // terminator: totals (candidates, end-list slots)
        A.s_n[sid] = n; A.s_C[sid] = C; A.s_passes[sid] = passes;
    }
    // smallest tier whose LDS holds the lattice arrays, else the segment tier (see gen_one)
    const uint64_t fixed = lattice_fixed_bytes(C, n, eo(n + 1));
    uint32_t tier = fallback;
    for (uint32_t t = 0; t < A.n_tiers; ++t)
        if (fixed <= A.tier_bytes[t]) { tier = t; break; }
    if (A.seg_tier < A.n_tiers && tier > A.seg_tier) tier = A.seg_tier;
    route(tier);
}

// Turns the per-sentence routing decisions into work lists: one atomic per (wave, list) instead of
// one per sentence.  only_list >= 0 restricts the pass to that list (the gen_candidates_large input).
__global__ void __launch_bounds__(1024) build_lists(BatchArgs A, int only_list) {
    // One global atomic per (workgroup, list): a returning atomic on a hot word costs ~11 ns, so the
    // 16 waves of a workgroup first agree on their shares through LDS.
    __shared__ uint32_t w_cnt[16][kMaxTiers + kListsBehindTiers];
    __shared__ uint32_t l_base[kMaxTiers + kListsBehindTiers];
    const uint32_t rel = blockIdx.x * 1024 + threadIdx.x, sid = A.sid0 + rel;
    const uint32_t t = rel < A.n ? A.s_tier[sid] : 0xFFu;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_lists = A.n_tiers + kListsBehindTiers;
    uint32_t my_rank = 0;
    for (uint32_t l = 0; l < n_lists; ++l) {
        const bool mine = t == l && (only_list < 0 || (int)l == only_list);
        const uint64_t m = __ballot(mine);
        if (lane == 0) w_cnt[wave][l] = (uint32_t)__popcll(m);
        if (mine) my_rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x < n_lists) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < 16; ++w) { const uint32_t c = w_cnt[w][threadIdx.x]; w_cnt[w][threadIdx.x] = tot; tot += c; }
        l_base[threadIdx.x] = tot ? atomicAdd(&A.cctrl[2 * threadIdx.x], tot) : 0u;
    }
    __syncthreads();
    const bool filed = t < n_lists && (only_list < 0 || (int)t == only_list);
    if (filed) {
        A.lists[(size_t)t * A.list_stride + A.list_off + l_base[t] + w_cnt[wave][t] + my_rank] = sid;
        if (only_list < 0) A.s_tier[sid] = kRouteDone;  // filed: a later pass must not file it again
    }
}

// Kernel 1: one single-wave workgroup per sentence (small LDS, high occupancy) ...
// (8 waves per SIMD: the kernel waits on memory three quarters of its time and its throughput follows its occupancy; left alone the
// compiler keeps 105 SGPRs -- the two argument structs -- and 112 allocated SGPRs per wave fit only 7 times into a SIMD's 800)
__global__ void __launch_bounds__(64) VBT_GEN_OCC_ATTR gen_candidates(DevDict D, BatchArgs A, uint32_t lds_bytes) {
    if (batch_rejected(A)) return;  // nothing gets routed: every later kernel finds empty work lists
    gen_one(D, A, A.sid0 + blockIdx.x, lds_bytes);
}
// ... and persistent workgroups (several wavefronts, a large LDS budget) for the sentences that did not fit: gen_long.
__global__ void __launch_bounds__(1024) VBT_GEN_OCC_ATTR gen_candidates_large(DevDict D, BatchArgs A, uint32_t lds_bytes, uint32_t level) {
    uint32_t* const next_item = reinterpret_cast<uint32_t*>(g_smem + lds_bytes - 16);  // (the last 16 bytes stay out of gen_long's arena)
    const uint32_t t = A.n_tiers + level;
    const uint32_t count = A.cctrl[2 * t];
    for (bool first = true;; first = false) {  // (first item = the workgroup's index, then the cursor: see tokenize_global)
        uint32_t k = blockIdx.x;
        if (!first) {
            if (threadIdx.x == 0) *next_item = gridDim.x + atomicAdd(&A.cctrl[2 * t + 1], 1u);
            __syncthreads();
            k = __builtin_amdgcn_readfirstlane(*next_item);
        }
        if (k >= count) break;
        gen_long(D, A, A.lists[(size_t)t * A.list_stride + A.list_off + k], lds_bytes - 16, level);
        __syncthreads();  // (also: next_item is read by every wave before thread 0 draws the next one)
    }
}

// Kernel 2: the lattice sweep of one sentence per wavefront, entirely in LDS (one list entry per workgroup; the escape
// tiers run persistent waves).  Sentences whose lattice does not fit after all go to the fallback list (fused kernel
// with global scratch).
//
// What lives in LDS per (segment of a) sentence: per end-list slot an 8-byte record {lo = (0xFFFE - sequence) << 16 | right id,
// hi = min_cost biased to unsigned order} -- the low word is static and written by the load phase, the cost by the step that
// inserts the node; per candidate 8 bytes {first cell of its matrix row, byte offset of its slot record | word cost << 16}
// (the low half of the first word becomes the node's back pointer once its step is done); the pass records (16 B each).
// Nothing per character: the per-character records of gen_candidates are consumed straight from global memory by the
// reachability sweep, 64 positions at a time, and the end lists were laid out by gen_candidates (every candidate arrives
// with its slot).
//
// The recurrence (lattice.rs:103-151) runs over PASSES of <= 16 candidates x <= 16 predecessors of one sweep step (LPass).
// Lane = (candidate cl = lane >> 2, phase k = lane & 3) walks the predecessors 4 i + k: it reads the predecessor's record
// (four addresses per instruction, each broadcast to 16 lanes), adds the connection cost of its pair -- gathered VBT_DEPTH passes
// ahead into a register ring -- and keeps the 64-bit minimum (cost, 0xFFFE - sequence of the predecessor): minimum cost, ties to
// the last inserted predecessor = the `<=` of lattice.rs:141-146.  At the end of the step two quad-permute levels combine the
// four phases, phase 0 adds the word cost and stores the node's cost into its slot record and the winner's field as its back
// pointer.  LDS operations of one wave execute in order, so no barrier separates a pass from the next.
//
// A sentence whose lattice does not fit the tier's LDS is swept in segments cut at ANY position b (a multiple of 8 positions
// behind the segment's start, not behind a space): slots are numbered by end position over the whole sentence, so the nodes
// of the finished segment that end behind the cut are final and sit in the slot range [eo(b), window end); that range is
// moved to the front of the slot window and the next segment carries on -- its load phase touches only the slots of its own
// candidates, the reachability state (three scalars) stays in registers.
template <bool kSpaceMode, bool kWide>
__device__ __forceinline__ uint32_t lattice_sentence(const DevDict& D, const BatchArgs& A, uint32_t tier, uint32_t sid) {
    typedef __attribute__((address_space(3))) const uint64_t lds_cu64;
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t ln = threadIdx.x;
    const uint32_t lds_bytes = A.tier_bytes[tier];
    constexpr uint32_t kD = VBT_DEPTH;  // passes whose gathers are in flight
    constexpr uint32_t kSh = kWide ? 2u : 1u;  // log2 of the matrix cell size
    // absolute LDS address of the dynamic shared memory (records hold absolute addresses: no base add per access)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)g_smem;
    auto uniform4 = [](uint4 q) {
        return make_uint4(__builtin_amdgcn_readfirstlane(q.x), __builtin_amdgcn_readfirstlane(q.y),
                          __builtin_amdgcn_readfirstlane(q.z), __builtin_amdgcn_readfirstlane(q.w));
    };
    {
        // profiling adds straight into the spread counters: nothing but the last time stamp lives between marks
        uint64_t prof_t = A.prof ? clock64() : 0;
        unsigned long long* const pr_ = A.prof ? A.prof + (size_t)(sid & (kProfSlots - 1)) * kProfWords : nullptr;
#define PROF_MARK(i) do { if (A.prof) { const uint64_t t_ = clock64(); if (ln == 0) atomicAdd(&pr_[i], (unsigned long long)(t_ - prof_t)); prof_t = t_; } } while (0)
        const uint32_t nT = __builtin_amdgcn_readfirstlane(A.s_n[sid]), CT = __builtin_amdgcn_readfirstlane(A.s_C[sid]);
        const uint32_t passesT = __builtin_amdgcn_readfirstlane(A.s_passes[sid]);
        const size_t slot0 = sentence_slot(A, uniform64(A.offsets[sid]), sid);
        const size_t node0 = (size_t)A.node_factor * slot0;
        const uint4* __restrict__ pcg = A.g_pc + slot0;   // per-character records (+ the terminator at nT)
        const uint4* __restrict__ ndg = A.g_cand + node0;  // candidate records in insertion order: {first cell of the matrix row, word cost | slot << 16, word_idx, end_char | right id << 16}
        const uint32_t ET = __builtin_amdgcn_readfirstlane(pcg[nT].z);  // end-list slots of the sentence (BOS included)
        const uint32_t kBosSeq = CT + 1;
        // The sentence's hit-staging region (dead after gen_candidates; 16 bytes per node slot): its lower half holds (total cost,
        // back pointer) of every node of a sentence that is swept in segments, its upper half the pass records of the current segment.
        const uint32_t nbT = (uint32_t)(uniform64(A.offsets[sid + 1]) - uniform64(A.offsets[sid]));
        const uint32_t half_bytes = 8u * A.node_factor * (nbT + kSentenceSlack);
        // (a sentence that is swept whole dumps nothing: its records take the whole region)
        const bool whole = lattice_fixed_bytes(CT, nT, ET) <= lds_bytes && passesT + 3 * kD + 4 <= 2 * half_bytes / (uint32_t)(sizeof(LPass) + 4);
        LPass* const rec = reinterpret_cast<LPass*>(reinterpret_cast<char*>(A.g_hits + node0) + (whole ? 0u : half_bytes));
        const uint32_t rec_cap = (whole ? 2 * half_bytes : half_bytes) / (uint32_t)(sizeof(LPass) + 4);
        uint32_t* const rec_w3 = reinterpret_cast<uint32_t*>(rec + rec_cap);  // per pass: step totals for the connection-id counting
        // where a dead predecessor's sentinel cost could meet a live cost, every predecessor's own field is tested instead (kDeadHi)
        const bool exact = kWide || nT >= 8000u;
        uint32_t seg_a = 0, seg_c = 0, seg_p = 0, sb = 0, m_in = 1, fail = 0;
        bool multi = false, done = false;
        uint32_t counted = A.lid_count ? __builtin_amdgcn_readfirstlane(A.s_counted[sid]) : 0u;
        uint32_t prof_S = 0, prof_SL = 0;
        uint32_t budget = lds_bytes;  // what a segment may be estimated at; shrinks when an estimate turns out too low
        // reachability state of the position sweep (tokenizer.rs:106-138), carried from segment to segment
        uint64_t sw_w = 0;
        uint32_t sw_cur = 1, sw_pend = 0;
        // the slot records sit at the start of the arena in every segment: the hand-over moves them in place
        uint2* const e_rec = reinterpret_cast<uint2*>(g_smem);
        const uint32_t offK = lds0;
        if (ln == 0) e_rec[0] = make_uint2(((0xFFFEu - kBosSeq) & 0xFFFFu) << 16, 0x80000000u);  // BOS: cost 0, right id 0 (lattice.rs:72-83)
        while (!done) {
        uint32_t seg_b = nT, seg_pass = passesT - seg_p, wend = ET;
        // (pass records of a segment live in global memory: rec_cap of them, the empty ones behind the last included)
        if (lattice_fixed_bytes(CT - seg_c, nT - seg_a, ET - sb) > budget || passesT - seg_p + 3 * kD + 4 > rec_cap) {
            // furthest admissible cut within 256 positions whose segment fits: any position a multiple of 8 behind the segment's
            // start (the bit-serial sweep below runs in groups of 8 positions) that does not follow a space (a visited space run and
            // the word it hands its visit to stay in one segment, tokenizer.rs:113-125)
            uint32_t best = 0, best_pass = 0, best_wend = 0, run = 0;
            for (uint32_t w0 = 0; w0 < 256 && seg_a + w0 < nT; w0 += 64) {
                const uint32_t b = seg_a + w0 + ln + 1;  // candidate segment end
                uint32_t nsl = 0, cx = 0, we = 0, sp = 0;
                if (b <= nT) {
                    const uint4 rp = pcg[b - 1], rb = pcg[b];
                    nsl = rp.y & 0x3FFFu;
                    if (nsl == 0x3FFFu) nsl = 1u << 20;  // saturated: unknown, treat as too many
                    cx = rb.x & 0xFFFFu;
                    we = (rp.y >> 14) & 0xFFFFu;  // end of the slot window of a segment that ends here
                    sp = rp.y >> 31;
                }
                uint32_t tot;
                const uint32_t incl = wave_exscan(nsl, tot) + nsl + run;
                const uint32_t est = b == nT ? passesT - seg_p : incl;  // (the sentence's bound includes the EOS step)
                const uint32_t wsl = b == nT ? ET : we;
                const bool fits = b <= nT && lattice_fixed_bytes((cx - seg_c) & 0xFFFFu, b - seg_a, wsl - sb) <= budget && est + 3 * kD + 4 <= rec_cap;
                const uint64_t m = __ballot(fits && (b == nT || (!sp && ((ln + 1) & 7u) == 0)));
                if (m) {
                    const uint32_t top = 63u - (uint32_t)__builtin_clzll(m);
                    best = seg_a + w0 + top + 1;
                    best_pass = (uint32_t)__builtin_amdgcn_readlane((int)est, (int)top);  // (top is wave-uniform)
                    best_wend = (uint32_t)__builtin_amdgcn_readlane((int)wsl, (int)top);
                }
                run += tot;
                if (__ballot(fits) == 0) break;
            }
            if (!best) { fail = 30; break; }
            seg_b = best; seg_pass = best_pass; wend = best_wend;
            multi = true;
        }
        const bool last_seg = seg_b == nT;
        const uint32_t n = seg_b - seg_a;
        const uint4* __restrict__ pc = pcg + seg_a;
        const uint4 rend = uniform4(pcg[seg_b]);  // record of the segment's end position (the terminator for the last segment)
        const uint32_t C = ((rend.x & 0xFFFFu) - seg_c) & 0xFFFFu;
        const uint32_t E = wend - sb;  // slots of the window [eo(seg_a), wend); slot E is the EOS node's (last segment)
        if (E >= 8190u || m_in > E) {  // the candidate records hold a slot's byte offset (slot * 8) in 16 bits: a shorter segment, or the next tier / the fused kernel
            if (budget > lds_bytes / 3 && !whole) { budget -= lds_bytes / 4; __syncthreads(); continue; }  // (a sentence taken for whole keeps its records where a segmented one dumps its nodes: the next tier sweeps it)
            fail = 26; break;
        }
        const uint4* __restrict__ nd = ndg + seg_c;

        Arena ar{g_smem, lds_bytes, 0, true};
        (void)ar.take<uint2>(E + 2);              // e_rec: the slot records
        uint2* cnd = ar.take<uint2>(C + 2);       // per candidate: {first cell of its matrix row (low half: its back pointer, once inserted), byte offset of its slot record | word_cost << 16}
        uint16_t* path = ar.take<uint16_t>(n + 4);  // the token path of the back-trace (tokens <= positions)
        const uint32_t sl_cap = rec_cap > 3 * kD + 4 ? rec_cap - (3 * kD + 2) : 0u;
        if (!ar.ok || sl_cap < 3) {  // the estimate was too low: try a shorter segment before giving up
            if (budget > lds_bytes / 3 && !whole) { budget -= lds_bytes / 4; __syncthreads(); continue; }  // (a sentence taken for whole keeps its records where a segmented one dumps its nodes: the next tier sweeps it)
            fail = 26; break;
        }
        const uint32_t offC = lds0 + (uint32_t)(reinterpret_cast<char*>(cnd) - g_smem);

        // the per-character records of the first 64 positions are requested now, ahead of the candidate loads: by the time the
        // reachability sweep wants them they have arrived (the sweep of a chunk then prefetches the next chunk's)
        uint4 rc_next = pc[ln < n ? ln : n], rn_next = pc[ln < n ? ln + 1 : n];
        // ---- load: candidates from global (every record carries its slot); EOS ----
        const uint32_t fld0 = 0xFFFEu - seg_c;  // own field of candidate c of this segment: fld0 - c
        for (uint32_t c0 = 0; c0 < C; c0 += 64 * 4) {  // 4 independent 16-byte loads per lane in flight
            uint4 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t c = c0 + u * 64 + ln;
                r[u] = nd[c < C ? c : 0u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t c = c0 + u * 64 + ln;
                if (c < C) {
                    const uint32_t es = (r[u].y >> 16) - sb;
                    // never inserted until a sweep step reaches its start position (exact mode: the field says so, the step writes it)
                    const uint32_t fld = exact ? 0xFFFFu : ((fld0 - c) & 0xFFFFu);
                    e_rec[es] = make_uint2((fld << 16) | (r[u].w >> 16), kDeadHi);
                    cnd[c] = make_uint2(r[u].x, (es << 3) | (r[u].y << 16));
                }
            }
        }
        if (last_seg && ln == 0) {
            // EOS pseudo candidate (insert_eos, lattice.rs:85-101): left_id 0 (matrix row 0), word cost 0
            cnd[C] = make_uint2(0u, E << 3);
            e_rec[E] = make_uint2(((fld0 - C) & 0xFFFFu) << 16, kDeadHi);
        }
        PROF_MARK(3);

        // ---- structural pre-pass (tokenizer.rs:106-138, control flow only) + pass records ----
        // Bit-serial sweep, all state in SGPRs.  w bit i <=> position p + 1 + i is the end of an inserted node;
        // cur <=> position p is reachable (has_previous_node, tokenizer.rs:108).  A visited position ORs its length
        // mask into w; a visited space run of r characters (ignore_space, tokenizer.rs:113-125) hands its visit over to
        // position p + r and drops the reachability of everything in between (the reference continues from
        // start_word + 1).  The length masks of 64 positions come straight from the per-character records in global
        // memory into one VGPR pair and are read with v_readlane.  The visited positions of a chunk become sweep steps,
        // every step is cut into passes of <= 16 candidates x <= 16 predecessors, and the pass records are laid out
        // contiguously (exclusive scan of the pass counts).  The state (w, cur, pend) is carried across segments: a non-final
        // segment is a multiple of 8 positions long, so the unrolled loop stops exactly at its end.
        uint32_t SL = 0, S = 0, sn_eos = n;
        bool windowed = true, overflow = false;
        uint64_t nx_w = 0;           // the state behind the segment's last position (committed at the hand-over: a segment may be retried shorter)
        uint32_t nx_cur = 0, nx_pend = 0;
        // pass P (candidate chunk k, round r of a step), written by the lane that owns the step: its issue half into record P, its
        // consume half into record P + kD
        auto put_pass = [&](uint32_t P, uint32_t p_beg, uint32_t np, uint32_t c_beg, uint32_t nc, uint32_t k, uint32_t r, uint32_t rounds, uint32_t first) {
            const uint32_t np_r = np - kRoundPreds * r < kRoundPreds ? np - kRoundPreds * r : kRoundPreds;
            const uint32_t nc_r = nc - kRoundCands * k < kRoundCands ? nc - kRoundCands * k : kRoundCands;
            const uint32_t nu = (np_r + 3u) >> 2, t = np_r - 4u * (nu - 1u);  // units, predecessors of the last one (1..4)
            const uint64_t cm = nc_r >= 16u ? ~0ull : (1ull << (4u * nc_r)) - 1ull;
            const uint32_t pat = t >= 4u ? 0xFFFFFFFFu : ((1u << t) - 1u) * 0x11111111u;  // phases below t, in every quad
            const uint64_t lm = cm & (((uint64_t)pat << 32) | pat);
            const uint32_t w0 = offK + ((p_beg + kRoundPreds * r) << 3), w1 = offC + ((c_beg + kRoundCands * k) << 3);
            const uint32_t ps = np >= 4u ? 0xFFFFFFFFu : ((1u << np) - 1u) * 0x11111111u;  // phases that see a predecessor in some unit of the step
            LPass& I = rec[P];
            I.w0 = w0; I.w1 = w1;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) I.m[i] = i + 1 < nu ? cm : (i + 1 == nu ? lm : 0ull);
            rec_w3[P] = np | (first ? 0x8000u : 0u) | (nc << 16);
            LPass& Cn = rec[P + kD];
            Cn.w0c = w0;
            Cn.w1c = w1 | (nu << 20) | (r == 0 ? 0x800000u : 0u) | (r + 1 == rounds ? 0x1000000u : 0u);
            Cn.lm = lm;
            Cn.vm = cm & (((uint64_t)ps << 32) | ps);
        };
        {
            uint64_t w = sw_w;
            uint32_t cur = sw_cur, pend = sw_pend, stop = 0;
            for (uint32_t chunk = 0; chunk < n && !stop; chunk += 64) {
                const uint32_t i = chunk + ln;
                const bool in = i < n;
                const uint4 rc = rc_next, rn = rn_next;  // this position's record and the next one's (n = the end record)
                if (chunk + 64 < n) { const uint32_t i2 = i + 64; rc_next = pc[i2 < n ? i2 : n]; rn_next = pc[i2 < n ? i2 + 1 : n]; }
                const bool is_space = kSpaceMode && in && (rc.y >> 31) != 0;
                const uint32_t l_lo = (in && !is_space) ? rc.z : 0u, l_hi = (in && !is_space) ? rc.w : 0u;
                const uint32_t gf = is_space ? rc.z : 0u;  // groupable run of a space position
                const uint64_t spm = kSpaceMode ? __ballot(is_space) : 0ull;
                const uint32_t cnt = n - chunk < 64 ? n - chunk : 64;
                uint64_t vis = 0, visp = 0;
#pragma unroll
                for (uint32_t k = 0; k < 64; ++k) {
                    if ((k & 7u) == 0 && k >= cnt) break;
                    const uint64_t bit = 1ull << k;
                    const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(l_hi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane(l_lo, k);
                    if constexpr (kSpaceMode) {
                        if (cur && !pend && (spm & bit)) {  // rare: a reachable space run
                            const uint32_t r = __builtin_amdgcn_readlane(gf, k);
                            if (chunk + k + r >= n) { sn_eos = chunk + k; stop = 1; w = 0; }  // only spaces left: EOS connects here
                            else if (r > 63) { windowed = false; stop = 1; w = 0; }
                            else {
                                visp |= bit;
                                w = (w & ~((1ull << r) - 1ull)) | (1ull << (r - 1));
                                pend = 1;
                            }
                        } else {
                            w |= cur ? m : 0ull;
                            vis |= (cur && !pend) ? bit : 0ull;
                            pend = cur ? 0u : pend;
                        }
                    } else {
                        w |= cur ? m : 0ull;
                        vis |= cur ? bit : 0ull;
                    }
                    cur = (uint32_t)w & 1u;
                    w >>= 1;
                }
                if (cnt < 64) { vis &= (1ull << cnt) - 1ull; visp &= (1ull << cnt) - 1ull; }
                const uint64_t any = vis | visp;
                // lanes = the visited positions of the chunk: step (start_node i, start_word sw)
                const bool step = (any >> ln) & 1ull;
                uint32_t xa = rc.x, xb = rn.x;  // candidate range of the start word
                if (kSpaceMode && ((visp >> ln) & 1ull)) {
                    const uint32_t sw = i + gf;  // < n: a run that reaches the end stops the sweep above
                    xa = pc[sw].x; xb = pc[sw + 1].x;
                }
                const uint32_t p_beg = (rc.x >> 16) - sb, np = ((rn.x >> 16) - (rc.x >> 16)) & 0xFFFFu;
                const uint32_t c_beg = ((xa & 0xFFFFu) - seg_c) & 0xFFFFu, nc = ((xb & 0xFFFFu) - (xa & 0xFFFFu)) & 0xFFFFu;
                const uint32_t rounds = (np + kRoundPreds - 1) / kRoundPreds;
                const uint32_t nsl = step ? rounds * ((nc + kRoundCands - 1) / kRoundCands) : 0u;
                uint32_t tot;
                const uint32_t ex = wave_exscan(nsl, tot);
                if (SL + tot + 2 > sl_cap) overflow = true;
                if (!overflow)
                    for (uint32_t q = 0, k = 0, r = 0; q < nsl; ++q) {
                        put_pass(SL + ex + q, p_beg, np, c_beg, nc, k, r, rounds, q == 0);
                        if (++r == rounds) { r = 0; ++k; }
                    }
                SL += tot;
                S += (uint32_t)__popcll(any);
            }
            nx_w = w; nx_cur = cur; nx_pend = pend;
        }
        if (!windowed) { fail = 27; break; }  // > 63 skipped spaces in a row: generic pre-pass of the fused kernel
        uint32_t eos_rec = 0;  // first pass record of the EOS step
        if (last_seg) {
            // + the EOS step (insert_eos(start_node), tokenizer.rs:138): predecessors = ends[sn_eos]
            const uint32_t y0 = __builtin_amdgcn_readfirstlane(pc[sn_eos].x) >> 16;
            const uint32_t y1 = sn_eos < n ? __builtin_amdgcn_readfirstlane(pc[sn_eos + 1].x) >> 16 : ET;
            const uint32_t p_beg = y0 - sb, np = y1 - y0;
            const uint32_t nsl = (np + kRoundPreds - 1) / kRoundPreds;
            if (SL + nsl + 2 > sl_cap) overflow = true;
            if (!overflow)
                for (uint32_t q = ln; q < nsl; q += 64) put_pass(SL + q, p_beg, np, C, 1u, 0u, q, nsl, q == 0);
            eos_rec = SL;
            SL += nsl;
            ++S;
        } else if (sn_eos != n) { fail = 31; break; }  // cannot happen: no cut follows a space
        prof_SL += SL; prof_S += S;
        if (overflow || SL >= (1u << 18)) {  // more passes than estimated (gen_candidates bounds them per position)
            if (budget > lds_bytes / 3 && !whole) { budget -= lds_bytes / 4; __syncthreads(); continue; }  // (a sentence taken for whole keeps its records where a segmented one dumps its nodes: the next tier sweeps it)
            fail = 29; break;
        }
        // Empty passes behind the last one (no units, no lanes): the sweep loop runs in trips of kD passes and reads kD + 1 records
        // ahead, i.e. up to record SL + 2 kD.  Their issue halves sit in the records [SL, SL + 2 kD + 2), their consume halves kD
        // records further on (the consume halves in [SL, SL + kD) are those of the last kD real passes).
        if (ln < 2 * kD + 2) {
            LPass& I = rec[SL + ln];
            I.w0 = offK; I.w1 = offC;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) I.m[i] = 0ull;
            LPass& Cn = rec[SL + kD + ln];
            Cn.w0c = offK; Cn.w1c = offC; Cn.lm = 0ull; Cn.vm = 0ull;
        }
        // the records are read back through the scalar cache: this wave's stores complete (workgroup scope: s_waitcnt vmcnt(0); the
        // vector L1 is write-through), then the scalar cache forgets whatever it holds of this region (an earlier segment's records)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt vmcnt(0)\n\ts_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        PROF_MARK(4);

        // ---- fused gather + cost recurrence (matrix_connector.rs:79-85, lattice.rs:103-151) ----
        auto recurrence = [&](auto exact_c) {
            constexpr bool kExact = decltype(exact_c)::value;
            // The connection matrix through a structured buffer resource (stride = one cell, index = left id * num_right + right id:
            // one SDWA add per gather, no 64-bit address per lane; num_records is set to the byte size, at least the cell count
            // under either reading of that field: lanes without a pair are masked off, nothing relies on the range check).  The gathers are
            // inline assembly: kUnits loads per pass whatever its shape, lanes without a pair masked off through EXEC -- so the number of
            // loads in flight is static and the one s_waitcnt per pass is exact.
            const uint64_t mb = (uint64_t)reinterpret_cast<uintptr_t>(D.matrix);
            u32x4 rsrc;
            rsrc.x = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)mb);
            rsrc.y = ((uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(mb >> 32)) & 0xFFFFu) | ((1u << kSh) << 16);
            rsrc.z = (uint32_t)__builtin_amdgcn_readfirstlane(D.matrix_bytes);
            rsrc.w = 0x00020000u;
            const uint32_t kk = ln & 3u, k8 = kk << 3, cl8 = (ln >> 2) << 3;
            constexpr uint64_t kPhase0 = 0x1111111111111111ull;  // the lanes that write a candidate's node: phase 0
            auto sel = [](uint64_t mask, uint32_t a, uint32_t b) { return __builtin_amdgcn_inverse_ballot_w64(mask) ? b : a; };  // bit ? b : a (v_cndmask on an SGPR mask)
            uint32_t word[kD][kUnits];      // VGPR ring: connection costs in flight (sign-extended), slot = pass % kD
            uint32_t best_hi = 0xFFFFFFFFu, best_lo = 0xFFFFFFFFu;
            // The pass records come through the scalar cache (s_load_dwordx16: constant address space).  The compiler treats such memory
            // as immutable, so the pointer is laundered behind the stores + s_dcache_inv above: no load of it can be moved in front of them.
            typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
            typedef __attribute__((address_space(4))) const u32x16 crec_t;
            uint64_t rbase = (uint64_t)reinterpret_cast<uintptr_t>(rec);
            rbase = uniform64(rbase);
            asm volatile("" : "+s"(rbase));
            const crec_t* const rq = reinterpret_cast<const crec_t*>(rbase);
            // issue side of a pass, part 1: this lane's addresses and the LDS reads the gathers need
            struct Iss { uint32_t leftidx, lo[kUnits]; };
            auto issue_reads = [&](const u32x16& pr) {
                Iss s;
                const uint32_t pa = pr[0] + k8, ca = pr[1] + cl8;
                s.leftidx = *reinterpret_cast<lds_cu32*>(ca);
#pragma unroll
                for (uint32_t i = 0; i < kUnits; ++i) s.lo[i] = *reinterpret_cast<lds_cu32*>(pa + 32u * i);  // low half: right id of predecessor 4 i + k (garbage behind the list: masked below)
                return s;
            };
            // part 2: the gathers into ring slot u.  Unit i runs under the EXEC mask its record holds: the lanes of the candidates that
            // exist while a later unit follows, the lanes that hold a pair as the last one, none behind it -- a load under EXEC = 0 moves
            // nothing and writes no register, but it takes its place in vmcnt (tools/calib/exec0_vmcnt.hip: 128 000 of 128 000 trials on
            // gfx950), so the count in flight stays static.  The empty passes behind the last one are never waited for: the counter is
            // drained behind the loop, before the ring's registers go back to the compiler -- a load that lands late must not find its
            // register reused (tools/check_ring_isa.py proves that on the compiled ISA).
            auto issue_gathers = [&](uint32_t u, const Iss& s, const u32x16& pr) {
                uint64_t m[kUnits];
                uint32_t vo[kUnits];
#pragma unroll
                for (uint32_t i = 0; i < kUnits; ++i) {
                    m[i] = ((uint64_t)pr[5 + 2 * i] << 32) | pr[4 + 2 * i];
                    vo[i] = (s.lo[i] & 0xFFFFu) + s.leftidx;
                }
#define VBT_LD(OP, I) "s_mov_b64 exec, %[m" #I "]\n\t" OP " %[d" #I "], %[a" #I "], %[rs], 0 idxen\n\t"
                if constexpr (kUnits == 4) {
#define VBT_GATHER(OP)                                                                                                        \
                    asm volatile(VBT_LD(OP, 0) VBT_LD(OP, 1) VBT_LD(OP, 2) VBT_LD(OP, 3) "s_mov_b64 exec, -1"                   \
                                 : [d0] "=&v"(word[u][0]), [d1] "=&v"(word[u][1]), [d2] "=&v"(word[u][kUnits - 2]), [d3] "=&v"(word[u][kUnits - 1]) \
                                 : [a0] "v"(vo[0]), [a1] "v"(vo[1]), [a2] "v"(vo[kUnits - 2]), [a3] "v"(vo[kUnits - 1]), [rs] "s"(rsrc),   \
                                   [m0] "s"(m[0]), [m1] "s"(m[1]), [m2] "s"(m[kUnits - 2]), [m3] "s"(m[kUnits - 1]))
                    if constexpr (kWide) VBT_GATHER("buffer_load_dword");
                    else VBT_GATHER("buffer_load_sshort");
#undef VBT_GATHER
                } else {
#define VBT_GATHER(OP)                                                                                                        \
                    asm volatile(VBT_LD(OP, 0) VBT_LD(OP, 1) "s_mov_b64 exec, -1"                                               \
                                 : [d0] "=&v"(word[u][0]), [d1] "=&v"(word[u][1])                                               \
                                 : [a0] "v"(vo[0]), [a1] "v"(vo[1]), [rs] "s"(rsrc), [m0] "s"(m[0]), [m1] "s"(m[1]))
                    if constexpr (kWide) VBT_GATHER("buffer_load_dword");
                    else VBT_GATHER("buffer_load_sshort");
#undef VBT_GATHER
                }
#undef VBT_LD
            };
            u32x16 pr = rq[0];  // the record in hand: issue half of the pass whose gathers go out next, consume half of the pass kD before it
#pragma unroll
            for (uint32_t p = 0; p < kD; ++p) {
                const u32x16 nx = rq[p + 1];
                const Iss s = issue_reads(pr);
                issue_gathers(p, s, pr);
                pr = nx;
                __builtin_amdgcn_sched_barrier(0);
            }
            for (uint32_t s0 = 0; s0 < SL; s0 += kD) {
                const crec_t* const rt = rq + s0;  // (the records of this trip sit at constant offsets from here)
#pragma unroll
                for (uint32_t u = 0; u < kD; ++u) {
                    // iteration si = s0 + u: pr = record si + kD
                    const uint32_t w2 = pr[3];
                    const uint64_t lm = ((uint64_t)pr[13] << 32) | pr[12], vm = ((uint64_t)pr[15] << 32) | pr[14];
                    const uint32_t nu = (w2 >> 20) & 7u;
                    const uint32_t pa = pr[2] + k8, ca = (w2 & 0xFFFFFu) + cl8;
                    // ---- all LDS reads of the iteration: what the issue side of pass si + kD needs, this pass's predecessor records, its
                    // candidate record ----
                    Iss is = issue_reads(pr);  // (first: the one place that waits for the record requested an iteration ago, with no LDS read in flight yet)
                    uint64_t kb[4];
                    kb[0] = *reinterpret_cast<lds_cu64*>(pa);
                    if (nu > 1u) {
                        kb[1] = *reinterpret_cast<lds_cu64*>(pa + 32u);
                        if constexpr (kUnits == 4)
                            if (nu > 2u) {
                                kb[2] = *reinterpret_cast<lds_cu64*>(pa + 64u);
                                kb[3] = *reinterpret_cast<lds_cu64*>(pa + 96u);
                            }
                    }
                    uint32_t cy = *reinterpret_cast<lds_cu32*>(ca + 4u);  // byte offset of the candidate's slot record | word cost << 16
                    // ---- the gathers of pass si have landed once at most those of the kD - 1 passes behind it are in flight ----
                    if constexpr (kUnits == 4)
                        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(word[u][0]), "+v"(word[u][1]), "+v"(word[u][kUnits - 2]), "+v"(word[u][kUnits - 1]) : "n"(kUnits * (kD - 1)));
                    else
                        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(word[u][0]), "+v"(word[u][1]) : "n"(kUnits * (kD - 1)));
                    // Every LDS read of the iteration has to be back before the next pass record is requested: scalar loads and LDS reads
                    // share one counter and return out of order with each other, so any later wait for LDS data would be a wait for the
                    // scalar load as well -- its whole latency on the pass's chain.  (The reads were issued back to back: the last one is
                    // a few cycles behind the first.)  Behind this point the iteration waits for nothing but the record itself, at the top
                    // of the next one.
                    if constexpr (kUnits == 4)
                        asm volatile("" : "+v"(kb[0]), "+v"(kb[1]), "+v"(kb[2]), "+v"(kb[3]), "+v"(cy), "+v"(is.leftidx), "+v"(is.lo[0]), "+v"(is.lo[1]), "+v"(is.lo[kUnits - 2]), "+v"(is.lo[kUnits - 1]));
                    else
                        asm volatile("" : "+v"(kb[0]), "+v"(kb[1]), "+v"(cy), "+v"(is.leftidx), "+v"(is.lo[0]), "+v"(is.lo[1]));
                    const u32x16 nrec = rt[u + kD + 1];
                    // ---- pass si ----
                    // the four phases of a candidate: minimum cost over the lanes that saw a predecessor, then among the lanes that hold
                    // it the smallest field (= the last inserted predecessor), by two quad-permute levels each; phase 0 adds the word
                    // cost and writes the node
                    auto finish_step = [&](uint32_t b_hi, uint32_t b_lo, uint64_t seen) {
                        const uint32_t v_hi = sel(seen, 0xFFFFFFFFu, b_hi);
                        const uint32_t m_hi = group_min_u32<2>(v_hi);
                        const uint32_t m_lo = group_min_u32<2>(v_hi == m_hi ? b_lo : 0xFFFFFFFFu);
                        // phase 0 of every candidate that exists writes: the node's cost into its slot record (+ word cost, lattice.rs:125),
                        // the winner's field as its back pointer (the low half of its candidate record) and, where dead predecessors are told
                        // by their field, its own field.  Inline assembly under an EXEC mask rather than a divergent `if`: with no
                        // divergent branch in the loop the compiler leaves its (all wave-uniform) control flow alone.
                        const uint64_t fm = vm & kPhase0;
                        const uint32_t sa = offK + (cy & 0xFFFFu);
                        const uint32_t cost = m_hi + (uint32_t)((int32_t)cy >> 16);
                        if constexpr (kExact) {
                            const uint32_t own = fld0 - ((ca - offC) >> 3);
                            asm volatile("s_mov_b64 exec, %[m]\n\tds_write_b32 %[a], %[v] offset:4\n\tds_write_b16 %[a], %[o] offset:2\n\t"
                                         "ds_write_b16_d16_hi %[c], %[b]\n\ts_mov_b64 exec, -1"
                                         :: [m] "s"(fm), [a] "v"(sa), [v] "v"(cost), [o] "v"(own), [c] "v"(ca), [b] "v"(m_lo) : "memory");
                        } else {
                            asm volatile("s_mov_b64 exec, %[m]\n\tds_write_b32 %[a], %[v] offset:4\n\tds_write_b16_d16_hi %[c], %[b]\n\ts_mov_b64 exec, -1"
                                         :: [m] "s"(fm), [a] "v"(sa), [v] "v"(cost), [c] "v"(ca), [b] "v"(m_lo) : "memory");
                        }
                    };
                    if ((w2 >> 20) == (1u | 8u | 16u)) {
                        // The common step -- at most 4 predecessors, at most 16 candidates: one unit that starts and ends the step -- straight
                        // through: add the connection cost, combine the phases, write the nodes.
                        const uint32_t hi = (uint32_t)(kb[0] >> 32) + word[u][0], lo = (uint32_t)kb[0];
                        uint64_t seen = vm;
                        if constexpr (kExact) seen &= __builtin_amdgcn_ballot_w64(lo < 0xFFFF0000u);
                        finish_step(hi, lo, seen);
                    } else if (nu) {
                        // A lane keeps the 64-bit minimum (cost, field) over the predecessors of its phase.  The first unit of a step's first
                        // round starts it in every lane -- a lane whose phase sees no predecessor in the whole step holds garbage until the
                        // combine at the end of the step masks it (vm) -- so nothing is reset in between; every unit before the last is
                        // full, the last one holds a pair in the lanes lm.
                        auto pair = [&](uint32_t i, uint32_t& hi, uint32_t& lo, uint64_t& alive) {
                            hi = (uint32_t)(kb[i] >> 32) + word[u][i < kUnits ? i : 0];  // wrapping i32 add of the connection cost (lattice.rs:139)
                            lo = (uint32_t)kb[i];                                         // the predecessor's own field | right id
                            alive = ~0ull;
                            if constexpr (kExact) alive = __builtin_amdgcn_ballot_w64(lo < 0xFFFF0000u);  // never inserted: field 0xFFFF
                        };
                        auto unit = [&](uint32_t i) {
                            uint32_t hi, lo;
                            uint64_t alive;
                            pair(i, hi, lo, alive);
                            const uint64_t nk = ((uint64_t)hi << 32) | lo, bk = ((uint64_t)best_hi << 32) | best_lo;
                            uint64_t lt = __builtin_amdgcn_ballot_w64(nk < bk) & (nu == i + 1u ? lm : ~0ull);
                            if constexpr (kExact) lt &= alive;
                            best_hi = sel(lt, best_hi, hi);
                            best_lo = sel(lt, best_lo, lo);
                        };
                        if (w2 & 0x800000u) {
                            uint32_t hi, lo;
                            uint64_t alive;
                            pair(0, hi, lo, alive);
                            if constexpr (kExact) { best_hi = sel(alive, 0xFFFFFFFFu, hi); best_lo = sel(alive, 0xFFFFFFFFu, lo); }
                            else { best_hi = hi; best_lo = lo; }
                        } else unit(0);
                        if (nu > 1u) {
                            unit(1);
                            if constexpr (kUnits == 4)
                                if (nu > 2u) {
                                    unit(2);
                                    if (nu > 3u) unit(3);
                                }
                        }
                        if (w2 & 0x1000000u) finish_step(best_hi, best_lo, vm);
                    }
                    // LDS operations of one wave execute in order: a compiler-level fence is all the next pass needs
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    // ---- gathers of pass si + kD into the ring slot this pass has just left ----
                    issue_gathers(u, is, pr);
                    pr = nrec;
                }
            }
            // the last gathers in flight are those of the empty passes (EXEC = 0: they retire at once): done before the ring's registers
            // go back to the compiler
#pragma unroll
            for (uint32_t u = 0; u < kD; ++u) {
                if constexpr (kUnits == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(word[u][0]), "+v"(word[u][1]), "+v"(word[u][kUnits - 2]), "+v"(word[u][kUnits - 1]));
                else asm volatile("s_waitcnt vmcnt(0)" : "+v"(word[u][0]), "+v"(word[u][1]));
            }
        };
        if (exact) recurrence(std::true_type{}); else recurrence(std::false_type{});
        PROF_MARK(6);

        // a node of this segment: its cost word and its back pointer (sequence of its best predecessor)
        auto node_cost = [&](uint32_t c) { return e_rec[(cnd[c].y & 0xFFFFu) >> 3].y ^ 0x80000000u; };
        auto node_pred = [&](uint32_t c) { return 0xFFFEu - (cnd[c].x & 0xFFFFu); };
        if (multi) {
            // leave (total cost, back pointer) of every node of the segment in the sentence's (dead) hit-staging region
            uint2* __restrict__ nb = reinterpret_cast<uint2*>(A.g_hits + node0) + seg_c;
            for (uint32_t c = ln; c < C; c += 64) nb[c] = make_uint2(node_cost(c), node_pred(c));
        }
        if (A.lid_count) {
            // Lattice::add_connid_counts (lattice.rs:170-183): for every inserted node r and every node l in
            // ends[r.start_node]: lid_count[r.left_id] += 1, rid_count[l.right_id] += 1; then the same for EOS
            // (left_id 0) against ends[len_char].  Only inserted ("live") nodes exist in the reference's lists.
            // s_counted[sid] remembers how far the sentence has been counted, so a retry in an escape tier or in the
            // fused kernel never counts a step twice.
            const uint32_t c_skip = counted >= nT ? CT : __builtin_amdgcn_readfirstlane(pcg[counted].x) & 0xFFFFu;  // candidates are in start order
            for (uint32_t k = 0; k < SL; ++k) {
                const uint32_t w3 = __builtin_amdgcn_readfirstlane(rec_w3[k]);
                if (!(w3 & 0x8000u)) continue;  // one record per step: its first pass
                const uint4 r = make_uint4(__builtin_amdgcn_readfirstlane(rec[k].w0), __builtin_amdgcn_readfirstlane(rec[k].w1), 0u, 0u);
                const uint32_t c_beg = (r.y - offC) >> 3, nc = w3 >> 16, np = w3 & 0x7FFFu;
                const bool eos_step = last_seg && k >= eos_rec;
                if (eos_step ? counted > nT : seg_c + c_beg < c_skip) continue;
                uint32_t p_beg = (r.x - offK) >> 3, p_end = p_beg + np;
                if (eos_step) { p_beg = (rend.x >> 16) - sb; p_end = E; }  // EOS pairs with ends[len_char]
                uint32_t live = 0;
                for (uint32_t j0 = p_beg; j0 < p_end; j0 += 64) {
                    const uint32_t j = j0 + ln;
                    const uint2 er = j < p_end ? e_rec[j] : make_uint2(0xFFFF0000u, kDeadHi);
                    const bool alive = j < p_end && (exact ? (er.x >> 16) != 0xFFFFu : er.y != kDeadHi);
                    live += (uint32_t)__popcll(__ballot(alive));
                    if (alive) atomicAdd(&A.rid_count[er.x & 0xFFFFu], (unsigned long long)nc);
                }
                if (eos_step) { if (ln == 0) atomicAdd(&A.lid_count[0], (unsigned long long)live); }
                else for (uint32_t c = c_beg + ln; c < c_beg + nc; c += 64) atomicAdd(&A.lid_count[nd[c].x / D.num_right], (unsigned long long)live);
            }
            const uint32_t upto = last_seg ? nT + 1 : seg_b;
            if (upto > counted) { counted = upto; if (ln == 0) A.s_counted[sid] = counted; }
        }
        if (!last_seg) {
            // hand-over: the slots behind the cut -- final nodes that start in front of it (and the still untouched slots of later
            // candidates among them) -- move to the front of the window, 64 records at a time, ascending (the destination of a
            // chunk never reaches the source of a later one)
            const uint32_t i0 = (rend.x >> 16) - sb, m_out = E - i0;
            for (uint32_t k0 = 0; k0 < m_out; k0 += 64) {
                const uint32_t k = k0 + ln;
                const uint2 r = e_rec[i0 + (k < m_out ? k : 0u)];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (k < m_out) e_rec[k] = r;
            }
            m_in = m_out;
            sw_w = nx_w; sw_cur = nx_cur; sw_pend = nx_pend;
            sb = rend.x >> 16; seg_a = seg_b; seg_c += C; seg_p += seg_pass;
            budget = lds_bytes;
            __syncthreads();
            continue;
        }
        done = true;

        // ---- back-trace + token records (append_top_nodes lattice.rs:159-168, token.rs:21-92) ----
        // A token starts where its best predecessor ends -- behind the space run there, if that position is a skipped
        // space (tokenizer.rs:113-125) -- so a lane needs its own candidate record and the previous token's.
        uint32_t T = 0;
        const uint16_t* __restrict__ c2b = A.g_c2b + slot0;
        auto start_of = [&](uint32_t prev_end) {
            if constexpr (kSpaceMode) {
                if (prev_end < nT) {
                    const uint4 rp = pcg[prev_end];
                    if (rp.y >> 31) return prev_end + rp.z;
                }
            }
            return prev_end;
        };
        if (!multi) {
            // the walk along the back pointers is serial: lane 0, one LDS round trip per token
            if (ln == 0) {
                uint32_t seq = node_pred(C);
                while (seq != kBosSeq && T < n) { path[T++] = (uint16_t)seq; seq = node_pred(seq); }
            }
            T = (uint32_t)__builtin_amdgcn_readfirstlane((int)T);
            __syncthreads();
            if (ln == 0) { A.tok_cnt[sid] = T; if (T) atomicAdd(&A.tile_sums[sid / kScanTile], T); }
            for (uint32_t t = ln; t < T; t += 64) {
                const uint32_t c = path[T - 1 - t];
                const uint4 r = ndg[c];
                const uint32_t prev_end = t ? ndg[path[T - t]].w & 0xFFFFu : 0u;
                const uint32_t stp = start_of(prev_end), en = r.w & 0xFFFFu;
                vbt_token_rec o;
                o.start_char = stp; o.end_char = en;
                o.start_byte = c2b[stp]; o.end_byte = c2b[en];
                o.word_idx = r.z;
                o.total_cost = (int32_t)node_cost(c);
                A.tok_stage[slot0 + t] = o;  // the sentence's own staging region: no allocation atomic (compact_tokens packs them)
            }
        } else {
            // segmented sentence: pull all back pointers into LDS (the arena is free now), walk, emit from global
            const uint32_t back_eos = node_pred(C);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's own dumps: stores complete
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            Arena a2{g_smem, lds_bytes, 0, true};
            uint16_t* path = a2.take<uint16_t>(nT + 1);
            uint16_t* back = a2.take<uint16_t>(0);
            const uint32_t W = a2.ok && lds_bytes > a2.used + 64 ? (uint32_t)((lds_bytes - a2.used - 64) / 2) : 0u;  // window of back pointers
            if (W < 1024) { fail = 33; break; }
            const uint2* __restrict__ nbg = reinterpret_cast<const uint2*>(A.g_hits + node0);
            // Back pointers only point backwards: walk from EOS, pulling windows of them [win_lo, win_hi) into LDS on demand.
            uint32_t seq = back_eos, win_lo = CT + 2;
            while (seq != kBosSeq && T < nT) {
                if (seq < win_lo) {
                    const uint32_t hi = seq + 1, lo = hi > W ? hi - W : 0u;
                    __syncthreads();
                    for (uint32_t c0 = lo; c0 < hi; c0 += 64 * 8) {
                        uint32_t v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { const uint32_t c = c0 + u * 64 + ln; v[u] = c < hi ? nbg[c].y : 0u; }
#pragma unroll
                        for (int u = 0; u < 8; ++u) { const uint32_t c = c0 + u * 64 + ln; if (c < hi) back[c - lo] = (uint16_t)v[u]; }
                    }
                    __syncthreads();
                    win_lo = lo;
                }
                if (ln == 0) {
                    while (seq != kBosSeq && seq >= win_lo && T < nT) {
                        path[T++] = (uint16_t)seq;
                        seq = back[seq - win_lo];
                    }
                }
                seq = __builtin_amdgcn_readfirstlane(seq);
                T = __builtin_amdgcn_readfirstlane(T);
            }
            __syncthreads();
            T = (uint32_t)__builtin_amdgcn_readfirstlane((int)T);
            __syncthreads();
            if (ln == 0) { A.tok_cnt[sid] = T; if (T) atomicAdd(&A.tile_sums[sid / kScanTile], T); }
            for (uint32_t t = ln; t < T; t += 64) {
                const uint32_t c = path[T - 1 - t];
                const uint4 r = ndg[c];
                const uint32_t prev_end = t ? ndg[path[T - t]].w & 0xFFFFu : 0u;
                const uint32_t stp = start_of(prev_end), en = r.w & 0xFFFFu;
                vbt_token_rec o;
                o.start_char = stp; o.end_char = en;
                o.start_byte = c2b[stp]; o.end_byte = c2b[en];
                o.word_idx = r.z;
                o.total_cost = (int32_t)nbg[c].x;
                A.tok_stage[slot0 + t] = o;
            }
        }
        }  // segments
        if (fail) return fail;
        PROF_MARK(7);
        if (A.prof && ln == 0) {
            atomicAdd(&pr_[kProfPhases + 1], (unsigned long long)prof_S);
            atomicAdd(&pr_[kProfPhases + 2], (unsigned long long)prof_SL);
            atomicAdd(&pr_[kProfPhases + 3], (unsigned long long)CT);
        }
#undef PROF_MARK
    }
    return 0;
}

template <bool kSpaceMode, bool kWide>
__global__ void __launch_bounds__(64, VBT_LAT_WAVES) lattice_lds(DevDict D, BatchArgs A, uint32_t tier, uint32_t persistent) {
    const uint32_t ln = threadIdx.x;
    // long sentences are the critical path of a batch: let their waves win issue arbitration
    if (A.tier_prio && (A.seg_tier < A.n_tiers ? tier >= A.seg_tier : tier + A.tier_prio >= A.n_tiers)) __builtin_amdgcn_s_setprio(2);
    const int src = (int)tier;  // the tier's own list
    const uint32_t* list = A.lists + (size_t)src * A.list_stride + A.list_off;
    const uint32_t count = A.cctrl[2 * src];
    uint32_t* cursor = &A.cctrl[2 * src + 1];
    // Work distribution: one list entry per workgroup (the grid covers the batch; a returning atomic on a hot word costs
    // ~11 ns of a serial resource, which bounds a kernel at ~88 M entries/s however fast the waves are), or -- escape tiers,
    // whose lists are short -- persistent waves that draw entries from a cursor.
    bool first_item = true;
    for (;;) {
        uint32_t item = blockIdx.x;  // (persistent waves too: their first item is their own index, see tokenize_global)
        if (persistent && !first_item) {
            if (ln == 0) item = gridDim.x + atomicAdd(cursor, 1u);
            item = __builtin_amdgcn_readfirstlane(item);  // lane 0 is always active here; keeps everything below scalar
        } else if (!first_item) break;
        first_item = false;
        if (item >= count) break;
        // newest entries first: the large-LDS generator levels append their (long, slow) sentences last, level by level,
        // so reading the list backwards starts the longest sentences first instead of leaving them as the tail
        const uint32_t sid = __builtin_amdgcn_readfirstlane(list[count - 1 - item]);
        const uint32_t fail = lattice_sentence<kSpaceMode, kWide>(D, A, tier, sid);
        if (fail) {
            // Could not be swept here (no admissible cut, estimates too low, ...): the next escape tier -- more LDS,
            // launched behind this one -- retries; after the last one the fused kernel with the global-memory
            // lattice redoes the sentence.
            const bool escape = tier >= A.seg_tier && tier + 1 < A.n_tiers && fail != 27;
            if (ln == 0 && !escape) atomicAdd(&A.ctrl[fail < 32 ? fail : 28], 1u);
            if (escape) list_push(A, tier + 1, sid); else list_push_fb(A, sid);
        }
        __syncthreads();
    }
}

// Worker::tokenize() latency path (worker.rs:49-55; the 3-call loop of tokenize/src/main.rs:78-82): ONE launch, one wavefront, one
// sentence.  The text comes straight out of the worker's pinned host block (`h_text`, one PCIe round trip: 16 bytes per lane per
// request into a device copy), the generator and the sweep run back to back in the same wave (what gen_one leaves in global memory is
// read back by the wave that wrote it: a workgroup-scope fence is all it takes), and the token records, their count and the status
// word go straight back into pinned host memory (posted writes): no copy engine, no second launch, no allocation.  status: 0 = done,
// 1 = this sentence needs the batch pipeline (longer than the generator's LDS, unsweepable in segments, a word > 64 characters...).
template <bool kSpaceMode, bool kWide>
__global__ void __launch_bounds__(64) tokenize_one(DevDict D, BatchArgs A, uint32_t lds_bytes, const uint8_t* h_text, uint32_t nb, uint32_t* status) {
    const uint32_t ln = threadIdx.x;
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(h_text);  // (the pinned block is padded to 16 bytes)
        u32x4* dst = reinterpret_cast<u32x4*>(const_cast<uint8_t*>(A.text));
        for (uint32_t i = ln; i * 16 < nb; i += 64) dst[i] = src[i];
        uint64_t* offs = const_cast<uint64_t*>(A.offsets);
        if (ln == 0) { offs[0] = 0; offs[1] = nb; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    gen_one(D, A, 0u, lds_bytes);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t tier = __builtin_amdgcn_readfirstlane((uint32_t)A.s_tier[0]);
    uint32_t st = 0;
    if (tier == 0u) st = lattice_sentence<kSpaceMode, kWide>(D, A, 0u, 0u) ? 1u : 0u;
    else if (tier != 0xFFu) st = 1u;  // 0xFF: an empty sentence, tok_cnt = 0 is already written
    // every lane's token stores have to be visible to the host before the status word is (the host may spin on it instead of
    // waiting for the stream): system-scope release by all lanes, then one releasing store
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ln == 0) __hip_atomic_store(status, st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void push_overflow(uint32_t* list, uint32_t* counter, uint32_t sid) {
    if (threadIdx.x == 0) list[atomicAdd(counter, 1u)] = sid;
}

// LDS tiers: one wavefront per sentence, lattice in `lds_bytes` of LDS.  in_list == nullptr: the
// grid covers all sentences (block b = sentence b); otherwise persistent waves drain in_list.
// Sentences that do not fit go to out_list for the next (larger) tier.
template <bool kWide>
__global__ void __launch_bounds__(64) tokenize_lds(DevDict D, BatchArgs A, uint32_t lds_bytes, const uint32_t* in_list,
                                                   const uint32_t* in_count, uint32_t* cursor, uint32_t* out_list,
                                                   uint32_t* out_count) {
    if (in_list == nullptr) {
        if (batch_rejected(A)) return;
        const uint32_t sid = blockIdx.x;
        if (process_sentence<uint16_t, false, kWide>(D, A, sid, g_smem, lds_bytes) != 0) push_overflow(out_list, out_count, sid);
        return;
    }
    const uint32_t count = *in_count;
    for (;;) {
        uint32_t k = 0;
        if (threadIdx.x == 0) k = atomicAdd(cursor, 1u);
        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
        if (k >= count) break;
        const uint32_t sid = in_list[k];
        if (process_sentence<uint16_t, false, kWide>(D, A, sid, g_smem, lds_bytes) != 0) push_overflow(out_list, out_count, sid);
        __syncthreads();
    }
}

// Last tier: persistent waves, lattice in a private global-memory slab (any sentence length).
template <bool kWide>
__global__ void __launch_bounds__(64) tokenize_global(DevDict D, BatchArgs A, const uint32_t* in_list, const uint32_t* in_count,
                                                      uint32_t* cursor) {
    const uint32_t count = *in_count;
    char* slab = nullptr;
    uint64_t slab_bytes = 0;
    unsigned long long* bump = reinterpret_cast<unsigned long long*>(&A.ctrl[kBump]);
    // (persistent waves: the first item of a workgroup is its own index, the following ones come from the cursor -- a launch
    // whose every workgroup opens with an atomic on the one cursor word pays ~11 ns per workgroup before any work starts)
    for (bool first = true;; first = false) {
        uint32_t k = blockIdx.x;
        if (!first) {
            if (threadIdx.x == 0) k = gridDim.x + atomicAdd(cursor, 1u);
            k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
        }
        if (k >= count) break;
        const uint32_t sid = in_list[k];
        for (int attempt = 0; attempt < 5; ++attempt) {
            const uint64_t need = process_sentence<uint32_t, true, kWide>(D, A, sid, slab, slab_bytes);
            if (need == 0) break;
            bool failed = need == kNoFit || attempt == 4;
            if (!failed) {  // grow: take a fresh slab from the bump arena
                uint64_t want = need + need / 4 + 4096;
                want = (want + 255) & ~255ull;
                unsigned long long off = 0;
                if (threadIdx.x == 0) off = atomicAdd(bump, (unsigned long long)want);
                off = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(off >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off);
                if (off + want > A.scratch_bytes) failed = true;
                else { slab = A.scratch + off; slab_bytes = want; }
            }
            if (failed) {
                if (threadIdx.x == 0) {
                    atomicOr(&A.ctrl[kError], need == kNoFit ? (uint32_t)kErrTooLong : (uint32_t)kErrScratch);
                    A.tok_cnt[sid] = 0;
                }
                break;
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

// Token compaction.  The sweep kernels leave the tokens of sentence s in its own region of the staging buffer and its
// count in tok_cnt[s], and add the count to the total of the sentence's tile (tile_sums[s / kScanTile]); compact_tokens turns
// that into the compact result: tok_off = exclusive prefix of the counts (so token ranges are in sentence order), the records
// packed back to back, the total in ctrl[kTotal].  (tok_tile_scan: the prefix over the tiles as a kernel of its own, for batches
// of more than 2048 tiles and for callers that pack later.)
constexpr uint32_t kScanItems = 1;
static_assert(kScanTile == kScanBlock * kScanItems, "one sentence per thread of a packing workgroup");
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t* warp_sums, uint32_t& block_total) {
    uint32_t wtot;
    const uint32_t ex = wave_exscan_any(v, wtot);
    const uint32_t w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63u) == 0) warp_sums[w] = wtot;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t i = 0; i < nw; ++i) { const uint32_t t = warp_sums[i]; if (i < w) base += t; tot += t; }
    __syncthreads();
    block_total = tot;
    return base + ex;
}
__global__ void __launch_bounds__(1024) tok_tile_scan(BatchArgs A, uint32_t* tile_sums, uint32_t n_tiles) {
    __shared__ uint32_t ws[16];
    uint32_t running = 0;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 1024) {
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < n_tiles ? tile_sums[t] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exscan(v, ws, tot);
        if (t < n_tiles) tile_sums[t] = running + ex;
        running += tot;
    }
    if (threadIdx.x == 0) A.ctrl[kTotal] = running;
}
// kPackSplit (8) workgroups share a tile: each redoes the tile's (cheap) offset scan and copies every kPackSplit-th stripe of its
// tokens -- the copy is a chain of dependent round trips per token (which sentence, where its slot starts, the record), so it
// wants many waves in flight: one workgroup per tile left 6 waves on a CU and took 73 us for the 68 MB of the headline batch.
#ifndef VBT_PACK_SPLIT
#define VBT_PACK_SPLIT 8
#endif
constexpr uint32_t kPackSplit = VBT_PACK_SPLIT;
// `scanned` = 0: tile_sums[] still holds the totals per tile -- every workgroup adds up the tiles in front of its own (a few
// hundred words out of L2: cheaper than a launch of the scan kernel in front of this one; the host picks the scan kernel for
// batches of more than 2048 tiles) and the first one leaves the grand total in ctrl[kTotal].
__global__ void __launch_bounds__(kScanBlock) compact_tokens(BatchArgs A, const uint32_t* tile_sums, uint32_t n_tiles, uint32_t scanned) {
    __shared__ uint32_t ws[kScanBlock / 64];
    __shared__ uint32_t red[2][kScanBlock / 64];
    __shared__ uint32_t offs[kScanTile + 1];  // exclusive token offsets of the tile's sentences, relative to the tile
    __shared__ uint64_t slot[kScanTile];      // first staging slot of each sentence of the tile (sentence_slot)
    if (A.ctrl[kError] & (uint32_t)kErrFatal) return;
    const uint32_t tile = blockIdx.x / kPackSplit, part = blockIdx.x % kPackSplit;
    const uint32_t tile0 = tile * kScanTile, s0 = tile0 + threadIdx.x * kScanItems;
    const uint64_t o0 = A.offsets[0];
    uint32_t c[kScanItems], v = 0;
    for (uint32_t i = 0; i < kScanItems; ++i) {
        c[i] = s0 + i < A.n ? A.tok_cnt[s0 + i] : 0u;
        v += c[i];
        slot[threadIdx.x * kScanItems + i] = s0 + i < A.n ? (A.offsets[s0 + i] - o0) + (uint64_t)kSentenceSlack * (s0 + i) : 0ull;
    }
    uint32_t tot;
    uint32_t ex = block_exscan(v, ws, tot);
    uint32_t base;
    if (scanned) base = tile_sums[tile];
    else {
        uint32_t before = 0, all = 0;
        for (uint32_t t = threadIdx.x; t < n_tiles; t += kScanBlock) { const uint32_t x = tile_sums[t]; all += x; before += t < tile ? x : 0u; }
        before = wave_sum(before);
        all = wave_sum(all);
        if ((threadIdx.x & 63u) == 0) { red[0][threadIdx.x >> 6] = before; red[1][threadIdx.x >> 6] = all; }
        __syncthreads();
        before = all = 0;
        for (uint32_t w = 0; w < kScanBlock / 64; ++w) { before += red[0][w]; all += red[1][w]; }
        base = before;
        if (blockIdx.x == 0 && threadIdx.x == 0) A.ctrl[kTotal] = all;
    }
    for (uint32_t i = 0; i < kScanItems; ++i) {
        offs[threadIdx.x * kScanItems + i] = ex;
        if (part == 0 && s0 + i < A.n) A.tok_off[s0 + i] = base + ex;
        ex += c[i];
    }
    if (threadIdx.x == 0) offs[kScanTile] = tot;
    __syncthreads();
    const uint64_t* __restrict__ src = reinterpret_cast<const uint64_t*>(A.tok_stage);
    uint64_t* __restrict__ dst = reinterpret_cast<uint64_t*>(A.tokens);
    for (uint32_t k = part * kScanBlock + threadIdx.x; k < tot; k += kScanBlock * kPackSplit) {
        uint32_t lo = 0, hi = kScanTile;  // last sentence of the tile whose offset is <= k (empty sentences share offsets: take the last)
        while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (offs[mid] <= k) lo = mid; else hi = mid; }
        const size_t from = (size_t)slot[lo] + (k - offs[lo]);
        const size_t to = (size_t)base + k;
#pragma unroll
        for (int w = 0; w < 3; ++w) dst[3 * to + w] = src[3 * from + w];
    }
}

// The same packing, written straight into the caller's (pinned, device-mapped) host block: tok_off, tok_cnt and the token
// records leave the GPU as the kernel's own stores -- posted PCIe writes, one fully coalesced 8-byte word per lane -- instead
// of a copy command behind the kernels (the runtime serves a device -> pinned-host hipMemcpyAsync with a copy KERNEL of
// ~1.3 ms for the 68 MB of the headline batch, serialised behind the batch's kernels: profiles/r03_h2h_timeline.md).
__global__ void __launch_bounds__(kScanBlock) compact_tokens_out(BatchArgs A, const uint32_t* tile_sums, uint32_t n_tiles, vbt_token_rec* out_tokens, uint32_t* out_off,
                                                                 uint32_t* out_cnt) {
    __shared__ uint32_t ws[kScanBlock / 64];
    __shared__ uint32_t offs[kScanTile + 1];
    if (A.ctrl[kError] & (uint32_t)kErrFatal) return;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {  // (the grid may be smaller than the tile count: VBT_PACK_WGS)
        const uint32_t tile0 = tile * kScanTile, s = tile0 + threadIdx.x;
        const uint32_t c = s < A.n ? A.tok_cnt[s] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exscan(c, ws, tot);
        const uint32_t base = tile_sums[tile];
        offs[threadIdx.x] = ex;
        if (s < A.n) { out_off[s] = base + ex; out_cnt[s] = c; }
        if (threadIdx.x == 0) offs[kScanTile] = tot;
        __syncthreads();
        const uint64_t o0 = A.offsets[0];
        const uint64_t* __restrict__ src = reinterpret_cast<const uint64_t*>(A.tok_stage);
        uint64_t* __restrict__ dst = reinterpret_cast<uint64_t*>(out_tokens) + 3 * (size_t)base;
        for (uint32_t w = threadIdx.x; w < 3 * tot; w += kScanBlock) {  // word w of the tile's packed records: token w / 3, part w % 3
            const uint32_t k = w / 3, part = w - 3 * k;
            uint32_t lo = 0, hi = kScanTile;  // last sentence of the tile whose offset is <= k
            while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (offs[mid] <= k) lo = mid; else hi = mid; }
            const uint32_t sn = tile0 + lo;
            const size_t from = (size_t)(A.offsets[sn] - o0) + (size_t)kSentenceSlack * sn + (k - offs[lo]);
            dst[w] = src[3 * from + part];
        }
        __syncthreads();  // offs[] is rewritten by the next tile
    }
}

// Compact connectors (RawConnector / DualConnector) are materialised once, when the tokenizer is created: one thread per
// (left, right) id pair evaluates the reference's cost function -- Scorer::accumulate_cost over the pair's feature rows
// (connector/raw_connector/scorer.rs:327-345), plus the small matrix over mapped ids for a dual connector
// (dual_connector.rs:267-279) -- into the dense i16 matrix the sweep reads.  288 GB of HBM make the reference's memory /
// speed trade-off moot: the hot path is the MatrixConnector's for every dictionary.
struct DevConnector {
    const uint32_t* bases; const uint32_t* checks; const int32_t* costs;
    uint32_t n_bases, n_checks;
    const uint32_t* right_feats; const uint32_t* left_feats;
    uint32_t width;
    const int16_t* m; const uint16_t* right_map; const uint16_t* left_map;  // dual only (m == nullptr: raw)
    uint32_t m_num_right;
};
template <typename CellT>
__global__ void __launch_bounds__(256) expand_connector(DevConnector c, CellT* out, uint32_t num_right, uint32_t num_left, uint32_t* range_flag) {
    const uint32_t right = blockIdx.x * 256 + threadIdx.x, left = blockIdx.y;
    if (right >= num_right) return;
    const uint32_t* __restrict__ k1 = c.right_feats + (size_t)right * c.width;
    const uint32_t* __restrict__ k2 = c.left_feats + (size_t)left * c.width;
    uint32_t sum = 0;  // wrapping i32
    for (uint32_t t = 0; t < c.width; ++t) {
        const uint32_t a = k1[t], b = k2[t];
        if (a < c.n_bases) {
            const uint32_t pos = c.bases[a] ^ b;
            if (pos < c.n_checks && c.checks[pos] == a) sum += (uint32_t)c.costs[pos];
        }
    }
    if (c.m) sum += (uint32_t)(int32_t)c.m[(size_t)c.left_map[left] * c.m_num_right + c.right_map[right]];
    const int32_t v = (int32_t)sum;
    if (v < -32768 || v > 32767) atomicOr(range_flag, 1u);
    out[(size_t)left * num_right + right] = (CellT)v;
}

template <typename T>
T* dev_upload(const std::vector<T>& v, std::vector<void*>& allocs) {
    T* p = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), bytes));
    allocs.push_back(p);
    if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* s = std::getenv(name);
    return s && *s ? (uint32_t)std::strtoul(s, nullptr, 10) : dflt;
}

}  // namespace

// ------------------------------------------------------------------ Tokenizer

Tokenizer::Tokenizer(const Dictionary* dict, bool ignore_space, uint32_t max_grouping_len, int device) : dict_(dict) {
    setenv("GPU_MAX_HW_QUEUES", "10", 0);  // one stream per LDS tier (no effect once HIP is initialised)
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        throw Error(VBT_ERR_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e) +
                                        " (libvibrato_hip has no CPU fallback)");
    if (device < 0) HIP_CHECK(hipGetDevice(&device));
    if (device >= count) throw Error(VBT_ERR_INVALID_ARGUMENT, "device: index out of range");
    HIP_CHECK(hipSetDevice(device));
    device_ = device;
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        throw Error(VBT_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");

    dev_.space_cateset = 0;
    if (ignore_space) {  // Tokenizer::ignore_space, tokenizer.rs:42-55
        int id = dict_->cate_id("SPACE");
        if (id < 0) throw Error(VBT_ERR_INVALID_ARGUMENT, "dict: SPACE is not defined in the input dictionary (i.e., char.def).");
        dev_.space_cateset = 1u << id;
    }
    dev_.max_grouping_len = max_grouping_len ? max_grouping_len : 0xFFFFFFFFu;  // tokenizer.rs:67-74
    try {
        upload_lexicon(dict_->system, dev_.sys);
        dev_.has_user = dict_->has_user ? 1 : 0;
        if (dict_->has_user) upload_lexicon(dict_->user, dev_.user);
        else dev_.user = dev_.sys;
        {   // lattice_lds addresses cells with 32-bit byte offsets through a buffer resource (checked before anything is allocated)
            const uint64_t bytes = (uint64_t)dict_->num_left * dict_->num_right * 2;
            if (bytes >= (1ull << 32)) throw Error(VBT_ERR_UNSUPPORTED, "connector: connection matrices of 4 GiB and more are not supported by the device image");
            dev_.matrix_bytes = (uint32_t)bytes;
        }
        if (dict_->conn_kind == kConnMatrix) {  // + 1 element: lattice_lds reads the aligned 32-bit word around a cell
            std::vector<int16_t> padded(dict_->matrix);
            padded.push_back(0);
            dev_.matrix = dev_upload(padded, allocs_);
        } else {
            // Compact connectors: evaluated once into a dense matrix (expand_connector).  i16 cells when every cost fits (the
            // released compact dictionaries: their costs are the matrix.def values of the full ones), else i32 cells -- the
            // reference's Raw / Dual cost is an i32 sum (raw_connector.rs:153-161, dual_connector.rs:267-279) -- read by the
            // kWide instances of the sweep kernels.
            const bool is_dual = dict_->conn_kind == kConnDual;
            const Scorer& sc = is_dual ? dict_->dual.scorer : dict_->raw.scorer;
            const size_t cells = (size_t)dict_->num_left * dict_->num_right;
            std::vector<void*> tmp;  // the compact structures are only needed by the expansion
            try {
                uint32_t* flag = nullptr;
                HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&flag), 16));
                tmp.push_back(flag);
                HIP_CHECK(hipMemset(flag, 0, 16));
                DevConnector c{};
                c.bases = dev_upload(sc.bases, tmp); c.checks = dev_upload(sc.checks, tmp); c.costs = dev_upload(sc.costs, tmp);
                c.n_bases = (uint32_t)sc.bases.size(); c.n_checks = (uint32_t)sc.checks.size();
                c.right_feats = dev_upload(is_dual ? dict_->dual.right_feats : dict_->raw.right_feats, tmp);
                c.left_feats = dev_upload(is_dual ? dict_->dual.left_feats : dict_->raw.left_feats, tmp);
                c.width = is_dual ? 8u : dict_->raw.width;
                if (is_dual) {
                    c.m = dev_upload(dict_->dual.matrix, tmp);
                    c.right_map = dev_upload(dict_->dual.right_map, tmp);
                    c.left_map = dev_upload(dict_->dual.left_map, tmp);
                    c.m_num_right = dict_->dual.m_num_right;
                }
                const dim3 grid((dict_->num_right + 255) / 256, dict_->num_left);
                int16_t* m = nullptr;
                HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m), (cells + 1) * 2));
                tmp.push_back(m);
                HIP_CHECK(hipMemset(m + cells, 0, 2));
                hipLaunchKernelGGL(expand_connector<int16_t>, grid, dim3(256), 0, nullptr, c, m, dict_->num_right, dict_->num_left, flag);
                uint32_t out_of_range = 0;
                HIP_CHECK(hipMemcpy(&out_of_range, flag, 4, hipMemcpyDeviceToHost));
                if (!out_of_range) {
                    tmp.pop_back();
                    allocs_.push_back(m);
                    dev_.matrix = m;
                } else {
                    if ((uint64_t)cells * 4 >= (1ull << 32))
                        throw Error(VBT_ERR_UNSUPPORTED, "connector: i32 connection matrices of 4 GiB and more are not supported by the device image");
                    (void)hipFree(m);
                    tmp.pop_back();
                    int32_t* w = nullptr;
                    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&w), (cells + 1) * 4));
                    tmp.push_back(w);
                    HIP_CHECK(hipMemset(w + cells, 0, 4));
                    hipLaunchKernelGGL(expand_connector<int32_t>, grid, dim3(256), 0, nullptr, c, w, dict_->num_right, dict_->num_left, flag);
                    HIP_CHECK(hipDeviceSynchronize());
                    tmp.pop_back();
                    allocs_.push_back(w);
                    dev_.matrix = reinterpret_cast<const int16_t*>(w);
                    dev_.matrix_wide = 1;
                    dev_.matrix_bytes = (uint32_t)(cells * 4);
                }
                for (void* p : tmp) (void)hipFree(p);
                tmp.clear();
            } catch (...) {
                for (void* p : tmp) (void)hipFree(p);
                throw;
            }
        }
        dev_.num_right = dict_->num_right;
        dev_.chr2inf = dev_upload(dict_->chr2inf, allocs_);
        dev_.unk_off = dev_upload(dict_->unk_offsets, allocs_);
        dev_.unk_entries = dev_upload(dict_->unk_entries, allocs_);
    } catch (...) {
        for (void* p : allocs_) (void)hipFree(p);
        throw;
    }
}

void Tokenizer::upload_lexicon(const Lexicon& lx, DevLexicon& out) {
    out.mapper = dev_upload(lx.mapper, allocs_);
    out.mapper_len = (uint32_t)lx.mapper.size();
    out.nodes = dev_upload(lx.nodes, allocs_);
    out.root_base = lx.nodes.empty() ? 0 : lx.nodes[0].base;
    out.entries = dev_upload(lx.entries, allocs_);
}

Tokenizer::~Tokenizer() {
    for (void* p : allocs_) (void)hipFree(p);
}

// ------------------------------------------------------------------ Workspace

Workspace::Workspace(const Tokenizer& t, uint64_t max_s, uint64_t max_b) : tok(t), max_sentences(max_s), max_bytes(max_b) {
    try {
    HIP_CHECK(hipSetDevice(tok.device()));
    if (max_b >= 0xFFFFFFFFull || max_s >= 0xFFFFFFFFull) throw Error(VBT_ERR_INVALID_ARGUMENT, "workspace: batch too large (split it)");
    fused = env_u32("VBT_FUSED", 0) != 0;
    // LDS tiers (bytes per wave), ascending; the global-memory tier always follows
    {
        const char* e = std::getenv("VBT_TIERS");
        std::string spec = e && *e ? e : (fused ? "16384,32768,65536" : env_u32("VBT_SEG_BYTES", 8192) ? "8192,49152,163840" : "8192,12288,16384,24576,32768,49152,65536,163840");
        size_t pos = 0;
        while (pos < spec.size()) {
            size_t c = spec.find(',', pos);
            if (c == std::string::npos) c = spec.size();
            uint32_t v = (uint32_t)std::strtoul(spec.substr(pos, c - pos).c_str(), nullptr, 10);
            if (v < 256 || v > 163840 || (fused && v > 65536) || (!tiers.empty() && v <= tiers.back()) || tiers.size() >= kMaxTiers)
                throw Error(VBT_ERR_INVALID_ARGUMENT, "VBT_TIERS: expected up to 8 ascending LDS sizes in [256, 163840]");
            tiers.push_back(v);
            pos = c + 1;
        }
    }
    const size_t ns = std::max<uint64_t>(max_s, 1), nbts = std::max<uint64_t>(max_b, 1);
    auto alloc = [&](size_t bytes) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 16)));
        pipe_allocs.push_back(p);
        return p;
    };
    d_tokens = static_cast<vbt_token_rec*>(alloc(nbts * sizeof(vbt_token_rec)));  // tokens <= chars <= bytes
    d_tok_stage = static_cast<vbt_token_rec*>(alloc((nbts + kSentenceSlack * ns + 1) * sizeof(vbt_token_rec)));  // per-sentence regions
    d_tile_sums = static_cast<uint32_t*>(alloc(((ns + kScanTile - 1) / kScanTile + 1) * 4));
    d_tok_off = static_cast<uint32_t*>(alloc(ns * 4));
    d_tok_cnt = static_cast<uint32_t*>(alloc(ns * 4));
    d_over = static_cast<uint32_t*>(alloc(2 * ns * 4 * (tiers.size() + kListsBehindTiers)));  // two regions per list: the launch stream's and (VBT_EARLY_LONG=1) the long sentences' side streams'
    d_ctrl = static_cast<uint32_t*>(alloc((kCtrlWords + (size_t)kBlockCtrlWords) * 4));  // one block: cleared by one memset per batch
    d_cctrl = d_ctrl + kCtrlWords;
    if (const char* e = std::getenv("VBT_TIER_WAVES")) {  // experiment: fixed lattice grid per tier
        std::string spec = e;
        size_t pos = 0;
        while (pos < spec.size()) {
            size_t c = spec.find(',', pos);
            if (c == std::string::npos) c = spec.size();
            tier_waves.push_back((uint32_t)std::strtoul(spec.substr(pos, c - pos).c_str(), nullptr, 10));
            pos = c + 1;
        }
    }
    d_prof = static_cast<unsigned long long*>(alloc(kProfSlots * kProfWords * 8));
    HIP_CHECK(hipMemset(d_prof, 0, kProfSlots * kProfWords * 8));
    const uint64_t mb = env_u32("VBT_SCRATCH_MB", 0);
    // (fused fallback only: rare sentences; a one-sentence Worker must not pin 256 MiB)
    scratch_bytes = mb ? mb << 20 : std::max<uint64_t>(std::min<uint64_t>(256ull << 20, (16ull << 20) + 512 * nbts), 64 * nbts);
    d_scratch = static_cast<char*>(alloc(scratch_bytes));
    profile = env_u32("VBT_PROFILE", 0) != 0;
    for (auto& e : ev) HIP_CHECK(hipEventCreate(reinterpret_cast<hipEvent_t*>(&e)));
    // two-kernel pipeline buffers
    const size_t slots = nbts + kSentenceSlack * ns + 1;
    pipe.s_n = static_cast<uint32_t*>(alloc(ns * 4));
    pipe.s_C = static_cast<uint32_t*>(alloc(ns * 4));
    if (!fused) {
        pipe.g_c2b = static_cast<uint16_t*>(alloc(slots * 2));
        pipe.g_pc = static_cast<uint4*>(alloc(slots * 16));
        pipe.node_factor = std::max<uint32_t>(1, env_u32("VBT_NODE_FACTOR", 8));  // candidate slots per input byte
        pipe.g_cand = static_cast<uint4*>(alloc((size_t)pipe.node_factor * slots * 16));
        pipe.g_hits = static_cast<uint4*>(alloc((size_t)pipe.node_factor * slots * 16));
        pipe.s_passes = static_cast<uint32_t*>(alloc(ns * 4));
        pipe.s_tier = static_cast<uint8_t*>(alloc(ns));
        for (size_t t = 0; t < tiers.size(); ++t) {
            hipStream_t st;
            HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            streams.push_back(st);
            hipEvent_t e;
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            tier_events.push_back(e);
        }
        HIP_CHECK(hipEventCreateWithFlags(reinterpret_cast<hipEvent_t*>(&ev_fork2), hipEventDisableTiming));

        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gen_candidates_large), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        if (tiers.back() > 65536)  // a single workgroup may use the CU's whole 160 KiB
        {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lattice_lds<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiers.back()));
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lattice_lds<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiers.back()));
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lattice_lds<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiers.back()));
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lattice_lds<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiers.back()));
        }
    }
    } catch (...) {  // a failed hipMalloc / stream / event must not leak what was created before it
        release();
        throw;
    }
}

void Workspace::release() {
    for (void* p : pipe_allocs) (void)hipFree(p);
    pipe_allocs.clear();
    for (auto& e : ev) if (e) { (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(e)); e = nullptr; }
    for (void* e : tier_events) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(e));
    tier_events.clear();
    if (ev_fork2) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(ev_fork2));
    ev_fork2 = nullptr;
    for (void* st : streams) (void)hipStreamDestroy(reinterpret_cast<hipStream_t>(st));
    streams.clear();
}

Workspace::~Workspace() { release(); }

void Workspace::run(const uint8_t* d_text, const uint64_t* d_offsets, uint64_t n, uint64_t total_bytes, void* stream_, bool defer_pack) {
    if (n > max_sentences || total_bytes > max_bytes) throw Error(VBT_ERR_INVALID_ARGUMENT, "batch exceeds the workspace capacity");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    HIP_CHECK(hipSetDevice(tok.device()));
    last_n = n;
    last_stream = stream_;
    HIP_CHECK(hipMemsetAsync(d_ctrl, 0, (kCtrlWords + (size_t)kBlockCtrlWords) * 4, stream));  // ctrl + cctrl
    if (n == 0) return;
    const size_t T = tiers.size();
    const size_t stride = 2 * std::max<uint64_t>(max_sentences, 1);
    BatchArgs a = pipe;
    a.text = d_text; a.offsets = d_offsets; a.n = (uint32_t)n;
    a.tokens = d_tokens; a.tok_stage = d_tok_stage; a.tok_cap = (uint32_t)std::max<uint64_t>(max_bytes, 1);
    a.tok_off = d_tok_off; a.tok_cnt = d_tok_cnt; a.ctrl = d_ctrl; a.tile_sums = d_tile_sums;
    a.scratch = d_scratch; a.scratch_bytes = scratch_bytes;
    a.prof = profile ? d_prof : nullptr;
    a.lists = d_over; a.list_stride = (uint32_t)stride; a.n_tiers = (uint32_t)T;
    a.tier_prio = env_u32("VBT_TIER_PRIO", 3);
    {   // tier whose waves sweep longer sentences segment by segment (VBT_SEG_BYTES=0: off, sentences use the big tiers)
        const uint32_t seg_bytes = env_u32("VBT_SEG_BYTES", 8192);
        a.seg_tier = 0xFFFFFFFFu;
        if (seg_bytes)
            for (size_t t = 0; t < T; ++t)
                if (tiers[t] >= seg_bytes) { a.seg_tier = (uint32_t)t; break; }
    }
    a.sid0 = 0; a.cctrl = d_cctrl; a.list_off = 0;
    a.lid_count = count_connids ? d_connid : nullptr;
    a.rid_count = count_connids ? d_connid + tok.dict().num_left : nullptr;
    a.s_counted = count_connids ? d_counted : nullptr;
    if (count_connids) HIP_CHECK(hipMemsetAsync(d_counted, 0, n * 4, stream));
    for (size_t t = 0; t < T; ++t) a.tier_bytes[t] = tiers[t];
    const DevDict& D = tok.dev();
    auto rec = [&](int i) { if (timing) HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(ev[i]), stream)); };
    auto over = [&](size_t t) { return d_over + t * stride; };
    auto waves_for = [&](uint32_t lds, uint64_t items) {
        const uint32_t per_cu = std::min<uint32_t>(32, std::max<uint32_t>(1, 163840 / lds));
        return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(items, (uint64_t)per_cu * 256));
    };
    rec(0);
    {   // device-side input contract (offsets, UTF-8); a rejected batch is skipped by every kernel below
        const uint64_t work = std::max<uint64_t>(n, total_bytes / 8 + 1);
        hipLaunchKernelGGL(validate_batch, dim3((uint32_t)std::min<uint64_t>((work + 255) / 256, 1u << 16)), dim3(256), 0, stream, a, total_bytes);
    }
    if (fused) {
        auto count = [&](size_t t) { return d_cctrl + 2 * t; };
        auto cursor = [&](size_t t) { return d_cctrl + 2 * t + 1; };
        auto fused_lds = D.matrix_wide ? tokenize_lds<true> : tokenize_lds<false>;
        hipLaunchKernelGGL(fused_lds, dim3((uint32_t)n), dim3(64), tiers[0], stream, D, a, tiers[0], (const uint32_t*)nullptr,
                           (const uint32_t*)nullptr, (uint32_t*)nullptr, over(0), count(0));
        rec(1);
        for (size_t t = 1; t < T; ++t)
            hipLaunchKernelGGL(fused_lds, dim3(waves_for(tiers[t], n)), dim3(64), tiers[t], stream, D, a, tiers[t],
                               (const uint32_t*)over(t - 1), (const uint32_t*)count(t - 1), cursor(t), over(t), count(t));
        hipLaunchKernelGGL(D.matrix_wide ? tokenize_global<true> : tokenize_global<false>, dim3((uint32_t)std::min<uint64_t>(n, 1024)), dim3(64), 0, stream, D, a,
                           (const uint32_t*)over(T - 1), (const uint32_t*)count(T - 1), cursor(T));
        rec(3);
    } else {
        // Stream plan.  The launch stream runs gen_candidates -> build_lists -> gen_candidates_large (gen_long: the sentences
        // that outgrew the bulk generator's LDS, one workgroup of several wavefronts each) and then forks one lattice_lds launch
        // per LDS tier onto the tier streams and joins them.
        // (the bulk generator keeps ~26 bytes of LDS per character: 4 KiB hold ~155 characters and 32+ waves per CU)
        const uint32_t gen_lds = env_u32("VBT_GEN_LDS", 4096);
        uint32_t gen_level_lds[kGenLevels] = {16384, 32768, 163840};  // the levels of gen_long (VBT_GEN_LEVELS=a,b,c): ~16 bytes per character
        if (const char* e = std::getenv("VBT_GEN_LEVELS")) {
            unsigned v[3];
            if (std::sscanf(e, "%u,%u,%u", &v[0], &v[1], &v[2]) == 3 && v[0] >= 4096 && v[0] < v[1] && v[1] < v[2] && v[2] <= 163840)
                for (int q = 0; q < 3; ++q) gen_level_lds[q] = v[q];
        }
        const uint32_t cn = (uint32_t)n, lb = (cn + 1023) / 1024;
        a.sid0 = 0; a.n = cn; a.cctrl = d_cctrl; a.list_off = 0; a.direct_push = 0;
        for (int q = 0; q < kGenLevels; ++q) a.gen_level_bytes[q] = gen_level_lds[q] - 16;
        const uint32_t persist = env_u32("VBT_LAT_PERSIST", 0);
        auto launch_lattice = [&](const BatchArgs& a_, dim3 grid_, uint32_t lds_, hipStream_t st_, uint32_t tier_, uint32_t persistent_) {
            auto k = D.space_cateset ? (D.matrix_wide ? lattice_lds<true, true> : lattice_lds<true, false>)
                                     : (D.matrix_wide ? lattice_lds<false, true> : lattice_lds<false, false>);
            hipLaunchKernelGGL(k, grid_, dim3(64), lds_, st_, D, a_, tier_, persistent_);
        };
        // (s_tier[] = 0xFF, "nothing routed yet", is written by validate_batch: one launch less per batch)
        hipLaunchKernelGGL(gen_candidates, dim3(cn), dim3(64), gen_lds, stream, D, a, gen_lds);
        hipLaunchKernelGGL(build_lists, dim3(lb), dim3(1024), 0, stream, a, -1);
        // gen_one files every sentence that outgrows it at the smallest level of gen_long that holds it; the levels run one after
        // the other on the launch stream: workgroups of 4 wavefronts (16 at the last level, which has a CU to itself), as many as
        // a CU's LDS and its 32 wave slots admit.  (Next to the bulk generator, on side streams, their 16-160 KiB workgroups do not
        // get onto a CU before the bulk launch has been dispatched: profiles/r03_long_first_experiment.md.)
        for (uint32_t lv = 1; lv <= (uint32_t)kGenLevels; ++lv) {
            const uint32_t lds = gen_level_lds[lv - 1];
            const uint32_t nw = lds > 65536 ? 16u : std::max<uint32_t>(1, std::min<uint32_t>(16, lv == 1 ? env_u32("VBT_GEN_WAVES1", env_u32("VBT_GEN_WAVES", 4)) : env_u32("VBT_GEN_WAVES", 4)));
            const uint32_t per_cu = std::max<uint32_t>(1, std::min<uint32_t>(32 / nw, 163840 / lds));
            hipLaunchKernelGGL(gen_candidates_large, dim3(std::max<uint32_t>(1, std::min<uint32_t>(cn, per_cu * 256))), dim3(nw * 64), lds, stream, D, a, lds, lv);
        }
        rec(1);
        HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_fork2), stream));
        // One lattice_lds launch per LDS tier up to the segment tier (default: a single tier).  The tiers above it are escape
        // tiers: nothing is routed to them up front, they take what the tier before them could not sweep (a window of end lists
        // wider than its LDS), so they are launched on the segment tier's stream, behind it.
        const size_t n_conc = a.seg_tier < T ? a.seg_tier + 1 : T;
        // The segment tier (the critical path: the longest sentences, then the escape tiers behind it) is launched on the launch
        // stream itself -- no event round trip before it starts nor before what follows it (VBT_MAIN_SEG=0: every tier on a side stream).
        const bool main_seg = env_u32("VBT_MAIN_SEG", 1) != 0 && a.seg_tier < T;
        for (size_t i = 0; i < n_conc; ++i) {
            const size_t t = n_conc - 1 - i;
            const bool on_main = main_seg && t == a.seg_tier;
            hipStream_t side = on_main ? stream : reinterpret_cast<hipStream_t>(streams[t]);
            if (!on_main) HIP_CHECK(hipStreamWaitEvent(side, reinterpret_cast<hipEvent_t>(ev_fork2), 0));
            // one workgroup per list entry: the lists are built on the device, so the grid covers the whole batch and the
            // workgroups beyond a list's length exit at once (VBT_LAT_PERSIST=1: persistent waves with a work cursor)
            const uint32_t grid = !persist ? cn : (t < tier_waves.size() && tier_waves[t]) ? std::min<uint32_t>(tier_waves[t], waves_for(tiers[t], cn)) : waves_for(tiers[t], cn);
            launch_lattice(a, dim3(grid), tiers[t], side, (uint32_t)t, persist);
            if (t == a.seg_tier)
                for (size_t x = t + 1; x < T; ++x)
                    launch_lattice(a, dim3(waves_for(tiers[x], std::min<uint32_t>(cn, 4096))), tiers[x], side, (uint32_t)x, 1u);
            if (!on_main) HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(tier_events[t]), side));
        }
        for (size_t t = 0; t < n_conc; ++t)
            if (!(main_seg && t == a.seg_tier)) HIP_CHECK(hipStreamWaitEvent(stream, reinterpret_cast<hipEvent_t>(tier_events[t]), 0));
        rec(3);
        // whatever the pipeline could not take: fused kernel, global-memory lattice (persistent waves with a work cursor; the list is
        // empty or a handful of sentences, and the kernel uses scratch memory: launching 1024 of them cost 15 us, 128 cost 6)
        hipLaunchKernelGGL(D.matrix_wide ? tokenize_global<true> : tokenize_global<false>, dim3((uint32_t)std::min<uint64_t>(cn, env_u32("VBT_FB_WGS", 128))), dim3(64), 0, stream, D, a,
                           (const uint32_t*)over(T), (const uint32_t*)(d_cctrl + 2 * T), d_cctrl + 2 * T + 1);
    }
    {   // pack the tokens in sentence order (tok_off, total)
        const uint32_t n_tiles = (uint32_t)((n + kScanTile - 1) / kScanTile);
        // (the totals per tile were added up by the kernels that emitted the tokens; their prefix is taken inside compact_tokens
        // unless the batch is huge or the caller packs later and wants the grand total first)
        const bool scan_kernel = defer_pack || n_tiles > 2048 || env_u32("VBT_PACK_SCAN", 0);
        if (scan_kernel) hipLaunchKernelGGL(tok_tile_scan, dim3(1), dim3(1024), 0, stream, a, d_tile_sums, n_tiles);
        if (!defer_pack) hipLaunchKernelGGL(compact_tokens, dim3(n_tiles * kPackSplit), dim3(kScanBlock), 0, stream, a, (const uint32_t*)d_tile_sums, n_tiles, scan_kernel ? 1u : 0u);
        last_args = a;
    }
    rec(2);
    HIP_CHECK(hipGetLastError());
}

void Workspace::run_one(const uint8_t* h_text_dev, uint32_t nb, uint8_t* d_text, uint64_t* d_offsets, vbt_token_rec* tokens_out, uint32_t* count_out,
                        uint32_t* status_out, void* stream_) {
    if (fused) throw Error(VBT_ERR_UNSUPPORTED, "the single-launch path needs the two-kernel pipeline (unset VBT_FUSED)");
    if (nb > max_bytes) throw Error(VBT_ERR_INVALID_ARGUMENT, "sentence exceeds the workspace capacity");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    last_n = 1;
    last_stream = stream_;
    BatchArgs a = pipe;
    a.text = d_text; a.offsets = d_offsets; a.n = 1;
    a.tokens = d_tokens; a.tok_stage = tokens_out; a.tok_cap = (uint32_t)std::max<uint64_t>(max_bytes, 1);
    a.tok_off = d_tok_off; a.tok_cnt = count_out; a.ctrl = d_ctrl; a.tile_sums = d_tile_sums;  // (tile sums: written, never read on this path)
    a.scratch = d_scratch; a.scratch_bytes = scratch_bytes;
    a.prof = nullptr;
    a.lists = d_over; a.list_stride = (uint32_t)(2 * std::max<uint64_t>(max_sentences, 1)); a.n_tiers = 1;
    a.tier_prio = 0; a.seg_tier = 0; a.sid0 = 0; a.cctrl = d_cctrl; a.list_off = 0; a.direct_push = 0;
    a.lid_count = nullptr; a.rid_count = nullptr; a.s_counted = nullptr;
    constexpr uint32_t kOneLds = 65536;  // generator arrays (~26 B per character), then the lattice (whole up to ~400 characters, in segments beyond)
    a.tier_bytes[0] = kOneLds;
    const DevDict& D = tok.dev();
    auto k = D.space_cateset ? (D.matrix_wide ? tokenize_one<true, true> : tokenize_one<true, false>)
                             : (D.matrix_wide ? tokenize_one<false, true> : tokenize_one<false, false>);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), kOneLds, stream, D, a, kOneLds, h_text_dev, nb, status_out);
    HIP_CHECK(hipGetLastError());
}

void Workspace::pack_to(vbt_token_rec* out_tokens, uint32_t* out_off, uint32_t* out_cnt, void* stream_) {
    if (last_n == 0) return;
    const uint32_t n_tiles = (uint32_t)((last_n + kScanTile - 1) / kScanTile);
    const uint32_t wgs = std::max<uint32_t>(1, std::min<uint32_t>(n_tiles, env_u32("VBT_PACK_WGS", n_tiles)));
    hipLaunchKernelGGL(compact_tokens_out, dim3(wgs), dim3(kScanBlock), 0, reinterpret_cast<hipStream_t>(stream_), last_args, (const uint32_t*)d_tile_sums, n_tiles,
                       out_tokens, out_off, out_cnt);
    HIP_CHECK(hipGetLastError());
}

void Workspace::stats(vbt_call_stats* out) {
    HIP_CHECK(hipSetDevice(tok.device()));
    HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(last_stream)));
    uint32_t ctrl[kCtrlWords];
    std::vector<uint32_t> cc((size_t)kBlockCtrlWords);
    HIP_CHECK(hipMemcpy(ctrl, d_ctrl, sizeof(ctrl), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(cc.data(), d_cctrl, cc.size() * 4, hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof(*out));
    const size_t T = tiers.size();
    out->n_sentences = last_n;
    if (fused) {
        out->n_tier0 = last_n - cc[0];
        out->n_tier2 = cc[2 * (T - 1)];
        out->n_tier1 = last_n - out->n_tier0 - out->n_tier2;
    } else {
        out->n_tier0 = cc[0];
        out->n_tier2 = cc[2 * T];
        for (size_t t = 1; t < T; ++t) out->n_tier1 += cc[2 * t];
    }
    out->n_tokens = ctrl[kTotal];
    out->error_flags = ctrl[kError];
    if (std::getenv("VBT_DEBUG")) {
        std::fprintf(stderr, "[vbt] lattice fallbacks: arena=%u window=%u passes=%u no-cut=%u space-tail=%u interface/backtrace=%u; lists:", ctrl[26], ctrl[27], ctrl[29], ctrl[30], ctrl[31], ctrl[28]);
        for (size_t t = 0; t < T + kListsBehindTiers; ++t) std::fprintf(stderr, " %u", cc[2 * t]);
        std::fprintf(stderr, "\n");
    }
    if (timing && last_n) {
        HIP_CHECK(hipEventElapsedTime(&out->ms_tier0, reinterpret_cast<hipEvent_t>(ev[0]), reinterpret_cast<hipEvent_t>(ev[1])));
        HIP_CHECK(hipEventElapsedTime(&out->ms_tier12, reinterpret_cast<hipEvent_t>(ev[1]), reinterpret_cast<hipEvent_t>(ev[3])));
        HIP_CHECK(hipEventElapsedTime(&out->ms_pack, reinterpret_cast<hipEvent_t>(ev[3]), reinterpret_cast<hipEvent_t>(ev[2])));
    }
}

void Workspace::enable_connid_counts(bool on) {
    HIP_CHECK(hipSetDevice(tok.device()));
    if (on && fused) throw Error(VBT_ERR_UNSUPPORTED, "connection-id counting needs the two-kernel pipeline (unset VBT_FUSED)");
    const size_t words = (size_t)tok.dict().num_left + tok.dict().num_right;
    if (on && !d_connid) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, std::max<size_t>(words * 8, 16)));
        pipe_allocs.push_back(p);
        d_connid = static_cast<unsigned long long*>(p);
        HIP_CHECK(hipMemset(d_connid, 0, words * 8));
        void* q = nullptr;
        HIP_CHECK(hipMalloc(&q, std::max<size_t>(max_sentences * 4, 16)));
        pipe_allocs.push_back(q);
        d_counted = static_cast<uint32_t*>(q);
    }
    count_connids = on;
}

void Workspace::read_connid_counts(uint64_t* lid, uint64_t* rid, bool reset) {
    HIP_CHECK(hipSetDevice(tok.device()));
    if (!d_connid) throw Error(VBT_ERR_INVALID_STATE, "connection-id counting was never enabled");
    HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(last_stream)));
    const size_t nl = tok.dict().num_left, nr = tok.dict().num_right;
    HIP_CHECK(hipMemcpy(lid, d_connid, nl * 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(rid, d_connid + nl, nr * 8, hipMemcpyDeviceToHost));
    if (reset) HIP_CHECK(hipMemset(d_connid, 0, (nl + nr) * 8));
}

void Workspace::reset_connid_counts() {
    HIP_CHECK(hipSetDevice(tok.device()));
    if (!d_connid) return;
    HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(last_stream)));
    HIP_CHECK(hipMemset(d_connid, 0, ((size_t)tok.dict().num_left + tok.dict().num_right) * 8));
}

void Workspace::read_profile(uint64_t* out, bool reset) {
    HIP_CHECK(hipSetDevice(tok.device()));
    HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(last_stream)));
    std::vector<uint64_t> h((size_t)kProfSlots * kProfWords);
    HIP_CHECK(hipMemcpy(h.data(), d_prof, h.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < kProfWords; ++i) out[i] = 0;
    for (int k = 0; k < kProfSlots; ++k)
        for (int i = 0; i < kProfWords; ++i) out[i] += h[(size_t)k * kProfWords + i];
    if (reset) HIP_CHECK(hipMemset(d_prof, 0, h.size() * 8));
}

}  // namespace vbt
