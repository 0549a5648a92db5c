// HIP kernels + launch sequence of the MI355X tokenizer (gfx950, wave64).
//
// One wavefront tokenizes one sentence end to end (fused): UTF-8 decode, char
// categories and groupable runs (Sentence::compile, sentence.rs:34-71), candidate
// generation by double-array common-prefix search + unknown-word rules
// (Tokenizer::add_lattice_edges tokenizer.rs:141-199, UnkHandler::gen_unk_words
// unknown.rs:69-137), the position sweep with per-node min-cost search over the
// connection matrix (build_lattice_inner tokenizer.rs:94-139, Lattice::insert_node /
// search_min_node lattice.rs:103-151, insert_eos 85-101) and the back-trace
// (append_top_nodes lattice.rs:159-168).  The whole lattice lives in LDS (tier 0/1)
// or, for sentences that do not fit, in a global scratch slab (tier 2).
//
// Bit-exactness notes: end lists are built with LDS atomics, so
// their order is arbitrary; every node carries its insertion sequence number and ties
// are broken towards the LARGEST sequence number, which is exactly what `<=` does in
// search_min_node (lattice.rs:141-146).  Candidates of positions the sweep never visits
// (unreachable or inside a skipped space run) keep cost = kInvalidCost and are ignored.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "engine.hpp"

namespace vbt {
namespace {

constexpr int32_t kInvalidCost = 0x7FFFFFFF;  // MAX_COST, lattice.rs:9
constexpr uint64_t kNoFit = ~0ull;

#define HIP_CHECK(expr)                                                                               \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            throw Error(VBT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

// ---------------------------------------------------------------- wave helpers

__device__ __forceinline__ uint32_t wave_exscan(uint32_t v, uint32_t& total) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d);
        if ((int)threadIdx.x >= d) x += y;
    }
    total = __shfl(x, 63);
    return x - v;
}

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t k) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)k, d), hi = __shfl_xor((uint32_t)(k >> 32), d);
        uint64_t o = ((uint64_t)hi << 32) | lo;
        k = o < k ? o : k;
    }
    return k;
}

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
        const uint4 nd = *reinterpret_cast<const uint4*>(&L.nodes[child]);
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
template <typename IdxT, bool kGlobal>
__device__ __forceinline__ uint64_t process_sentence(const DevDict& D, const BatchArgs& A, uint32_t sid, char* abase,
                                                     uint64_t acap) {
    const uint32_t ln = threadIdx.x;
    const uint64_t lt_mask = (1ull << ln) - 1ull;
    constexpr uint64_t kIdxMax = (uint64_t)(IdxT) ~(IdxT)0;
    const uint64_t b0 = A.offsets[sid], nb64 = A.offsets[sid + 1] - b0;
    if (nb64 == 0) {
        if (ln == 0) { A.tok_off[sid] = 0; A.tok_cnt[sid] = 0; }
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
        if (ln == 0) { A.tok_off[sid] = 0; A.tok_cnt[sid] = 0; }
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
            carry = __shfl(g, 0);
        }
    }
    __syncthreads();

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
    if ((uint64_t)C + 2 >= kIdxMax) return kNoFit;
    if (ln == 0) cand_off[n] = (IdxT)C;

    // start-major node arrays
    uint32_t* nd_word = ar.take<uint32_t>(C);
    int32_t* e_mc = ar.take<int32_t>(C + 1);  // end-major: min cost (kInvalidCost = never inserted)
    uint16_t* nd_left = ar.take<uint16_t>(C);
    int16_t* nd_wcost = ar.take<int16_t>(C);
    uint16_t* e_right = ar.take<uint16_t>(C + 1);
    IdxT* nd_end = ar.take<IdxT>(C);
    IdxT* nd_eslot = ar.take<IdxT>(C);
    IdxT* e_seq = ar.take<IdxT>(C + 1);
    IdxT* e_back = ar.take<IdxT>(C + 1);
    if (!ar.ok) return ar.used;
    uint16_t* tmp_right = reinterpret_cast<uint16_t*>(e_back);  // right ids until the end lists exist

    // zero the end counters while the fill pass runs
    for (uint32_t p = ln; p < n + 2; p += 64) end_off[p] = 0;
    for (uint32_t p = ln; p < n + 1; p += 64) reach[p] = 0;
    __syncthreads();

    // ---- P1b: fill candidates in reference insertion order (tokenizer.rs:155-198) --------
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        const uint32_t i = c0 + ln;
        if (i < n) {
            uint32_t k = cand_off[i];
            bool matched = false;
            auto put = [&](const Entry* ent, uint32_t v, uint32_t c, uint32_t end, uint32_t lex) {
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
        }
    }
    __syncthreads();

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
        e_seq[es] = (IdxT)c;
        e_mc[es] = kInvalidCost;
    }
    __syncthreads();  // all tmp_right reads done before e_back is written
    if (ln == 0) {
        e_right[0] = 0;  // BOS: right_id = BOS_EOS_CONNECTION_ID, min_cost = 0 (lattice.rs:72-83)
        e_seq[0] = 0;
        e_mc[0] = 0;
        e_back[0] = 0;
        reach[0] = 1;
    }
    __syncthreads();

    // ---- P4: position sweep (build_lattice_inner tokenizer.rs:106-138) ------------------------
    const int16_t* __restrict__ matrix = D.matrix;
    const uint32_t NR = D.num_right;
    uint32_t sn = 0, sw = 0;
    while (sw < n) {
        if (!__builtin_amdgcn_readfirstlane(reach[sn])) {  // has_previous_node, lattice.rs:155-157
            sw += 1;
            sn = sw;
            continue;
        }
        if (D.space_cateset) {  // tokenizer.rs:117-125
            const uint32_t cs = __builtin_amdgcn_readfirstlane(ci[sn]);
            if (cs & D.space_cateset) sw += __builtin_amdgcn_readfirstlane((uint32_t)grp[sn]);
        }
        if (sw == n) break;  // input ends with spaces, tokenizer.rs:128-130
        const uint32_t c_beg = __builtin_amdgcn_readfirstlane((uint32_t)cand_off[sw]);
        const uint32_t c_end = __builtin_amdgcn_readfirstlane((uint32_t)cand_off[sw + 1]);
        const uint32_t p_beg = __builtin_amdgcn_readfirstlane(end_off[sn]);
        const uint32_t p_end = __builtin_amdgcn_readfirstlane(end_off[sn + 1]);
        for (uint32_t cb = c_beg; cb < c_end; cb += 64) {
            const uint32_t c = cb + ln;
            if (c < c_end) {
                // search_min_node(start_node, left_id), lattice.rs:129-151
                const int16_t* row = matrix + (size_t)nd_left[c] * NR;  // matrix_connector.rs:79-85
                // Branch-free argmin over a packed key: (cost biased to unsigned) << 32 | ~seq, so the
                // minimum key = minimum cost, ties -> largest insertion sequence number (`<=`, l.143).
                uint64_t best = ~0ull;
                for (uint32_t j = p_beg; j < p_end; j += 8) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const bool in = j + k < p_end;
                        const uint32_t jj = in ? j + k : p_end - 1;
                        const int32_t mc = e_mc[jj];
                        const int32_t conn = row[e_right[jj]];
                        const uint32_t v = (uint32_t)mc + (uint32_t)conn;  // wrapping add, as in release builds
                        uint64_t key = ((uint64_t)(v ^ 0x80000000u) << 32) | (uint32_t)(~(uint32_t)e_seq[jj]);
                        key = (in && mc != kInvalidCost) ? key : ~0ull;
                        best = key < best ? key : best;
                    }
                }
                const uint32_t bseq = ~(uint32_t)best;
                const uint32_t bj = sn == 0 ? 0u : (uint32_t)nd_eslot[bseq];  // ends[0] holds only BOS (slot 0)
                const uint32_t bcost = (uint32_t)(best >> 32) ^ 0x80000000u;
                const uint32_t es = nd_eslot[c];
                e_mc[es] = (int32_t)(bcost + (uint32_t)(int32_t)nd_wcost[c]);  // lattice.rs:125
                e_back[es] = (IdxT)bj;
                reach[nd_end[c]] = 1;
            }
        }
        __syncthreads();
        sw += 1;
        sn = sw;
    }

    // ---- EOS (insert_eos lattice.rs:85-101): left_id = 0 --------------------------------------
    uint32_t eos_pred;
    {
        const uint32_t p_beg = end_off[sn], p_end = end_off[sn + 1];
        uint64_t key = ~0ull;
        for (uint32_t j = p_beg + ln; j < p_end; j += 64) {
            const int32_t mc = e_mc[j];
            if (mc != kInvalidCost) {
                const int32_t v = (int32_t)((uint32_t)mc + (uint32_t)(int32_t)matrix[e_right[j]]);
                // min cost, ties -> largest sequence number; low word also identifies the slot
                const uint64_t k2 = ((uint64_t)((uint32_t)v ^ 0x80000000u) << 32) | (uint32_t)(~(uint32_t)e_seq[j]);
                key = k2 < key ? k2 : key;
            }
        }
        key = wave_min_u64(key);
        const uint32_t seq = ~(uint32_t)key;
        eos_pred = sn == 0 ? 0u : (uint32_t)nd_eslot[seq];
    }

    // ---- P5: back-trace (append_top_nodes lattice.rs:159-168) + token records ------------------
    IdxT* path = grp;  // groupable is dead after the sweep; tokens <= chars
    uint32_t T = 0;
    if (ln == 0) {
        uint32_t cur = eos_pred;
        while (cur != 0 && T < n) {  // tokens <= chars; the bound also keeps a corrupted chain finite
            path[T++] = (IdxT)cur;
            cur = e_back[cur];
        }
    }
    T = __shfl(T, 0);
    uint32_t out_base = 0;
    if (ln == 0) out_base = atomicAdd(&A.ctrl[kTotal], T);
    out_base = __shfl(out_base, 0);
    __syncthreads();
    if ((uint64_t)out_base + T > A.tok_cap) {
        if (ln == 0) { atomicOr(&A.ctrl[kError], (uint32_t)kErrTokCap); A.tok_off[sid] = 0; A.tok_cnt[sid] = 0; }
        return 0;
    }
    if (ln == 0) { A.tok_off[sid] = out_base; A.tok_cnt[sid] = T; }
    for (uint32_t t = ln; t < T; t += 64) {
        const uint32_t es = path[T - 1 - t];  // Worker::token: index = n-1-i (worker.rs:65-68)
        const uint32_t c = e_seq[es];
        // start_word = the position whose candidate range contains c: upper_bound(cand_off, c) - 1
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint32_t)cand_off[mid + 1] <= c) lo = mid + 1; else hi = mid;
        }
        const uint32_t st = lo, en = nd_end[c];
        vbt_token_rec r;
        r.start_char = st; r.end_char = en;
        r.start_byte = c2b[st]; r.end_byte = c2b[en];
        r.word_idx = nd_word[c];
        r.total_cost = e_mc[es];
        A.tokens[out_base + t] = r;
    }
    return 0;
}

extern __shared__ __attribute__((aligned(16))) char g_smem[];

__device__ __forceinline__ void push_overflow(uint32_t* list, uint32_t* counter, uint32_t sid) {
    if (threadIdx.x == 0) list[atomicAdd(counter, 1u)] = sid;
}

// Tier 0: one single-wave workgroup per sentence, lattice in `lds_bytes` of LDS.
__global__ void __launch_bounds__(64) tokenize_tier0(DevDict D, BatchArgs A, uint32_t lds_bytes) {
    const uint32_t sid = blockIdx.x;
    if (process_sentence<uint16_t, false>(D, A, sid, g_smem, lds_bytes) != 0) push_overflow(A.overflow0, &A.ctrl[kOver0], sid);
}

// Tier 1: persistent waves with a large LDS budget drain the tier-0 overflow list.
__global__ void __launch_bounds__(64) tokenize_tier1(DevDict D, BatchArgs A, uint32_t lds_bytes) {
    const uint32_t count = A.ctrl[kOver0];
    for (;;) {
        uint32_t k = 0;
        if (threadIdx.x == 0) k = atomicAdd(&A.ctrl[kCursor1], 1u);
        k = __shfl(k, 0);
        if (k >= count) break;
        const uint32_t sid = A.overflow0[k];
        if (process_sentence<uint16_t, false>(D, A, sid, g_smem, lds_bytes) != 0) push_overflow(A.overflow1, &A.ctrl[kOver1], sid);
        __syncthreads();
    }
}

// Tier 2: persistent waves, lattice in a private global-memory slab (any sentence length).
__global__ void __launch_bounds__(64) tokenize_tier2(DevDict D, BatchArgs A) {
    const uint32_t count = A.ctrl[kOver1];
    char* slab = nullptr;
    uint64_t slab_bytes = 0;
    unsigned long long* bump = reinterpret_cast<unsigned long long*>(&A.ctrl[kBump]);
    for (;;) {
        uint32_t k = 0;
        if (threadIdx.x == 0) k = atomicAdd(&A.ctrl[kCursor2], 1u);
        k = __shfl(k, 0);
        if (k >= count) break;
        const uint32_t sid = A.overflow1[k];
        for (int attempt = 0; attempt < 4; ++attempt) {
            const uint64_t need = process_sentence<uint32_t, true>(D, A, sid, slab, slab_bytes);
            if (need == 0) break;
            bool failed = need == kNoFit || attempt == 3;
            if (!failed) {  // grow: take a fresh slab from the bump arena
                uint64_t want = need + need / 4 + 4096;
                want = (want + 255) & ~255ull;
                unsigned long long off = 0;
                if (threadIdx.x == 0) off = atomicAdd(bump, (unsigned long long)want);
                off = ((unsigned long long)__shfl((uint32_t)(off >> 32), 0) << 32) | __shfl((uint32_t)off, 0);
                if (off + want > A.scratch_bytes) failed = true;
                else { slab = A.scratch + off; slab_bytes = want; }
            }
            if (failed) {
                if (threadIdx.x == 0) {
                    atomicOr(&A.ctrl[kError], need == kNoFit ? (uint32_t)kErrTooLong : (uint32_t)kErrScratch);
                    A.tok_off[sid] = 0;
                    A.tok_cnt[sid] = 0;
                }
                break;
            }
            __syncthreads();
        }
        __syncthreads();
    }
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
        dev_.matrix = dev_upload(dict_->matrix, allocs_);
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
    HIP_CHECK(hipSetDevice(tok.device()));
    if (max_b >= 0xFFFFFFFFull || max_s >= 0xFFFFFFFFull) throw Error(VBT_ERR_INVALID_ARGUMENT, "workspace: batch too large (split it)");
    const size_t ns = std::max<uint64_t>(max_s, 1), nbts = std::max<uint64_t>(max_b, 1);
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_tokens), nbts * sizeof(vbt_token_rec)));  // tokens <= chars <= bytes
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_tok_off), ns * 4));
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_tok_cnt), ns * 4));
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_over0), ns * 4));
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_over1), ns * 4));
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_ctrl), kCtrlWords * 4));
    const uint64_t mb = env_u32("VBT_SCRATCH_MB", 0);
    scratch_bytes = mb ? mb << 20 : std::max<uint64_t>(256ull << 20, 128 * nbts);
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_scratch), scratch_bytes));
    lds0 = env_u32("VBT_LDS0", 12288);
    lds1 = env_u32("VBT_LDS1", 65536);
    if (lds0 > 65536 || lds1 > 65536) throw Error(VBT_ERR_INVALID_ARGUMENT, "VBT_LDS0/VBT_LDS1 must be <= 65536");
    for (auto& e : ev) HIP_CHECK(hipEventCreate(reinterpret_cast<hipEvent_t*>(&e)));
}

Workspace::~Workspace() {
    (void)hipFree(d_tokens); (void)hipFree(d_tok_off); (void)hipFree(d_tok_cnt); (void)hipFree(d_over0);
    (void)hipFree(d_over1); (void)hipFree(d_ctrl); (void)hipFree(d_scratch);
    for (auto& e : ev) if (e) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(e));
}

void Workspace::run(const uint8_t* d_text, const uint64_t* d_offsets, uint64_t n, uint64_t total_bytes, void* stream_) {
    if (n > max_sentences || total_bytes > max_bytes) throw Error(VBT_ERR_INVALID_ARGUMENT, "batch exceeds the workspace capacity");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    HIP_CHECK(hipSetDevice(tok.device()));
    last_n = n;
    last_stream = stream_;
    HIP_CHECK(hipMemsetAsync(d_ctrl, 0, kCtrlWords * 4, stream));
    if (n == 0) return;
    BatchArgs a;
    a.text = d_text; a.offsets = d_offsets; a.n = (uint32_t)n;
    a.tokens = d_tokens; a.tok_cap = (uint32_t)std::max<uint64_t>(max_bytes, 1);
    a.tok_off = d_tok_off; a.tok_cnt = d_tok_cnt; a.ctrl = d_ctrl; a.overflow0 = d_over0; a.overflow1 = d_over1;
    a.scratch = d_scratch; a.scratch_bytes = scratch_bytes;
    const DevDict& D = tok.dev();
    auto rec = [&](int i) { if (timing) HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(ev[i]), stream)); };
    rec(0);
    hipLaunchKernelGGL(tokenize_tier0, dim3((uint32_t)n), dim3(64), lds0, stream, D, a, lds0);
    rec(1);
    const uint32_t g1 = (uint32_t)std::min<uint64_t>(n, 512), g2 = (uint32_t)std::min<uint64_t>(n, 512);
    hipLaunchKernelGGL(tokenize_tier1, dim3(g1), dim3(64), lds1, stream, D, a, lds1);
    hipLaunchKernelGGL(tokenize_tier2, dim3(g2), dim3(64), 0, stream, D, a);
    rec(2);
    HIP_CHECK(hipGetLastError());
}

void Workspace::stats(vbt_call_stats* out) {
    HIP_CHECK(hipSetDevice(tok.device()));
    HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(last_stream)));
    uint32_t ctrl[kCtrlWords];
    HIP_CHECK(hipMemcpy(ctrl, d_ctrl, sizeof(ctrl), hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof(*out));
    out->n_sentences = last_n;
    out->n_tier0 = last_n - ctrl[kOver0];
    out->n_tier1 = ctrl[kOver0] - ctrl[kOver1];
    out->n_tier2 = ctrl[kOver1];
    out->n_tokens = ctrl[kTotal];
    out->error_flags = ctrl[kError];
    if (timing && last_n) {
        HIP_CHECK(hipEventElapsedTime(&out->ms_tier0, reinterpret_cast<hipEvent_t>(ev[0]), reinterpret_cast<hipEvent_t>(ev[1])));
        HIP_CHECK(hipEventElapsedTime(&out->ms_tier12, reinterpret_cast<hipEvent_t>(ev[1]), reinterpret_cast<hipEvent_t>(ev[2])));
    }
}

}  // namespace vbt
