// Candidate generation: the device-side input contract, the bulk generator (one wavefront per sentence:
// gen_one in gen_device.hpp), gen_long (one workgroup per sentence that outgrew it) and the work lists.
// Reference: Sentence::compile sentence.rs:34-71, Tokenizer::add_lattice_edges tokenizer.rs:141-199, UnkHandler::gen_unk_words
// unknown.rs:69-137, Lexicon::common_prefix_iterator lexicon.rs:33-46.
#include "gen_device.hpp"

namespace vbt {
namespace {

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
    if (tid == 0) A.s_hdr[sid] = make_uint4(0u, 0xFFu << 16, 0u, 0u);
#if VBT_GENLONG_PROF  // developer aid (tools/dbg/genlong_profile.py): wall cycles of the workgroup between its barriers
    uint64_t gl_t = clock64(), gl_acc[10] = {};
#define GL_MARK(i) do { const uint64_t t_ = clock64(); gl_acc[i] += t_ - gl_t; gl_t = t_; } while (0)
#else
#define GL_MARK(i) do { } while (0)
#endif
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
    GL_MARK(0);
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
    // what is left of the workgroup's LDS stages the trie hits (8 bytes each: pack_hit; see gen_one), the overflow goes to global memory
    uint2* const lhits = ar.take<uint2>(0);
    const uint32_t lcap = VBT_LDS_HITS && lds_bytes > ar.used ? (uint32_t)((lds_bytes - ar.used) / sizeof(uint2)) : 0u;
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
                const uint2 e = D.cpinfo[cp < D.cpinfo_len ? cp : D.cpinfo_len];  // {character class (character.rs:112-116), trie codes}: one load
                ci[idx] = e.x;
                code[idx] = (uint16_t)(e.y & 0xFFFFu);
                if (D.has_user) ucode[idx] = (uint16_t)(e.y >> 16);
                c2b[idx] = (uint16_t)bi;
            }
        }
        if (tid == 0) c2b[n] = (uint16_t)nb;
    }
    __syncthreads();
    GL_MARK(1);
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
    GL_MARK(2);

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
                const uint32_t len = end - i;
                if (h < lcap && hit_packs(v, c, i, len, cnt)) lhits[h] = pack_hit(v, c, lex, i, len, cnt);
                else {
                    if (h < lcap) lhits[h] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                    if (h < region) hits[h] = make_uint4(v, c | (lex << 16), end | (i << 16), cnt);
                }
                cnt += c;
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
    GL_MARK(3);
    // candidates before a position (CSR offsets) and the end-list offsets: two independent prefix sums, one wave each
    if (wv == 0) {
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
    }
    if (wv == (nw > 1 ? 1u : 0u)) {
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
    GL_MARK(4);

    // expand the hits: threads = hits (see gen_one)
    // (the next round's hit records are requested before this round's entries: one round trip per round instead of two; in
    // gen_one, with three rounds per sentence, the same was measured slightly slower)
    auto hit_at = [&](uint32_t h) {
        const uint2 q = h < lcap ? lhits[h] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        return q.y != 0xFFFFFFFFu ? unpack_hit(q) : hits[h];
    };
    uint4 hr_next = tid < H ? hit_at(tid) : make_uint4(0, 0, 0, 0);
    const uint32_t row_cells = D.num_right;  // (see gen_one)
    for (uint32_t h = tid; h < H; h += nthreads) {
        const uint4 hr = hr_next;
        if (h + nthreads < H) hr_next = hit_at(h + nthreads);
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
    GL_MARK(5);
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
    GL_MARK(6);
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
    GL_MARK(7);
    uint32_t passes = __builtin_amdgcn_readfirstlane(red[kPasses]), maxcnt = __builtin_amdgcn_readfirstlane(red[kMaxCnt]);
    {   // EOS connects to the end list of the last visited position: bounded by the longest list
        const uint32_t last = eo(n + 1) - eo(n);
        maxcnt = last > maxcnt ? last : maxcnt;
        passes += step_passes(1u, maxcnt);
    }
    if (tid == 0) A.g_pc[slot0 + n] = make_uint4(C | (eo(n) << 16), 0, eo(n + 1), 0);  // terminator: totals (candidates, end-list slots)
    // smallest tier whose LDS holds the lattice arrays, else the segment tier (see gen_one)
    const uint64_t fixed = lattice_fixed_bytes(C, n, eo(n + 1), passes);
    uint32_t tier = fallback;
    for (uint32_t t = 0; t < A.n_tiers; ++t)
        if (t >= A.n_lean && fixed <= A.tier_bytes[t]) { tier = t; break; }
    if (A.seg_tier < A.n_tiers && tier > A.seg_tier) tier = A.seg_tier;
    if (tid == 0) A.s_hdr[sid] = make_uint4(n | (nb << 16), C | (tier << 16), passes, (uint32_t)(b0 - uniform64(A.offsets[0])));
    route(tier);
#if VBT_GENLONG_PROF
    GL_MARK(8);
    if (A.prof && tid == 0) {
        unsigned long long* pr_ = A.prof + ((size_t)kProfSlots + (sid & (kProfSlots - 1))) * kProfWords;  // the second half of the buffer
        for (int i = 0; i < 9; ++i) atomicAdd(&pr_[i], (unsigned long long)gl_acc[i]);
        atomicAdd(&pr_[9], 1ull); atomicAdd(&pr_[10], (unsigned long long)n); atomicAdd(&pr_[11], (unsigned long long)H);
    }
#endif
#undef GL_MARK
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
    if (A.density_out && only_list < 0) {
        // lattice density of the batch, for the tokenizer's choice of tiers for the batches behind this one (Tokenizer::candidates_per_byte):
        // candidates and bytes of the sentences the bulk generator took (their headers are final), one pair of atomics per workgroup; the
        // last workgroup stores the sums into the tokenizer's pinned slot
        __shared__ uint32_t d_c[16], d_b[16];
        uint32_t c = 0, b = 0;
        if (rel < A.n) {
            const uint4 hq = A.s_hdr[sid];
            if (((hq.y >> 16) & 0xFFu) != 0xFFu) { c = hq.y & 0xFFFFu; b = hq.x >> 16; }
        }
        c = wave_sum(c); b = wave_sum(b);
        if (lane == 0) { d_c[wave] = c; d_b[wave] = b; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long cs = 0, bs = 0;
            for (uint32_t w = 0; w < 16; ++w) { cs += d_c[w]; bs += d_b[w]; }
            unsigned long long* acc_c = reinterpret_cast<unsigned long long*>(&A.ctrl[kDensCand]);
            unsigned long long* acc_b = reinterpret_cast<unsigned long long*>(&A.ctrl[kDensBytes]);
            atomicAdd(acc_c, cs); atomicAdd(acc_b, bs);
            __threadfence();
            if (atomicAdd(&A.ctrl[kDensDone], 1u) == gridDim.x - 1) {
                const unsigned long long tc = atomicAdd(acc_c, 0ull), tb = atomicAdd(acc_b, 0ull);
                if (tb) {
                    __hip_atomic_store(&A.density_out[0], tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(&A.density_out[1], tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        __syncthreads();
    }
    {   // sentences the generator's own wave swept (gen_sweep) are in no list: counted in the first tier's cursor word (unused: lean tiers have no cursor)
        const uint64_t m = __ballot(t == kRouteInline && only_list < 0);
        if (lane == 0 && m) atomicAdd(&A.cctrl[1], (uint32_t)__popcll(m));
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

}  // namespace

namespace kern {

void validate_batch(uint32_t blocks, hipStream_t stream, const BatchArgs& a, uint64_t total_bytes) {
    hipLaunchKernelGGL(vbt::validate_batch, dim3(blocks), dim3(256), 0, stream, a, total_bytes);
}
void gen_candidates(uint32_t n, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a) {
    hipLaunchKernelGGL(vbt::gen_candidates, dim3(n), dim3(64), lds_bytes, stream, D, a, lds_bytes);
}
void build_lists(uint32_t blocks, hipStream_t stream, const BatchArgs& a, int only_list) {
    hipLaunchKernelGGL(vbt::build_lists, dim3(blocks), dim3(1024), 0, stream, a, only_list);
}
void gen_candidates_large(uint32_t workgroups, uint32_t waves, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a, uint32_t level) {
    hipLaunchKernelGGL(vbt::gen_candidates_large, dim3(workgroups), dim3(waves * 64), lds_bytes, stream, D, a, lds_bytes, level);
}
void gen_set_max_lds(int bytes) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(vbt::gen_candidates_large), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
}

}  // namespace kern
}  // namespace vbt
