// gen_one: Sentence::compile + candidate enumeration of ONE sentence by ONE wavefront (the body of the bulk generator kernel in
// gen.hip and of the resident Worker kernel in lattice.hip).
#pragma once
#include "device_common.hpp"

namespace vbt {
namespace {

// Kernel 1 body: Sentence::compile + candidate enumeration of one sentence by one wavefront.
// Per-character working arrays live in LDS (the vector L1 stalls on hit-under-miss, so nothing
// is re-read from global while in flight); outputs: per-char records, byte offsets and the
// candidates in reference insertion order, each tagged with its (start position, left_id) group:
// search_min_node's result depends only on that pair (lattice.rs:129-151), so the lattice kernel
// evaluates one row per group instead of one per candidate.
// (Sentences that outgrow this wavefront's LDS -- ~26 bytes per character -- are handed to gen_long: one workgroup per sentence.)
// Returns the sentence's header as written to s_hdr (tier 0xFF: nothing to sweep here -- empty, or filed for gen_long / the fallback).
__device__ __forceinline__ uint4 gen_one(const DevDict& D, const BatchArgs& A, uint32_t sid, uint32_t lds_bytes) {
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
    // (header with tier 0xFF: nothing for lattice_lds -- the final header is written with the routing decision, by gen_long for what is filed there)
    auto init = [&]() { if (ln == 0) { A.s_hdr[sid] = make_uint4(0u, 0xFFu << 16, 0u, 0u); A.s_tier[sid] = 0xFF; } };
    const uint4 none = make_uint4(0u, 0xFFu << 16, 0u, 0u);
    if (nb64 == 0) {
        init();
        if (ln == 0) A.tok_cnt[sid] = 0;
        return none;
    }
    if (nb64 >= 65535) { init(); route(fallback); return none; }  // positions are u16 in the LDS lattice
    const uint32_t nb = (uint32_t)nb64;
    const uint8_t* __restrict__ txt = A.text + b0;
    const size_t slot0 = sentence_slot(A, b0, sid);

    // The text comes in as ONE unaligned 32-bit load per lane and 64-byte chunk -- byte bi and the three behind it, what a lead byte
    // needs to decode its character -- four chunks (256 bytes: 85 % of the headline's sentences whole) requested together and, for a
    // sentence of at most 256 bytes, kept in registers from the character count to the decode.  (Round 5 loaded single bytes chunk by
    // chunk, once to count the characters and once more to decode them, fetched the continuation bytes from the neighbouring lanes
    // through the LDS crossbar -- six ds_bpermute per chunk -- and waited for every chunk's chr2inf / mapper gathers before it looked at
    // the next chunk: 6-7 dependent round trips for the mean sentence where there are two, text and gathers.)
    constexpr uint32_t kGroup = 4;  // chunks per group
    auto bytes4 = [&](uint32_t bi) -> uint32_t {  // bytes bi .. bi + 3 of the sentence, 0 behind its end; 0x80 (no lead byte) for bi >= nb
        if (bi >= nb) return 0x80u;
        if (nb >= 4) {
            const uint32_t a = bi + 4 <= nb ? bi : nb - 4;  // never reads outside [0, nb)
            uint32_t v;
            __builtin_memcpy(&v, txt + a, 4);
            return v >> (8u * (bi - a));
        }
        uint32_t v = txt[bi];
        if (bi + 1 < nb) v |= (uint32_t)txt[bi + 1] << 8;
        if (bi + 2 < nb) v |= (uint32_t)txt[bi + 2] << 16;
        return v;
    };
    uint32_t w0[kGroup];
    const bool one_group = nb <= 64 * kGroup;
    uint32_t n = 0;
    if (one_group) {
#pragma unroll
        for (uint32_t q = 0; q < kGroup; ++q) w0[q] = bytes4(q * 64 + ln);
#pragma unroll
        for (uint32_t q = 0; q < kGroup; ++q) n += (uint32_t)__popcll(__ballot((w0[q] & 0xC0u) != 0x80u));
    } else n = count_chars(txt, nb);
    if (n == 0) {
        init();
        if (ln == 0) A.tok_cnt[sid] = 0;
        return none;
    }
    const GenOneLds L = carve_gen_one(g_smem, lds_bytes, n, D.has_user != 0);
    if (!L.ok) {
        // Outgrows this wavefront: gen_long, one workgroup per sentence.
        init();
        route(A.n_tiers + 1 + gen_long_level(A, n, nb, D.has_user != 0));
        return none;
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

    // decode (Sentence::compute_basic / compute_categories, sentence.rs:40-55): per group of four chunks the characters, then ALL their
    // chr2inf / mapper gathers, then the stores
    {
        uint16_t* c2b = A.g_c2b + slot0;
        uint32_t cb = 0;
        for (uint32_t g0 = 0; g0 < nb; g0 += 64 * kGroup) {
            uint32_t w[kGroup], cp[kGroup], idx[kGroup];
#pragma unroll
            for (uint32_t q = 0; q < kGroup; ++q) w[q] = one_group ? w0[q] : bytes4(g0 + q * 64 + ln);
#pragma unroll
            for (uint32_t q = 0; q < kGroup; ++q) {
                const uint32_t b = w[q] & 0xFFu, t0 = (w[q] >> 8) & 0x3Fu, t1 = (w[q] >> 16) & 0x3Fu, t2 = (w[q] >> 24) & 0x3Fu;
                const bool lead = (b & 0xC0u) != 0x80u;
                const uint64_t m = __ballot(lead);
                idx[q] = lead ? cb + (uint32_t)__popcll(m & lt_mask) : 0xFFFFFFFFu;
                cp[q] = b < 0x80 ? b : b < 0xE0 ? ((b & 0x1F) << 6) | t0 : b < 0xF0 ? ((b & 0x0F) << 12) | (t0 << 6) | t1 : ((b & 0x07) << 18) | (t0 << 12) | (t1 << 6) | t2;
                cb += (uint32_t)__popcll(m);
            }
            uint2 cpv[kGroup];  // {character class, system code | user code << 16}: one load per character (DevDict::cpinfo)
#pragma unroll
            for (uint32_t q = 0; q < kGroup; ++q) {
                const bool lead = idx[q] != 0xFFFFFFFFu;
                const uint32_t c = lead ? cp[q] : 0u;
#if VBT_CPINFO
                cpv[q] = D.cpinfo[c < D.cpinfo_len ? c : D.cpinfo_len];  // character.rs:112-116 (code points beyond the table: entry 0's class, no code)
#else  // (A/B: round 5's two or three gathers per character)
                cpv[q].x = D.chr2inf[c < 65536u ? c : 0u];
                cpv[q].y = (c < D.sys.mapper_len ? D.sys.mapper[c] : 0u) | ((D.has_user && c < D.user.mapper_len ? D.user.mapper[c] : 0u) << 16);
#endif
            }
#pragma unroll
            for (uint32_t q = 0; q < kGroup; ++q) {
                if (idx[q] != 0xFFFFFFFFu) {
                    ci[idx[q]] = cpv[q].x;
                    code[idx[q]] = (uint16_t)(cpv[q].y & 0xFFFFu);
                    if (D.has_user) ucode[idx[q]] = (uint16_t)(cpv[q].y >> 16);
                    c2b[idx[q]] = (uint16_t)(g0 + q * 64 + ln);
                }
            }
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
    // Hit h sits in LDS when it can: 8 bytes {first entry (21 bits) | entries (9) | lexicon (2), start (14) | length - 1 (6) | candidates
    // of the start position before it (12)} (pack_hit / unpack_hit), as many as the wavefront's LDS has room for behind the per-character arrays (the mean
    // sentence's ~230 hits fit the bulk generator's 4 KiB).  What does not fit -- the tail of a long sentence, a field out of range --
    // is staged in the sentence's region of global memory as before, with an all-ones marker in its LDS slot; only then does the
    // expansion below have to wait for this wave's stores.  (Was: every hit through global memory -- 16 bytes written, drained and read
    // back: 0.74 GB of the step's traffic and two dependent round trips per sentence.)
    uint2* const lhits = L.lhits;
    const uint32_t lcap = L.lcap;
    if (ln == 0) *hcount = 0;
    __syncthreads();
    uint32_t C = 0;
    bool any_long = false, any_global = false;
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        const uint32_t i = c0 + ln;
        uint32_t cnt = 0;
        uint64_t lmask = 0;
        bool is_long = false;
        if (i < n) {
            auto seen = [&](uint32_t v, uint32_t c, uint32_t end, uint32_t lex) {
                if (c == 0) return;  // a category without unknown-word entries contributes nothing (unknown.rs:118-130)
                const uint32_t h = atomicAdd(hcount, 1u);
                const uint32_t len = end - i;
                const bool packed = h < lcap && hit_packs(v, c, i, len, cnt);
                if (packed) lhits[h] = pack_hit(v, c, lex, i, len, cnt);
                else {
                    if (h < lcap) lhits[h] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                    if (h < region) hits[h] = make_uint4(v, c | (lex << 16), end | (i << 16), cnt);
                    any_global = true;
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
            set_lens(i, lmask);
        }
        uint32_t tot;
        const uint32_t ex = wave_exscan(cnt, tot);
        if (i < n) set_co(i, C + ex);
        C += tot;
        any_long |= __ballot(is_long) != 0;
    }
    any_global = __ballot(any_global) != 0;  // (wave-uniform from here on)
    // words > 64 chars need the generic pre-pass, > 65531 nodes need u32 indices: fused kernel
    if (C >= 65532 || any_long) { route(fallback); return none; }
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
    if (C > region) { route(fallback); return none; }  // denser than the region: fused path
    // The staged hits are read back by this wave only: its stores have to be complete (workgroup scope:
    // s_waitcnt vmcnt(0); the vector L1 is write-through and never held these lines).  An agent-scope
    // release would write the whole L2 back (buffer_wbl2) once per sentence.
    if (any_global) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else __syncthreads();  // (hits in LDS only: the LDS operations of one wave execute in order)
    PROF_MARK(1);

    // expand the hits: lanes = hits, every entry load independent of every other; a candidate becomes ONE 16-byte record --
    // {first cell of its left id's matrix row, (u16) word_cost | end-list slot << 16, word_idx, end_char | right_id << 16} -- at its
    // place in the reference's insertion order (cand_off[start] + candidates of that start before the hit).  One scattered store
    // per candidate: the generator is bound by the number of its scattered store requests (two 8-byte stores into separate arrays
    // cost 5 % more; the sweep's load phase does not notice the wider record)
    const uint32_t H = *hcount;  // <= C <= region
    const uint32_t row_cells = D.num_right;  // a left id's row of the connection matrix starts at cell left_id * num_right
#if VBT_FILL_ROUNDS > 1
    // (VBT_FILL_ROUNDS rounds of 64 hits at a time: all their hits and first entries are requested before the first candidate is stored,
    // so a round's entry loads do not queue up behind the previous round's stores in the in-order vmcnt)
    constexpr uint32_t kR = VBT_FILL_ROUNDS;
    for (uint32_t h0 = 0; h0 < H; h0 += 64 * kR) {
        uint4 hr[kR];
        Entry e0[kR];
#pragma unroll
        for (uint32_t r = 0; r < kR; ++r) {
            const uint32_t h = h0 + 64 * r + ln;
            hr[r] = make_uint4(0, 0, 0, 0);
            if (h < H) {
                const uint2 q = h < lcap ? lhits[h] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                hr[r] = q.y != 0xFFFFFFFFu ? unpack_hit(q) : hits[h];
            }
        }
#pragma unroll
        for (uint32_t r = 0; r < kR; ++r) {
            const uint32_t lex = hr[r].y >> 16;
            const Entry* __restrict__ ent = lex == 0 ? D.sys.entries : lex == 1 ? D.user.entries : D.unk_entries;
            e0[r] = ent[(hr[r].y & 0xFFFFu) ? hr[r].x : 0u];
        }
#pragma unroll
        for (uint32_t r = 0; r < kR; ++r) {
            const uint32_t c = hr[r].y & 0xFFFFu, lex = hr[r].y >> 16, end = hr[r].z & 0xFFFFu, pos = hr[r].z >> 16;
            if (c) {
                const Entry* __restrict__ ent = lex == 0 ? D.sys.entries : lex == 1 ? D.user.entries : D.unk_entries;
                const uint32_t dest = get_co(pos) + hr[r].w;
                const uint32_t slot0 = atomicAdd(&endc[end], c);  // the hit's run of slots in ends[end]
                A.g_cand[base + dest] = make_uint4((e0[r].left_right & 0xFFFFu) * row_cells, (e0[r].cost & 0xFFFFu) | (slot0 << 16),
                                                   (lex << 30) | e0[r].word_id, end | (e0[r].left_right & 0xFFFF0000u));
                for (uint32_t t0 = 1; t0 < c; t0 += 4) {  // (homographs: the rest of the run, four entries at a time)
                    Entry e[4];
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) e[q] = ent[hr[r].x + (t0 + q < c ? t0 + q : t0)];
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q)
                        if (t0 + q < c)
                            A.g_cand[base + dest + t0 + q] = make_uint4((e[q].left_right & 0xFFFFu) * row_cells, (e[q].cost & 0xFFFFu) | ((slot0 + t0 + q) << 16),
                                                                        (lex << 30) | e[q].word_id, end | (e[q].left_right & 0xFFFF0000u));
                }
            }
        }
    }
#else
    for (uint32_t h0 = 0; h0 < H; h0 += 64) {
        const uint32_t h = h0 + ln;
        uint4 hr = make_uint4(0, 0, 0, 0);
        if (h < H) {
            const uint2 q = h < lcap ? lhits[h] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            hr = q.y != 0xFFFFFFFFu ? unpack_hit(q) : hits[h];
        }
        if (h < H) {
            const uint32_t c = hr.y & 0xFFFFu, lex = hr.y >> 16, end = hr.z & 0xFFFFu, pos = hr.z >> 16;
            const Entry* __restrict__ ent = lex == 0 ? D.sys.entries : lex == 1 ? D.user.entries : D.unk_entries;
            const uint32_t dest = get_co(pos) + hr.w;
            const uint32_t slot0 = atomicAdd(&endc[end], c);  // the hit's run of slots in ends[end]
            for (uint32_t t0 = 0; t0 < c; t0 += 4) {
                Entry e[4];
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
#if VBT_ABLATE == 2  // (timing probe: no entry loads)
                    e[q].word_id = hr.x + q; e[q].left_right = hr.x & 0x00FF00FFu; e[q].cost = hr.x & 0xFFu;
#else
                    e[q] = ent[hr.x + (t0 + q < c ? t0 + q : t0)];
#endif
                }
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
#if VBT_ABLATE == 1  // (timing probe: the entry loads stay, no candidate is stored)
                    if (t0 + q < c && e[q].cost == 0x7FFFFFF0u + dest) {
#elif VBT_ABLATE == 3  // (timing probe: every candidate store goes to the sentence's first slots -- 64 lanes, 64 neighbouring slots)
                    if (t0 + q < c) {
                        const uint32_t k = ln + 0 * dest;
                        A.g_cand[base + k] = make_uint4((e[q].left_right & 0xFFFFu) * row_cells, (e[q].cost & 0xFFFFu) | ((slot0 + t0 + q) << 16),
                                                        (lex << 30) | e[q].word_id, end | (e[q].left_right & 0xFFFF0000u));
                    }
                    if (false) {
#else
                    if (t0 + q < c) {
#endif
                        const uint32_t k = dest + t0 + q;
                        A.g_cand[base + k] = make_uint4((e[q].left_right & 0xFFFFu) * row_cells, (e[q].cost & 0xFFFFu) | ((slot0 + t0 + q) << 16),
                                                        (lex << 30) | e[q].word_id, end | (e[q].left_right & 0xFFFF0000u));
                    }
                }
            }
        }
    }
#endif
    __syncthreads();
    // exclusive end-list offset of position p (0 .. n + 1): the cursors now hold the inclusive prefix
    auto eo = [&](uint32_t p) { return p == 0 ? 0u : p == 1 ? 1u : endc[p - 1]; };
    uint32_t passes = 0, maxcnt = 1;
#if VBT_GEN_RECORDS
    // The sweep's structural pre-pass done HERE for a sentence that lattice_lds will sweep whole: which (start node, start word) steps
    // the position sweep takes depends only on where words end (tokenizer.rs:106-138, control flow only), so the 8-byte pass
    // records of the assembly sweep loop (sweep_asm.hpp) can be laid out by the wave that holds the length masks in LDS anyway -- and
    // that runs in a kernel waiting for memory three quarters of its time, while lattice_lds is bound by what its waves issue.
    // Same state machine, same records as the pre-pass in lattice_sentence (which keeps doing it for what is swept in segments,
    // for the builds and modes without the assembly loop, and in the escape tiers): w bit i <=> a node ends at p + 1 + i, cur <=>
    // position p is reachable, a visited space run hands its visit to the position behind it.  LDS addresses in the records are
    // relative to the wave's arena (slot records at 0, candidate records behind the ET + 2 slot records); they are written to the
    // sentence's dead hit-staging region, and the header's third word becomes the exact pass count | 1 << 31.
    struct PreState { bool on, windowed; uint32_t stop, cur, pend, sn_eos, SL, offC, cap; uint64_t w; uint2* recs; };
    PreState pre;
    constexpr bool kSweepTakesRecords = (VBT_ASM_LOOP != 0) && (VBT_LDS_REC != 0);  // (the assembly loop over records in LDS: the only consumer)
    pre.on = kSweepTakesRecords && !A.lid_count && !A.direct_push && !D.matrix_wide && n < 8000u;
    pre.windowed = true; pre.stop = 0; pre.cur = 1; pre.pend = 0; pre.sn_eos = n; pre.SL = 0; pre.w = 0;
    pre.offC = 8u * (eo(n + 1) + 2u);
    pre.recs = reinterpret_cast<uint2*>(A.g_hits + base);
    pre.cap = (uint32_t)(region < 0x7FFFFFu ? 2u * region : 0xFFFFFFu);  // 8-byte records in 16-byte hit slots
    if (pre.offC + 8u * (C + 2u) > 65535u) pre.on = false;  // (16-bit LDS addresses; such a sentence is not swept whole anyway)
    {   // nothing to lay out for a sentence that no tier up to the segment tier holds whole, however few passes it takes
        const uint32_t t_max = A.seg_tier < A.n_tiers ? A.seg_tier : A.n_tiers - 1u;
        if (lattice_fixed_bytes(C, n, eo(n + 1), 1u) > A.tier_bytes[t_max]) pre.on = false;
    }
    auto put_rec = [&](uint32_t P, uint32_t p_beg, uint32_t np, uint32_t c_beg, uint32_t nc, uint32_t k, uint32_t r, uint32_t rounds) {
        const uint32_t np_r = np - kRoundPreds * r < kRoundPreds ? np - kRoundPreds * r : kRoundPreds;
        const uint32_t nc_r = nc - kRoundCands * k < kRoundCands ? nc - kRoundCands * k : kRoundCands;
        const uint32_t nu = (np_r + 3u) >> 2;
        const uint32_t fl = nu | (r == 0 ? 8u : 0u) | (r + 1 == rounds ? 16u : 0u);
        pre.recs[P] = make_uint2(((p_beg + kRoundPreds * r) << 3) | ((np < 4u ? np : 4u) << 16) | (nc_r << 24),
                                 (pre.offC + ((c_beg + kRoundCands * k) << 3)) | (fl << 16) | (np_r << 24));
    };
    // (lm, space, co_i, nc, eo_i, np: what the record loop below holds of position i anyway)
    auto pre_chunk = [&](auto space_c, uint32_t c0, uint64_t lm, uint32_t space, uint32_t co_i, uint32_t nc_i, uint32_t eo_i, uint32_t np_i) {
        constexpr bool kSp = decltype(space_c)::value;
        const uint32_t i = c0 + ln;
        const bool in = i < n;
        const bool is_space = kSp && in && space != 0;
        const uint64_t lmi = (in && !is_space) ? lm : 0ull;
        const uint32_t l_lo = (uint32_t)lmi, l_hi = (uint32_t)(lmi >> 32);
        const uint32_t gf = is_space ? (uint32_t)grp[i] : 0u;  // groupable run of a space position
        const uint64_t spm = kSp ? __ballot(is_space) : 0ull;
        const uint32_t cnt = n - c0 < 64 ? n - c0 : 64;
        uint64_t w = pre.w, vis = 0, visp = 0;
        uint32_t cur = pre.cur, pend = pre.pend, stop = 0;
        // (groups of 8 positions, the lane index of v_readlane in an SGPR: unrolled 64 times as in lattice_sentence the two instances
        // of this loop made the kernel 90 KB of code -- more than the instruction cache two CUs share -- and every phase of it slower)
        auto bits = [&](auto sp2_c) {
            constexpr bool kS = decltype(sp2_c)::value;
            for (uint32_t k0 = 0; k0 < cnt; k0 += 8)
#pragma unroll
            for (uint32_t kj = 0; kj < 8; ++kj) {
                const uint32_t k = k0 + kj;
                const uint64_t bit = 1ull << k;
                const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(l_hi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane(l_lo, k);
                if constexpr (kS) {
                    if (cur && !pend && (spm & bit)) {  // rare: a reachable space run
                        const uint32_t r = __builtin_amdgcn_readlane(gf, k);
                        if (c0 + k + r >= n) { pre.sn_eos = c0 + k; stop = 1; w = 0; }  // only spaces left: EOS connects here
                        else if (r > 63) { pre.windowed = false; stop = 1; w = 0; }
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
        };
        if constexpr (kSp) {
            // (a chunk without a space position and with no hand-over pending sweeps exactly as with ignore_space off)
            if (spm != 0 || pend) bits(std::true_type{}); else bits(std::false_type{});
        } else bits(std::false_type{});
        if (cnt < 64) { vis &= (1ull << cnt) - 1ull; visp &= (1ull << cnt) - 1ull; }
        pre.w = w; pre.cur = cur; pre.pend = pend; pre.stop = stop;
        const uint64_t any = vis | visp;
        const bool step = (any >> ln) & 1ull;  // lanes = the visited positions of the chunk: step (start node i, start word sw)
        uint32_t nsl = 0, p_beg = 0, np = 0, c_beg = 0, nc = 0, rounds = 0;
        if (step) {
            p_beg = eo_i; np = np_i;
            c_beg = co_i; nc = nc_i;  // (a space position's count is that of the position behind its run already)
            if (kSp && ((visp >> ln) & 1ull)) c_beg = get_co(i + gf);  // (< n: a run that reaches the end stops the sweep above)
            rounds = (np + kRoundPreds - 1) / kRoundPreds;
            nsl = rounds * ((nc + kRoundCands - 1) / kRoundCands);
        }
        uint32_t tot;
        const uint32_t ex = wave_exscan(nsl, tot);
        if (pre.SL + tot + 64u > pre.cap) { pre.on = false; return; }  // (wave-uniform; the region is 16 bytes per candidate slot: never in practice)
        for (uint32_t q = 0, k = 0, r = 0; q < nsl; ++q) {
            put_rec(pre.SL + ex + q, p_beg, np, c_beg, nc, k, r, rounds);
            if (++r == rounds) { r = 0; ++k; }
        }
        pre.SL += tot;
    };
#else
    struct { bool on; uint32_t stop; } pre{false, 0};
    auto pre_chunk = [&](auto, uint32_t, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) {};
#endif
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
            uint32_t nc_i = 0, eo_i = 0;
            if (i < n) {
                const uint32_t cinfo = ci[i];
                space = (D.space_cateset && (cinfo & D.space_cateset)) ? 0x80000000u : 0u;
                lm = get_lens(i);
                e = lm ? i + 64u - (uint32_t)__builtin_clzll(lm) : i + 1;
                co_i = get_co(i);
                uint32_t nc = get_co(i + 1) - co_i;
                eo_i = eo(i);
                if (space) {
                    const uint32_t sw = i + grp[i];
                    const uint64_t lw = sw < n ? get_lens(sw) : 0ull;
                    const uint32_t e2 = sw < n ? (lw ? sw + 64u - (uint32_t)__builtin_clzll(lw) : sw + 1) : n;
                    e = e2 > e ? e2 : e;
                    nc = sw < n ? get_co(sw + 1) - get_co(sw) : 0u;  // the step taken from a space position starts its words behind the run
                }
                cnt = eo(i + 1) - eo_i;
                nsl = step_passes(nc, cnt);
                nc_i = nc;
            }
            if (pre.on && !pre.stop) {
                if (D.space_cateset) pre_chunk(std::true_type{}, c0, lm, space, co_i, nc_i, eo_i, cnt);
                else pre_chunk(std::false_type{}, c0, lm, space, co_i, nc_i, eo_i, cnt);
            }
            uint32_t m = e;  // inclusive prefix maximum over the lanes
            m = wave_inscan_max_dpp(m);
            if (i < n) {
                const uint32_t upto = m > far ? m : far;  // furthest end of any candidate of the positions <= i (<= n)
                const uint64_t third = space ? (uint64_t)grp[i] : lm;
                const uint32_t yw = (nsl < 0x3FFFu ? nsl : 0x3FFFu) | (eo(upto + 1) << 14) | space;
                pc[i] = make_uint4(co_i | (eo_i << 16), yw, (uint32_t)third, (uint32_t)(third >> 32));
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
    uint32_t tier = fallback, pre_flag = 0u;
#if VBT_GEN_RECORDS
    if (pre.on && pre.windowed) {
        // + the EOS step (insert_eos(start_node), tokenizer.rs:138): predecessors = ends[sn_eos]
        const uint32_t y0 = eo(pre.sn_eos), y1 = pre.sn_eos < n ? eo(pre.sn_eos + 1) : eo(n + 1);
        const uint32_t np = y1 - y0, nsl = (np + kRoundPreds - 1) / kRoundPreds;
        const uint32_t exact = pre.SL + nsl;  // (<= the bound `passes`: every step was counted with its position's bound)
        // the records count only for a sentence that fits a tier up to the segment tier WHOLE (with its exact pass count), in LDS
        // the records' 16-bit addresses reach; everything else keeps the bound (a segmented sweep budgets its segments with it)
        const uint64_t fixed_x = lattice_fixed_bytes(C, n, eo(n + 1), exact);
        uint32_t tx = fallback;
        for (uint32_t t = 0; t < A.n_tiers; ++t)
            if (fixed_x <= A.tier_bytes[t]) { tx = t; break; }
        if (pre.SL + nsl + 64u <= pre.cap && tx < A.n_tiers && (A.seg_tier >= A.n_tiers || tx <= A.seg_tier) && A.tier_bytes[tx] <= 65536u) {
            for (uint32_t q = ln; q < nsl; q += 64) put_rec(pre.SL + q, y0, np, C, 1u, 0u, q, nsl);
            tier = tx; passes = exact; pre_flag = 0x80000000u;
        }
    }
#endif
    if (!pre_flag) {
        // smallest tier whose LDS holds the lattice arrays (connection costs are never staged)
        const uint64_t fixed = lattice_fixed_bytes(C, n, eo(n + 1), passes);
        for (uint32_t t = 0; t < A.n_tiers; ++t)
            if (t >= A.n_lean && fixed <= A.tier_bytes[t]) { tier = t; break; }  // (the lean instance only takes sentences with records)
        // longer sentences are swept in segments inside the segment tier instead of one huge LDS block (lattice_lds cuts anywhere;
        // what it cannot sweep there -- a window of end lists wider than the tier -- it hands to the escape tiers itself)
        if (A.seg_tier < A.n_tiers && tier > A.seg_tier) tier = A.seg_tier;
    }
    const uint4 hdr = make_uint4(n | (nb << 16), C | (tier << 16), passes | pre_flag, (uint32_t)(b0 - uniform64(A.offsets[0])));
    if (ln == 0) A.s_hdr[sid] = hdr;
    if (A.inline_lean && pre_flag && tier < A.n_lean) { if (ln == 0) A.s_tier[sid] = kRouteInline; }  // the caller sweeps it now
    else route(tier);
    PROF_MARK(2);
    if (A.prof && ln == 0) {
        unsigned long long* pr_ = A.prof + (size_t)(sid & (kProfSlots - 1)) * kProfWords;
        for (int i = 0; i < 3; ++i) atomicAdd(&pr_[i], (unsigned long long)prof_acc[i]);
        atomicAdd(&pr_[kProfPhases], 1ull);
    }
#undef PROF_MARK
    return hdr;
}

}  // namespace
}  // namespace vbt
