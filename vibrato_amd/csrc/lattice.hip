// The lattice sweep and the resident Worker kernel (section 3.5).
// Reference: build_lattice_inner tokenizer.rs:94-139, Lattice::insert_node / search_min_node lattice.rs:103-151, insert_eos 85-101,
// append_top_nodes lattice.rs:159-168, MatrixConnector::cost matrix_connector.rs:79-125, Worker::tokenize worker.rs:49-55.
#include "gen_device.hpp"
#include "sweep_asm.hpp"

namespace vbt {
namespace {

// Kernel 2: the lattice sweep of one sentence per wavefront, entirely in LDS (one list entry per workgroup; the escape
// tiers run persistent waves).  Sentences whose lattice does not fit after all go to the fallback list (fused kernel
// with global scratch).
//
// What lives in LDS per (segment of a) sentence: per end-list slot an 8-byte record {lo = (0xFFFE - sequence) << 16 | right id,
// hi = min_cost biased to unsigned order} -- the low word is static and written by the load phase, the cost by the step that
// inserts the node; per candidate 8 bytes {first cell of its matrix row, byte offset of its slot record | word cost << 16}
// (the low half of the first word becomes the node's back pointer once its step is done); the token path; the first three pass
// records of the assembly loop.  The pass records themselves live in global memory (the sentence's dead hit-staging region).
// Nothing per character: the per-character records of gen_candidates are consumed straight from global memory by the
// reachability sweep, 64 positions at a time, and the end lists were laid out by gen_candidates (every candidate arrives
// with its slot).
//
// The recurrence (lattice.rs:103-151) runs over PASSES of <= 16 candidates x <= 16 predecessors of one sweep step.
// Lane = (candidate cl = lane >> 2, phase k = lane & 3) walks the predecessors 4 i + k: it reads the predecessor's record
// (four addresses per instruction, each broadcast to 16 lanes), adds the connection cost of its pair -- gathered a few passes
// ahead into a register ring -- and keeps the 64-bit minimum (cost, 0xFFFE - sequence of the predecessor): minimum cost, ties to
// the last inserted predecessor = the `<=` of lattice.rs:141-146.  At the end of the step two quad-permute levels combine the
// four phases, the word cost is added, the node's cost goes into its slot record and the winner's field becomes its back
// pointer.  LDS operations of one wave execute in order, so no barrier separates a pass from the next.  Two loops state this:
// the common build's in assembly over 8-byte vector-fetched records (sweep_asm.hpp: the one that is measured), and a C++ loop
// over 64-byte scalar records with precomputed lane masks (LPass) for i32 cells, sentences >= 8000 characters, tiers of more
// than 64 KiB and connection-id counting.
//
// A sentence whose lattice does not fit the tier's LDS is swept in segments cut at ANY position b (a multiple of 8 positions
// behind the segment's start, not behind a space): slots are numbered by end position over the whole sentence, so the nodes
// of the finished segment that end behind the cut are final and sit in the slot range [eo(b), window end); that range is
// moved to the front of the slot window and the next segment carries on -- its load phase touches only the slots of its own
// candidates, the reachability state (three scalars) stays in registers.
// `h`: the sentence's header (BatchArgs::s_hdr), wave-uniform.
#ifdef VBT_SEG_TRACE
#define SEG_TRACE(...) do { if (ln == 0) printf(__VA_ARGS__); } while (0)
#else
#define SEG_TRACE(...) do { } while (0)
#endif
// kSlim: the instance for what the default build routes to the segment tier on running text -- i16 cells, sentences short enough for
// the dead-predecessor sentinel, an LDS tier the records' 16-bit addresses reach, no connection-id counting: the assembly loop is the
// only loop, and the C++ loop, the `exact` instance and the counting phase are not compiled in (anything else: return 34, the caller
// hands the sentence to the general instance's escape tier).
template <bool kSpaceMode, bool kWide, bool kSlim = false>
__device__ __forceinline__ uint32_t lattice_sentence(const DevDict& D, const BatchArgs& A, uint32_t tier, uint32_t sid, uint4 h) {
    typedef __attribute__((address_space(3))) const uint64_t lds_cu64;
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t ln = threadIdx.x;
    const uint32_t lds_bytes = A.tier_bytes[tier];
    unsigned long long* const lid_count_ = kSlim ? nullptr : A.lid_count;  // (slim: neither counting nor profiling is compiled in)
    unsigned long long* const prof_ = kSlim ? nullptr : A.prof;
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
        uint64_t prof_t = prof_ ? clock64() : 0;
        unsigned long long* const pr_ = prof_ ? prof_ + (size_t)(sid & (kProfSlots - 1)) * kProfWords : nullptr;
#define PROF_MARK(i) do { if (prof_) { const uint64_t t_ = clock64(); if (ln == 0) atomicAdd(&pr_[i], (unsigned long long)(t_ - prof_t)); prof_t = t_; } } while (0)
        const uint32_t nT = h.x & 0xFFFFu, CT = h.y & 0xFFFFu, passesT = h.z & 0x7FFFFFFFu;
        const size_t slot0 = (size_t)h.w + (size_t)kSentenceSlack * sid;  // (sentence_slot: the header holds the byte offset relative to the batch)
        const size_t node0 = (size_t)A.node_factor * slot0;
        const uint4* __restrict__ pcg = A.g_pc + slot0;   // per-character records (+ the terminator at nT)
        const uint4* __restrict__ ndg = A.g_cand + node0;  // candidate records in insertion order: {first cell of the matrix row, word cost | slot << 16, word_idx, end_char | right id << 16}
        const uint32_t ET = CT + 1u;  // end-list slots of the sentence: one per candidate + BOS (= the terminator record's pcg[nT].z: not worth a round trip)
        const uint32_t kBosSeq = CT + 1;
        // The sentence's hit-staging region (dead after gen_candidates; 16 bytes per node slot): its lower half holds (total cost,
        // back pointer) of every node of a sentence that is swept in segments, its upper half the pass records of the current segment.
        const uint32_t nbT = h.x >> 16;
        const uint32_t half_bytes = 8u * A.node_factor * (nbT + kSentenceSlack);
        // where a dead predecessor's sentinel cost could meet a live cost, every predecessor's own field is tested instead (kDeadHi)
        const bool exact = !kSlim && (kWide || nT >= 8000u);
        if (kSlim && (nT >= 8000u || lds0 + lds_bytes > 65536u)) return 34;
        // the common build sweeps in assembly (sweep_asm.hpp) over 8-byte records (16-bit LDS addresses) -- in the sentence's LDS, or
        // (VBT_LDS_REC=0) vector-fetched from the region below; the 64-byte scalar records (LPass) feed the C++ loop of the other builds
        const bool vrec_mode = kSlim ? true : (VBT_ASM_LOOP && !kWide && !exact && kD == 2 && !lid_count_ && lds0 + lds_bytes <= 65536u);
        constexpr bool kLdsRec = VBT_LDS_REC != 0;
        const bool lds_rec = kLdsRec && vrec_mode;  // the pass records live in LDS: nothing bounds them but the tier
        // (a sentence that is swept whole dumps nothing: its records take the whole region)
        const bool whole = lattice_fixed_bytes(CT, nT, ET, passesT) <= lds_bytes && (lds_rec || passesT + 3 * kD + 4 <= 2 * half_bytes / (uint32_t)(sizeof(LPass) + 4));
        // The generator laid out the pass records of a sentence that is swept whole (gen_device.hpp: header word 2, bit 31; the word
        // is then the exact pass count): no per-character record is read, no pre-pass runs -- the records come in with the candidates.
        // Only where this instance would have built the same records itself: the assembly loop over records in LDS.
        const bool pre = (VBT_GEN_RECORDS != 0) && (h.z >> 31) != 0 && lds_rec && whole;
        LPass* const rec = reinterpret_cast<LPass*>(reinterpret_cast<char*>(A.g_hits + node0) + (whole ? 0u : half_bytes));
        const uint32_t rec_cap = (whole ? 2 * half_bytes : half_bytes) / (uint32_t)(sizeof(LPass) + 4);
        uint32_t* const rec_w3 = reinterpret_cast<uint32_t*>(rec + rec_cap);  // per pass: step totals for the connection-id counting
        uint2* const vrec = reinterpret_cast<uint2*>(rec);
        uint32_t seg_a = 0, seg_c = 0, seg_p = 0, sb = 0, m_in = 1, fail = 0;
        bool multi = false, done = false;
        uint32_t counted = lid_count_ ? __builtin_amdgcn_readfirstlane(A.s_counted[sid]) : 0u;
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
        if (lattice_fixed_bytes(CT - seg_c, nT - seg_a, ET - sb, passesT - seg_p) > budget || (!lds_rec && passesT - seg_p + 3 * kD + 4 > rec_cap)) {
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
                const bool fits = b <= nT && lattice_fixed_bytes((cx - seg_c) & 0xFFFFu, b - seg_a, wsl - sb, est) <= budget && (lds_rec || est + 3 * kD + 4 <= rec_cap);
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
        SEG_TRACE("tier %u seg [%u,%u) of %u: seg_c %u sb %u wend %u seg_pass %u budget %u m_in %u multi %d\n", tier, seg_a, seg_b, nT, seg_c, sb, wend, seg_pass, budget, m_in, (int)multi);
        const uint32_t n = seg_b - seg_a;
        const uint4* __restrict__ pc = pcg + seg_a;
        // record of the segment's end position (the terminator for the last segment; pre: all that is used of it is the candidate total)
        const uint4 rend = pre ? make_uint4(CT, 0u, 0u, 0u) : uniform4(pcg[seg_b]);
        const uint32_t C = ((rend.x & 0xFFFFu) - seg_c) & 0xFFFFu;
        const uint32_t E = wend - sb;  // slots of the window [eo(seg_a), wend); slot E is the EOS node's (last segment)
        if (E >= 8190u || m_in > E) SEG_TRACE("   retry/fail: E %u m_in %u\n", E, m_in);
        if (E >= 8190u || m_in > E) {  // the candidate records hold a slot's byte offset (slot * 8) in 16 bits: a shorter segment, or the next tier / the fused kernel
            if (budget > lds_bytes / 3 && !whole) { budget -= lds_bytes / 4; __syncthreads(); continue; }  // (a sentence taken for whole keeps its records where a segmented one dumps its nodes: the next tier sweeps it)
            fail = 26; break;
        }
        const uint4* __restrict__ nd = ndg + seg_c;

        Arena ar{g_smem, lds_bytes, 0, true};
        (void)ar.take<uint2>(E + 2);              // e_rec: the slot records
        uint2* cnd = ar.take<uint2>(C + 2);       // per candidate: {first cell of its matrix row (low half: its back pointer, once inserted), byte offset of its slot record | word_cost << 16}
        uint16_t* path = ar.take<uint16_t>(n + 4);  // the token path of the back-trace (tokens <= positions)
        // byte offsets of the characters (gen_candidates' c2b) of a sentence that is swept whole: requested with the candidates, read by emit
        uint16_t* c2bl = ar.take<uint16_t>(VBT_C2B_LDS ? n + 4 : 0);
        const bool c2b_lds = VBT_C2B_LDS && !multi && last_seg;
#if VBT_GUARD
        uint32_t* guard = ar.take<uint32_t>(16);
#define VBT_GUARD_CHECK(k) do { if (ar.ok && ln < 16 && guard[ln] != 0xDEADBEEFu + ln) atomicAdd(&A.ctrl[20 + (k)], 1u); } while (0)
#else
#define VBT_GUARD_CHECK(k) do { } while (0)
#endif
        // the pass records of the assembly loop: all of them (seg_pass bounds the segment's passes; + the empty ones behind the last), or
        // (VBT_LDS_REC=0) the first three, which its prologue reads from here instead of waiting for them to come back from global memory
        uint2* vhead = ar.take<uint2>(kLdsRec ? (seg_pass < (1u << 20) ? seg_pass : (1u << 20)) + 10u : 3u);
        const uint32_t sl_cap = lds_rec ? seg_pass + 2u : rec_cap > 3 * kD + 4 ? rec_cap - (3 * kD + 2) : 0u;
        if (!ar.ok || sl_cap < 3) SEG_TRACE("   retry/fail: arena ok %d used %u sl_cap %u (C %u E %u n %u)\n", (int)ar.ok, (uint32_t)ar.used, sl_cap, C, E, n);
        if (!ar.ok || sl_cap < 3) {  // the estimate was too low: try a shorter segment before giving up
            if (budget > lds_bytes / 3 && !whole) { budget -= lds_bytes / 4; __syncthreads(); continue; }  // (a sentence taken for whole keeps its records where a segmented one dumps its nodes: the next tier sweeps it)
            fail = 26; break;
        }
        const uint32_t offC = lds0 + (uint32_t)(reinterpret_cast<char*>(cnd) - g_smem);
#if VBT_GUARD
        if (ln < 16) guard[ln] = 0xDEADBEEFu + ln;
#endif

        // the per-character records of the first 64 positions are requested now, ahead of the candidate loads: by the time the
        // reachability sweep wants them they have arrived (the sweep of a chunk then prefetches the next chunk's)
        uint4 rc_next = make_uint4(0, 0, 0, 0), rn_next = rc_next;
        if (!pre) { rc_next = pc[ln < n ? ln : n]; rn_next = pc[ln < n ? ln + 1 : n]; }
        // (pre: the first two records per lane are requested with the candidates)
        const uint2* __restrict__ grec = reinterpret_cast<const uint2*>(A.g_hits + node0);
        uint2 pr_early[2] = {make_uint2(0, 0), make_uint2(0, 0)};
        if (pre) {
#pragma unroll
            for (uint32_t u = 0; u < 2; ++u) { const uint32_t P = u * 64 + ln; pr_early[u] = grec[P < seg_pass ? P : 0u]; }
        }
        // ---- load: candidates from global (every record carries its slot); EOS ----
        // The first kEarly records per lane are only REQUESTED here: the reachability sweep below needs nothing of them, so their
        // round trip runs under it and they are put into LDS behind it (load_rest); what is left follows there.
        const uint32_t fld0 = 0xFFFEu - seg_c;  // own field of candidate c of this segment: fld0 - c
        auto put_cand = [&](uint32_t c, const uint4& r) {
            const uint32_t es = (r.y >> 16) - sb;
            // never inserted until a sweep step reaches its start position (exact mode: the field says so, the step writes it)
            const uint32_t fld = exact ? 0xFFFFu : ((fld0 - c) & 0xFFFFu);
            e_rec[es] = make_uint2((fld << 16) | (r.w >> 16), kDeadHi);
            cnd[c] = make_uint2(r.x, (es << 3) | (r.y << 16));
        };
        constexpr uint32_t kEarly = VBT_EARLY_LOADS;
        const uint16_t* __restrict__ c2bg = A.g_c2b + slot0;
        uint32_t cb_early[2] = {0u, 0u};
        if (c2b_lds) {
#pragma unroll
            for (uint32_t u = 0; u < 2; ++u) { const uint32_t i = u * 64 + ln; cb_early[u] = c2bg[i <= n ? i : n]; }
        }
        uint4 r_early[kEarly ? kEarly : 1];
#pragma unroll
        for (uint32_t u = 0; u < kEarly; ++u) {
            const uint32_t c = u * 64 + ln;
            r_early[u] = nd[c < C ? c : 0u];
        }
        auto load_rest = [&]() {
            if (pre) {  // the generator's records: LDS addresses relative to the arena
                const uint2 adj = make_uint2(lds0, lds0);
#pragma unroll
                for (uint32_t u = 0; u < 2; ++u) { const uint32_t P = u * 64 + ln; if (P < seg_pass) vhead[P] = make_uint2(pr_early[u].x + adj.x, pr_early[u].y + adj.y); }
                for (uint32_t P = 128 + ln; P < seg_pass; P += 64) { const uint2 r = grec[P]; vhead[P] = make_uint2(r.x + adj.x, r.y + adj.y); }
            }
            if (c2b_lds) {
#pragma unroll
                for (uint32_t u = 0; u < 2; ++u) { const uint32_t i = u * 64 + ln; if (i <= n) c2bl[i] = (uint16_t)cb_early[u]; }
                for (uint32_t i = 128 + ln; i <= n; i += 64) c2bl[i] = c2bg[i];
            }
#pragma unroll
            for (uint32_t u = 0; u < kEarly; ++u) {
                const uint32_t c = u * 64 + ln;
                if (c < C) put_cand(c, r_early[u]);
            }
            for (uint32_t c0 = 64 * kEarly; c0 < C; c0 += 64 * 4) {  // 4 independent 16-byte loads per lane in flight
                uint4 r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t c = c0 + u * 64 + ln;
                    r[u] = nd[c < C ? c : 0u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t c = c0 + u * 64 + ln;
                    if (c < C) put_cand(c, r[u]);
                }
            }
            if (last_seg && ln == 0) {
                // EOS pseudo candidate (insert_eos, lattice.rs:85-101): left_id 0 (matrix row 0), word cost 0
                cnd[C] = make_uint2(0u, E << 3);
                e_rec[E] = make_uint2(((fld0 - C) & 0xFFFFu) << 16, kDeadHi);
            }
        };
        if constexpr (kEarly == 0) load_rest();
#if !VBT_LOOP_PROF
        PROF_MARK(3);
#else
        if (prof_) prof_t = clock64();  // (slot 3 is the loop's LDS wait in these builds)
#endif

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
            if (vrec_mode) {
                // {slot record of the first predecessor | phases that see a predecessor in some unit of the step << 16 | candidates << 24,
                //  record of the first candidate | (units | first round << 3 | last round << 4) << 16 | predecessors of this round << 24}
                const uint32_t fl = nu | (r == 0 ? 8u : 0u) | (r + 1 == rounds ? 16u : 0u);
                const uint2 vr = make_uint2((offK + ((p_beg + kRoundPreds * r) << 3)) | ((np < 4u ? np : 4u) << 16) | (nc_r << 24),
                                            (offC + ((c_beg + kRoundCands * k) << 3)) | (fl << 16) | (np_r << 24));
                if constexpr (kLdsRec) vhead[P] = vr;
                else { vrec[P] = vr; if (P < 3) vhead[P] = vr; }
                return;
            }
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
        if (!pre) {
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
                auto bits = [&](auto sp_c) {
                    constexpr bool kSp = decltype(sp_c)::value;
#pragma unroll
                    for (uint32_t k = 0; k < 64; ++k) {
                        if ((k & 7u) == 0 && k >= cnt) break;
                        const uint64_t bit = 1ull << k;
                        const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(l_hi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane(l_lo, k);
                        if constexpr (kSp) {
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
                };
                if constexpr (kSpaceMode) {
                    // (a chunk without a space position and with no hand-over pending sweeps exactly as with ignore_space off: 11 instead
                    // of ~19 scalar instructions per position on the path taken -- most chunks of running text)
                    if (spm != 0 || pend) bits(std::true_type{}); else bits(std::false_type{});
                } else {
#pragma unroll
                    for (uint32_t k = 0; k < 64; ++k) {  // (spelled out: this instance's code stays what it was before the lambda above existed)
                        if ((k & 7u) == 0 && k >= cnt) break;
                        const uint64_t bit = 1ull << k;
                        const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(l_hi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane(l_lo, k);
                        w |= cur ? m : 0ull;
                        vis |= cur ? bit : 0ull;
                        cur = (uint32_t)w & 1u;
                        w >>= 1;
                    }
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
        if (pre) SL = seg_pass;  // (exact; the EOS step's records are among them)
        else if (last_seg) {
            // + the EOS step (insert_eos(start_node), tokenizer.rs:138): predecessors = ends[sn_eos]
            // (the builtin returns int: shifted as it comes, an end-list offset of 32 768 or more -- a sentence of more than 32 767
            // lattice nodes -- would be sign-extended and EOS would be connected to garbage; found in round 5 by the first test with
            // sentences of 6 500+ characters that stay in the pipeline)
            const uint32_t y0 = (uint32_t)__builtin_amdgcn_readfirstlane(pc[sn_eos].x) >> 16;
            const uint32_t y1 = sn_eos < n ? (uint32_t)__builtin_amdgcn_readfirstlane(pc[sn_eos + 1].x) >> 16 : ET;
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
        SEG_TRACE("   S %u SL %u overflow %d sn_eos %u\n", S, SL, (int)overflow, sn_eos);
        if (overflow || SL >= (1u << 18)) {  // more passes than estimated (gen_candidates bounds them per position)
            if (budget > lds_bytes / 3 && !whole) { budget -= lds_bytes / 4; __syncthreads(); continue; }  // (a sentence taken for whole keeps its records where a segmented one dumps its nodes: the next tier sweeps it)
            fail = 29; break;
        }
        // Empty passes behind the last one (no units, no lanes): the sweep loop runs in trips of kD passes and reads kD + 1 records
        // ahead, i.e. up to record SL + 2 kD.  Their issue halves sit in the records [SL, SL + 2 kD + 2), their consume halves kD
        // records further on (the consume halves in [SL, SL + kD) are those of the last kD real passes).
        if (vrec_mode) {
            if (ln < 10) {  // (no candidates, no units: records are read up to SL + 7)
                if constexpr (kLdsRec) vhead[SL + ln] = make_uint2(offK, offC);
                else { vrec[SL + ln] = make_uint2(offK, offC); if (SL + ln < 3) vhead[SL + ln] = make_uint2(offK, offC); }
            }
        } else if (ln < 2 * kD + 2) {
            LPass& I = rec[SL + ln];
            I.w0 = offK; I.w1 = offC;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) I.m[i] = 0ull;
            LPass& Cn = rec[SL + kD + ln];
            Cn.w0c = offK; Cn.w1c = offC; Cn.lm = 0ull; Cn.vm = 0ull;
        }
        // the records are read back through the scalar cache: this wave's stores complete (workgroup scope: s_waitcnt vmcnt(0); the
        // vector L1 is write-through), then the scalar cache forgets whatever it holds of this region (an earlier segment's records)
        PROF_MARK(4);
        VBT_GUARD_CHECK(0);
        if constexpr (kEarly != 0) load_rest();
        VBT_GUARD_CHECK(1);
        if (lds_rec) {
            // records in LDS: the LDS operations of one wave execute in order -- a compiler-level fence is all the loop needs
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (vrec_mode) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (vector loads read them back: nothing to invalidate)
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        }
#if !VBT_LOOP_PROF
        PROF_MARK(3);  // (the candidates' way into LDS counts as load time wherever it happens)
#else
        if (prof_) prof_t = clock64();
#endif

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
            if constexpr (VBT_ASM_LOOP && !kExact && !kWide && kD == 2) {
                if (vrec_mode) {
                    // the loop in assembly (sweep_asm.hpp): same LDS layout, same results
                    const uint32_t sl_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)SL);
                    const uint32_t offk_v = offK;
                    const uint32_t hd_v = lds0 + (uint32_t)(reinterpret_cast<char*>(vhead) - g_smem);
#if VBT_LDS_REC
#define VBT_REC_OPERANDS [rp] "v"(hd_v)
#else
#define VBT_REC_OPERANDS [rp] "s"(rbase), [hd] "v"(hd_v)
#endif
#if VBT_LOOP_PROF
                    // (cycles parked at the loop's two waits, left in the still unused token path array: phase slots 5 and 3)
                    const uint32_t plds = lds0 + (uint32_t)(reinterpret_cast<char*>(path) - g_smem);
                    asm volatile(VBT_SWEEP_TEXT
                                 :: VBT_REC_OPERANDS, [sl] "s"(sl_s), [rs] "s"(rsrc), [ln] "v"(ln), [offk] "v"(offk_v), [plds] "v"(plds)
                                 : VBT_SWEEP_CLOBBERS);
                    if (prof_ && ln == 0) {
                        const uint64_t* q = reinterpret_cast<const uint64_t*>(path);
                        atomicAdd(&pr_[5], (unsigned long long)q[0]); atomicAdd(&pr_[3], (unsigned long long)q[1]);
                        if (VBT_LOOP_PROF == 2) {  // (whole iterations by kind: slots 5 / 3 = cycles of the common / the other iterations, 0 / 1 = their counts)
                            atomicAdd(&pr_[0], (unsigned long long)(uint32_t)q[2]); atomicAdd(&pr_[1], (unsigned long long)(q[2] >> 32));
                        }
                    }
#else
                    asm volatile(VBT_SWEEP_TEXT
                                 :: VBT_REC_OPERANDS, [sl] "s"(sl_s), [rs] "s"(rsrc), [ln] "v"(ln), [offk] "v"(offk_v)
                                 : VBT_SWEEP_CLOBBERS);
#endif
                    return;
                }
            }
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
        VBT_GUARD_CHECK(2);

        // a node of this segment: its cost word and its back pointer (sequence of its best predecessor)
        auto node_cost = [&](uint32_t c) { return e_rec[(cnd[c].y & 0xFFFFu) >> 3].y ^ 0x80000000u; };
        auto node_pred = [&](uint32_t c) { return 0xFFFEu - (cnd[c].x & 0xFFFFu); };
        if (multi) {
            // leave (total cost, back pointer) of every node of the segment in the sentence's (dead) hit-staging region
            uint2* __restrict__ nb = reinterpret_cast<uint2*>(A.g_hits + node0) + seg_c;
            for (uint32_t c = ln; c < C; c += 64) nb[c] = make_uint2(node_cost(c), node_pred(c));
        }
        if (lid_count_) {
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
                if (eos_step) { if (ln == 0) atomicAdd(&lid_count_[0], (unsigned long long)live); }
                else for (uint32_t c = c_beg + ln; c < c_beg + nc; c += 64) atomicAdd(&lid_count_[nd[c].x / D.num_right], (unsigned long long)live);
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
                if (c2b_lds) { o.start_byte = c2bl[stp]; o.end_byte = c2bl[en]; }
                else { o.start_byte = c2b[stp]; o.end_byte = c2b[en]; }
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
        if (prof_ && ln == 0) {
            atomicAdd(&pr_[kProfPhases + 1], (unsigned long long)prof_S);
            atomicAdd(&pr_[kProfPhases + 2], (unsigned long long)prof_SL);
            atomicAdd(&pr_[kProfPhases + 3], (unsigned long long)CT);
        }
#undef PROF_MARK
    }
    return 0;
}

#if VBT_ASM_LOOP && VBT_LDS_REC && VBT_GEN_RECORDS && VBT_C2B_LDS
#define VBT_HAS_LEAN 1
// The lean instance of the sweep: a sentence that fits its tier WHOLE and arrives with the generator's pass records (header word 2,
// bit 31: gen_device.hpp) -- four sentences in five on running text.  Nothing of lattice_sentence's machinery for the rest is here
// (segments, hand-over, the pre-pass, the C++ loop, connection-id counting, retries): header -> candidates, records and byte offsets
// into LDS -> the assembly loop (sweep_asm.hpp) -> back-trace -> token records.  Same LDS layout, same records, same results; what
// differs is what the compiler has to keep alive around the loop: this instance is built for FIVE waves per SIMD (96 VGPRs) with an
// 8 KiB tier -- the loop is bound by VALU issue at 74 % utilisation with four, a fifth wave fills the gaps -- where the general
// instance needs 128 VGPRs (round 5 ran the general instance at five waves: 18 VGPRs spilled, and lost).
// Returns 0, or a reason for the caller to hand the sentence to the fused kernel (cannot happen: the generator sized it exactly).
template <bool kSpaceMode>
__device__ __forceinline__ uint32_t lattice_whole(const DevDict& D, const BatchArgs& A, uint32_t lds_bytes, uint32_t sid, uint4 h) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t ln = threadIdx.x;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)g_smem;
    const uint32_t n = h.x & 0xFFFFu, C = h.y & 0xFFFFu, SL = h.z & 0x7FFFFFFFu;  // characters, candidates, passes (exact)
    const size_t slot0 = (size_t)h.w + (size_t)kSentenceSlack * sid;
    const size_t node0 = (size_t)A.node_factor * slot0;
    const uint4* __restrict__ ndg = A.g_cand + node0;
    const uint2* __restrict__ grec = reinterpret_cast<const uint2*>(A.g_hits + node0);
    const uint16_t* __restrict__ c2bg = A.g_c2b + slot0;
    const uint32_t E = C + 1u;       // end-list slots: one per candidate + BOS (slot 0); slot E is the EOS node's
    const uint32_t kBosSeq = C + 1u;
    if (!(h.z >> 31) || E >= 8190u || lds0 + lds_bytes > 65536u) return 26;
    Arena ar{g_smem, lds_bytes, 0, true};
    uint2* const e_rec = ar.take<uint2>(E + 2);
    uint2* const cnd = ar.take<uint2>(C + 2);
    uint16_t* const path = ar.take<uint16_t>(n + 4);
    uint16_t* const c2bl = ar.take<uint16_t>(n + 4);
    uint2* const vhead = ar.take<uint2>(SL + 10u);
    if (!ar.ok) return 26;
    const uint32_t offK = lds0, offC = lds0 + (uint32_t)(reinterpret_cast<char*>(cnd) - g_smem);
    // everything the sentence needs from global memory is requested up front: the first two candidate records, pass records and
    // byte offsets per lane; what is left follows in rounds of 64
#if VBT_ABLATE_LEAN == 5  // (timing probe: the load phase alone with 8 instead of 16 bytes per candidate -- wrong data, half the traffic)
    const uint2* __restrict__ nd8 = reinterpret_cast<const uint2*>(ndg);
    auto ld8 = [&](uint32_t c) { const uint2 q = nd8[c]; return make_uint4(q.x, q.y & 0x00FFFFFFu, q.x, q.y); };
    uint4 r0 = ld8(ln < C ? ln : 0u), r1 = ld8(64 + ln < C ? 64 + ln : 0u);
#else
    uint4 r0 = ndg[ln < C ? ln : 0u], r1 = ndg[64 + ln < C ? 64 + ln : 0u];
#endif
#if VBT_LEAN_UPFRONT > 2  // (candidates 128 .. 64 VBT_LEAN_UPFRONT - 1 requested with the first two rounds instead of round by round behind them)
    uint4 rx[VBT_LEAN_UPFRONT - 2];
#pragma unroll
    for (uint32_t q = 0; q < VBT_LEAN_UPFRONT - 2; ++q) rx[q] = ndg[128 + 64 * q + ln < C ? 128 + 64 * q + ln : 0u];
#endif
    uint2 p0 = grec[ln < SL ? ln : 0u], p1 = grec[64 + ln < SL ? 64 + ln : 0u];
    uint32_t cb0 = c2bg[ln <= n ? ln : n], cb1 = c2bg[64 + ln <= n ? 64 + ln : n];
    auto put_cand = [&](uint32_t c, const uint4& r) {
        const uint32_t es = r.y >> 16;
        e_rec[es] = make_uint2((((0xFFFEu - c) & 0xFFFFu) << 16) | (r.w >> 16), kDeadHi);
        cnd[c] = make_uint2(r.x, (es << 3) | (r.y << 16));
    };
    if (ln == 0) {
        e_rec[0] = make_uint2(((0xFFFEu - kBosSeq) & 0xFFFFu) << 16, 0x80000000u);  // BOS: cost 0, right id 0 (lattice.rs:72-83)
        cnd[C] = make_uint2(0u, E << 3);                                              // EOS: left id 0, word cost 0 (lattice.rs:85-101)
        e_rec[E] = make_uint2(((0xFFFEu - C) & 0xFFFFu) << 16, kDeadHi);
    }
    if (ln < C) put_cand(ln, r0);
    if (64 + ln < C) put_cand(64 + ln, r1);
#if VBT_ABLATE_LEAN == 5
    for (uint32_t c = 128 + ln; c < C; c += 64) put_cand(c, ld8(c));
#elif VBT_LEAN_UPFRONT > 2
#pragma unroll
    for (uint32_t q = 0; q < VBT_LEAN_UPFRONT - 2; ++q)
        if (128 + 64 * q + ln < C) put_cand(128 + 64 * q + ln, rx[q]);
    for (uint32_t c = 64 * VBT_LEAN_UPFRONT + ln; c < C; c += 64) put_cand(c, ndg[c]);
#else
    for (uint32_t c = 128 + ln; c < C; c += 64) put_cand(c, ndg[c]);
#endif
    if (ln < SL) vhead[ln] = make_uint2(p0.x + lds0, p0.y + lds0);
    if (64 + ln < SL) vhead[64 + ln] = make_uint2(p1.x + lds0, p1.y + lds0);
    for (uint32_t P = 128 + ln; P < SL; P += 64) { const uint2 r = grec[P]; vhead[P] = make_uint2(r.x + lds0, r.y + lds0); }
    if (ln < 10) vhead[SL + ln] = make_uint2(offK, offC);  // the empty passes behind the last one (records are read up to SL + 7)
    if (ln <= n) c2bl[ln] = (uint16_t)cb0;
    if (64 + ln <= n) c2bl[64 + ln] = (uint16_t)cb1;
    for (uint32_t i = 128 + ln; i <= n; i += 64) c2bl[i] = c2bg[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {   // ---- fused gather + cost recurrence: the assembly loop (matrix_connector.rs:79-85, lattice.rs:103-151) ----
        const uint64_t mb = (uint64_t)reinterpret_cast<uintptr_t>(D.matrix);
        u32x4 rsrc;
        rsrc.x = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)mb);
        rsrc.y = ((uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(mb >> 32)) & 0xFFFFu) | (2u << 16);
        rsrc.z = (uint32_t)__builtin_amdgcn_readfirstlane(D.matrix_bytes);
        rsrc.w = 0x00020000u;
        const uint32_t sl_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)SL);
        const uint32_t hd_v = lds0 + (uint32_t)(reinterpret_cast<char*>(vhead) - g_smem);
#if VBT_ABLATE_LEAN != 1 && VBT_ABLATE_LEAN != 4 && VBT_ABLATE_LEAN != 5  // (timing probes of the lean instance, results wrong: 1 = no loop, no back-trace, no records; 2 = loop, nothing behind it; 3 = no records; 4 = the loop skipped only)
        asm volatile(VBT_SWEEP_TEXT :: [rp] "v"(hd_v), [sl] "s"(sl_s), [rs] "s"(rsrc), [ln] "v"(ln), [offk] "v"(offK) : VBT_SWEEP_CLOBBERS);
#else
        asm volatile("" :: "v"(hd_v), "s"(sl_s), "s"(rsrc), "v"(ln), "v"(offK) : "memory");
#endif
    }
#if VBT_ABLATE_LEAN == 1 || VBT_ABLATE_LEAN == 2 || VBT_ABLATE_LEAN == 5
    if (ln == 0) A.tok_cnt[sid] = e_rec[1].y & 0u;
    return 0;
#endif
    // ---- back-trace + token records (append_top_nodes lattice.rs:159-168, token.rs:21-92) ----
    auto node_cost = [&](uint32_t c) { return e_rec[(cnd[c].y & 0xFFFFu) >> 3].y ^ 0x80000000u; };
    auto node_pred = [&](uint32_t c) { return 0xFFFEu - (cnd[c].x & 0xFFFFu); };
    uint32_t T = 0;
#if VBT_LEAN_REG_TRACE
    // The walk from EOS over registers instead of LDS: every lane holds the back pointers of candidates ln, ln + 64, ... (one round of LDS
    // reads for all of them), and a step of the walk is a v_readlane -- a few cycles where a dependent ds_read_u16 takes an LDS round trip
    // (~28 steps per sentence; 6 % of the kernel's time as LDS reads: profiles/EXPERIMENTS.md, the ablation probes).
    if (C < 64u * 5u) {
        uint32_t pr[5];
#pragma unroll
        for (uint32_t q = 0; q < 5; ++q) pr[q] = 64 * q + ln <= C ? node_pred(64 * q + ln) : kBosSeq;
        uint32_t seq = C;  // (wave-uniform from here)
        for (;;) {
            const uint32_t q = seq >> 6, l = seq & 63u;
            uint32_t nx;
            switch (q) {
                case 0: nx = (uint32_t)__builtin_amdgcn_readlane((int)pr[0], (int)l); break;
                case 1: nx = (uint32_t)__builtin_amdgcn_readlane((int)pr[1], (int)l); break;
                case 2: nx = (uint32_t)__builtin_amdgcn_readlane((int)pr[2], (int)l); break;
                case 3: nx = (uint32_t)__builtin_amdgcn_readlane((int)pr[3], (int)l); break;
                default: nx = (uint32_t)__builtin_amdgcn_readlane((int)pr[4], (int)l); break;
            }
            seq = nx;
            if (seq == kBosSeq || T >= n || seq > C) break;  // (seq > C cannot happen: back pointers point at inserted candidates)
            if (ln == 0) path[T] = (uint16_t)seq;
            ++T;
        }
    } else
#endif
    if (ln == 0) {
        uint32_t seq = node_pred(C);
        while (seq != kBosSeq && T < n) { path[T++] = (uint16_t)seq; seq = node_pred(seq); }
    }
    T = (uint32_t)__builtin_amdgcn_readfirstlane((int)T);
    __syncthreads();
#if VBT_ABLATE_LEAN == 3 || VBT_ABLATE_LEAN == 4
    if (ln == 0) A.tok_cnt[sid] = T & 0u;
    return 0;
#endif
    if (ln == 0) { A.tok_cnt[sid] = T; if (T) atomicAdd(&A.tile_sums[sid / kScanTile], T); }
    const uint4* __restrict__ pcg = A.g_pc + slot0;
    for (uint32_t t = ln; t < T; t += 64) {
        const uint32_t c = path[T - 1 - t];
        const uint4 r = ndg[c];
        uint32_t stp = t ? ndg[path[T - t]].w & 0xFFFFu : 0u;  // a token starts where its predecessor on the path ends ...
        if constexpr (kSpaceMode) {
            if (stp < n) {
                const uint4 rp = pcg[stp];
                if (rp.y >> 31) stp += rp.z;  // ... behind the space run there (tokenizer.rs:113-125)
            }
        }
        const uint32_t en = r.w & 0xFFFFu;
        vbt_token_rec o;
        o.start_char = stp; o.end_char = en;
        o.start_byte = c2bl[stp]; o.end_byte = c2bl[en];
        o.word_idx = r.z;
        o.total_cost = (int32_t)node_cost(c);
        A.tok_stage[slot0 + t] = o;
    }
    return 0;
}

#ifndef VBT_LEAN_WAVES
#define VBT_LEAN_WAVES 6
#endif
template <bool kSpaceMode>
__global__ void __launch_bounds__(64, VBT_LEAN_WAVES) lattice_lean(DevDict D, BatchArgs A, uint32_t tier) {
#if VBT_ABLATE_LEAN == 7  // (timing probe: the launch alone)
    if (A.n != 0xFFFFFFFFu) return;
#endif
    const uint32_t* list = A.lists + (size_t)tier * A.list_stride + A.list_off;
    const uint32_t count = A.cctrl[2 * tier];
    const uint32_t item = blockIdx.x;  // one list entry per workgroup (the grid covers the batch), newest entries first: see lattice_lds
    if (item >= count) return;
    const uint32_t sid = __builtin_amdgcn_readfirstlane(list[count - 1 - item]);
    const uint4 hq = A.s_hdr[sid];
    const uint4 h = make_uint4(__builtin_amdgcn_readfirstlane(hq.x), __builtin_amdgcn_readfirstlane(hq.y), __builtin_amdgcn_readfirstlane(hq.z), __builtin_amdgcn_readfirstlane(hq.w));
#if VBT_ABLATE_LEAN == 6  // (timing probe: launch, list entry and header)
    if (threadIdx.x == 0 && h.x == 0xFFFFFFFFu) A.tok_cnt[sid] = 0;
    return;
#endif
    const uint32_t fail = lattice_whole<kSpaceMode>(D, A, A.tier_bytes[tier], sid, h);
    if (fail) {
        if (threadIdx.x == 0) atomicAdd(&A.ctrl[fail < 32 ? fail : 28], 1u);
        list_push_fb(A, sid);
    }
}

// Generator and lean sweep in ONE wave: gen_one, and when it routed its sentence to a lean tier (whole, with pass records) the same wave
// sweeps it at once -- what gen_one left in global memory is read back by the wave that wrote it (L2 hits; a workgroup-scope fence is
// all it takes, as in tokenize_serve) -- so at any moment some of a CU's waves wait for memory (generator) while others issue VALU
// (sweep).  Everything else is filed as by gen_candidates.
template <bool kSpaceMode>
__global__ void __launch_bounds__(64, VBT_LEAN_WAVES) gen_sweep(DevDict D, BatchArgs A, uint32_t lds_bytes) {
    if (batch_rejected(A)) return;
    const uint32_t sid = A.sid0 + blockIdx.x;
    const uint4 hq = gen_one(D, A, sid, lds_bytes);
    const uint4 h = make_uint4(__builtin_amdgcn_readfirstlane(hq.x), __builtin_amdgcn_readfirstlane(hq.y), __builtin_amdgcn_readfirstlane(hq.z), __builtin_amdgcn_readfirstlane(hq.w));
    const uint32_t tier = (h.y >> 16) & 0xFFu;
    if (!(tier < A.n_lean && (h.z >> 31))) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t fail = lattice_whole<kSpaceMode>(D, A, A.tier_bytes[tier], sid, h);
    if (fail) {
        if (threadIdx.x == 0) atomicAdd(&A.ctrl[fail < 32 ? fail : 28], 1u);
        list_push_fb(A, sid);
    }
}
#else
#define VBT_HAS_LEAN 0
#endif

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
    // (Round 5 tried to do without the list for the bulk of a batch -- workgroup -> sentence id -> header, two dependent round trips
    // fewer in front of the candidate loads: no faster at 5 waves per SIMD, and 5 % SLOWER when the sentences are visited in their
    // linear order -- the ~5000 in flight then neighbours in memory -- than in the order build_lists files them, blocks of 1024 in
    // the order their workgroups happen to finish; profiles/EXPERIMENTS.md.)
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
        const uint4 hq = A.s_hdr[sid];  // ONE load: sizes, pass bound, where the sentence's regions are
        const uint4 h = make_uint4(__builtin_amdgcn_readfirstlane(hq.x), __builtin_amdgcn_readfirstlane(hq.y), __builtin_amdgcn_readfirstlane(hq.z), __builtin_amdgcn_readfirstlane(hq.w));
        const uint32_t fail = lattice_sentence<kSpaceMode, kWide>(D, A, tier, sid, h);
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

#if VBT_HAS_LEAN
#ifndef VBT_SLIM_WAVES
#define VBT_SLIM_WAVES 5
#endif
// The segment tier's launch of the default build on i16 cells without counting: lattice_sentence<.., kSlim> (see there).
template <bool kSpaceMode>
__global__ void __launch_bounds__(64, VBT_SLIM_WAVES) lattice_slim(DevDict D, BatchArgs A, uint32_t tier) {
    if (A.tier_prio) __builtin_amdgcn_s_setprio(2);
    const uint32_t* list = A.lists + (size_t)tier * A.list_stride + A.list_off;
    const uint32_t count = A.cctrl[2 * tier];
    const uint32_t item = blockIdx.x;
    if (item >= count) return;
    const uint32_t sid = __builtin_amdgcn_readfirstlane(list[count - 1 - item]);
    const uint4 hq = A.s_hdr[sid];
    const uint4 h = make_uint4(__builtin_amdgcn_readfirstlane(hq.x), __builtin_amdgcn_readfirstlane(hq.y), __builtin_amdgcn_readfirstlane(hq.z), __builtin_amdgcn_readfirstlane(hq.w));
    const uint32_t fail = lattice_sentence<kSpaceMode, false, true>(D, A, tier, sid, h);
    if (fail) {
        const bool escape = tier + 1 < A.n_tiers && fail != 27;
        if (threadIdx.x == 0 && !escape) atomicAdd(&A.ctrl[fail < 32 ? fail : 28], 1u);
        if (escape) list_push(A, tier + 1, sid); else list_push_fb(A, sid);
    }
}
#endif

// Worker::tokenize() latency path (worker.rs:49-55; the 3-call loop of tokenize/src/main.rs:78-82): one wavefront, one sentence at
// a time, and NO launch per sentence: the kernel stays resident and serves the worker's calls out of its pinned host block.
//   written by the kernel: ctl[0] status = the sequence number of the last sentence served (written last, system-scope release: the
//       host spins on it), ctl[1] token count, ctl[2] 1 = the kernel has left (idle for `idle_polls` polls, or told to), ctl[3] outcome of
//       the last sentence: 0 = done, 1 = it needs the batch pipeline (longer than the generator's LDS, a window of end lists wider than
//       the sweep's LDS, a word > 64 characters...)
//   written by the host, ONE aligned 16-byte group the kernel polls with one PCIe read: ctl[4] doorbell = sequence number of the
//       sentence the host wants (written last, release), ctl[5] bytes of that sentence, ctl[6] 1 = leave now
// The text comes straight out of the pinned block (`h_text`, one PCIe round trip: 16 bytes per lane per request into a device copy),
// the generator and the sweep run back to back in the same wave (what gen_one leaves in global memory is read back by the wave
// that wrote it: a workgroup-scope fence is all it takes), and the token records, their count and the status word go straight back
// into pinned host memory (posted writes): no copy engine, no launch, no allocation.  Before it leaves for idleness the kernel
// raises ctl[4] and looks at the doorbell once more: a call that rang in between is either served or finds ctl[4] set and
// starts the kernel again (capi.cpp).  `idle_polls` = 0: one sentence, then out (the launch-per-call form of round 3, kept for A/B).
template <bool kSpaceMode, bool kWide>
__global__ void __launch_bounds__(64) tokenize_serve(DevDict D, BatchArgs A, uint32_t lds_bytes, const uint8_t* h_text, uint32_t* ctl, uint32_t last_seq,
                                                       uint32_t idle_polls, uint32_t max_served) {
    const uint32_t ln = threadIdx.x;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // (one 16-byte read of the host's group: the doorbell is the word the host writes last, so a new doorbell comes with its length)
    auto poll = [&]() {
        u32x4 g = {0, 0, 0, 0};
        if (ln == 0) {
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(reinterpret_cast<const u32x4*>(ctl + 4)) : "memory");
        }
        return make_uint4((uint32_t)__builtin_amdgcn_readfirstlane((int)g.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)g.y),
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)g.z), 0u);
    };
    uint32_t idle = 0, served = 0;
    for (;;) {
        // ---- wait for the doorbell (lane 0 polls pinned host memory; everything below is wave-uniform) ----
        const uint4 hg = poll();
        const uint32_t bell = hg.x, nb = hg.y, leave = hg.z;
        if (bell != last_seq + 1u) {
            if (leave || ++idle > idle_polls) {
                if (ln == 0) __hip_atomic_store(&ctl[2], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
                const uint4 again = poll();
                if (leave || again.x != last_seq + 1u) return;
                if (ln == 0) __hip_atomic_store(&ctl[2], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // rang while leaving: stay
                idle = 0;
                continue;
            }
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        idle = 0;
        // (acquire at system scope: the vector L1 may still hold lines of the previous sentence's text -- the kernel never ends between
        // two sentences, so nothing else invalidates them)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        // ---- one sentence ----
        {
            const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(h_text);  // (the pinned block is padded to 16 bytes)
            u32x4* dst = reinterpret_cast<u32x4*>(const_cast<uint8_t*>(A.text));
            for (uint32_t i = ln; i * 16 < nb; i += 64) dst[i] = __builtin_nontemporal_load(&src[i]);
            uint64_t* offs = const_cast<uint64_t*>(A.offsets);
            if (ln == 0) { offs[0] = 0; offs[1] = nb; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t st = 0;
        if (nb) {
            gen_one(D, A, 0u, lds_bytes);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t tier = __builtin_amdgcn_readfirstlane((uint32_t)A.s_tier[0]);
            const uint4 hq = A.s_hdr[0];
            const uint4 h = make_uint4(__builtin_amdgcn_readfirstlane(hq.x), __builtin_amdgcn_readfirstlane(hq.y), __builtin_amdgcn_readfirstlane(hq.z), __builtin_amdgcn_readfirstlane(hq.w));
            if (tier == 0u) st = lattice_sentence<kSpaceMode, kWide>(D, A, 0u, 0u, h) ? 1u : 0u;
            else if (tier != 0xFFu) st = 1u;  // 0xFF: an empty sentence, tok_cnt = 0 is already written
        } else if (ln == 0) A.tok_cnt[0] = 0;
        // every lane's token stores have to be visible to the host before the status word is (the host spins on it):
        // system-scope release by all lanes, then one releasing store
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        last_seq = bell;
        if (ln == 0) {
            __hip_atomic_store(&ctl[3], st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&ctl[0], bell, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (idle_polls == 0) return;
        // Residency is bounded: a Worker that is called in a tight loop would otherwise keep this kernel on the device for ever, and
        // every device-synchronising call of another thread (hipFree of a pooled workspace, hipHostFree, a caller's own
        // hipDeviceSynchronize) waits for it.  After max_served sentences the kernel raises "has left" and goes; the next call
        // finds the word set and starts another (capi.cpp: the same handshake as leaving for idleness).
        if (max_served && ++served >= max_served) {
            if (ln == 0) __hip_atomic_store(&ctl[2], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        __syncthreads();
    }
}

template <typename F>
auto pick(const DevDict& D, F&& f) {  // the instance for {ignore_space, i32 matrix cells}
    return D.space_cateset ? (D.matrix_wide ? f(std::true_type{}, std::true_type{}) : f(std::true_type{}, std::false_type{}))
                           : (D.matrix_wide ? f(std::false_type{}, std::true_type{}) : f(std::false_type{}, std::false_type{}));
}

}  // namespace

namespace kern {

void lattice_lds(uint32_t workgroups, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a, uint32_t tier, uint32_t persistent) {
    auto k = pick(D, [](auto s, auto w) { return &vbt::lattice_lds<decltype(s)::value, decltype(w)::value>; });
    hipLaunchKernelGGL(k, dim3(workgroups), dim3(64), lds_bytes, stream, D, a, tier, persistent);
}
bool lattice_has_lean() { return VBT_HAS_LEAN != 0; }
void lattice_slim(uint32_t workgroups, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a, uint32_t tier) {
#if VBT_HAS_LEAN
    if (D.space_cateset) hipLaunchKernelGGL(vbt::lattice_slim<true>, dim3(workgroups), dim3(64), lds_bytes, stream, D, a, tier);
    else hipLaunchKernelGGL(vbt::lattice_slim<false>, dim3(workgroups), dim3(64), lds_bytes, stream, D, a, tier);
#else
    (void)workgroups; (void)lds_bytes; (void)stream; (void)D; (void)a; (void)tier;
#endif
}
void gen_sweep(uint32_t n, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a) {
#if VBT_HAS_LEAN
    if (D.space_cateset) hipLaunchKernelGGL(vbt::gen_sweep<true>, dim3(n), dim3(64), lds_bytes, stream, D, a, lds_bytes);
    else hipLaunchKernelGGL(vbt::gen_sweep<false>, dim3(n), dim3(64), lds_bytes, stream, D, a, lds_bytes);
#else
    (void)n; (void)lds_bytes; (void)stream; (void)D; (void)a;
#endif
}
void lattice_lean(uint32_t workgroups, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a, uint32_t tier) {
#if VBT_HAS_LEAN
    if (D.space_cateset) hipLaunchKernelGGL(vbt::lattice_lean<true>, dim3(workgroups), dim3(64), lds_bytes, stream, D, a, tier);
    else hipLaunchKernelGGL(vbt::lattice_lean<false>, dim3(workgroups), dim3(64), lds_bytes, stream, D, a, tier);
#else
    (void)workgroups; (void)lds_bytes; (void)stream; (void)D; (void)a; (void)tier;
#endif
}
void lattice_set_max_lds(int bytes) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(vbt::lattice_lds<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(vbt::lattice_lds<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(vbt::lattice_lds<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(vbt::lattice_lds<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
}
void tokenize_serve(uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a, const uint8_t* h_text, uint32_t* ctl, uint32_t last_seq,
                    uint32_t idle_polls, uint32_t max_served) {
    auto k = pick(D, [](auto s, auto w) { return &vbt::tokenize_serve<decltype(s)::value, decltype(w)::value>; });
    hipLaunchKernelGGL(k, dim3(1), dim3(64), lds_bytes, stream, D, a, lds_bytes, h_text, ctl, last_seq, idle_polls, max_served);
}

}  // namespace kern
}  // namespace vbt
