// The original single-kernel design: one wavefront does everything for a sentence, the lattice in LDS
// (VBT_FUSED=1: an A/B reference) or in a private global-memory slab (the fallback for what the pipeline cannot take: sentences
// >= 64 KiB, >= 65 532 candidates, dictionary words > 64 characters, > 63 skipped spaces in a row, windows of end lists wider than
// the largest LDS tier).  Same algorithm and results as gen.hip + lattice.hip, nothing shared with them but device_common.hpp.
#include "device_common.hpp"

namespace vbt {
namespace {

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

}  // namespace

namespace kern {

void fused_lds(uint32_t workgroups, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a, const uint32_t* in_list, const uint32_t* in_count,
               uint32_t* cursor, uint32_t* out_list, uint32_t* out_count) {
    auto k = D.matrix_wide ? vbt::tokenize_lds<true> : vbt::tokenize_lds<false>;
    hipLaunchKernelGGL(k, dim3(workgroups), dim3(64), lds_bytes, stream, D, a, lds_bytes, in_list, in_count, cursor, out_list, out_count);
}
void fused_global(uint32_t workgroups, hipStream_t stream, const DevDict& D, const BatchArgs& a, const uint32_t* in_list, const uint32_t* in_count, uint32_t* cursor) {
    hipLaunchKernelGGL(D.matrix_wide ? vbt::tokenize_global<true> : vbt::tokenize_global<false>, dim3(workgroups), dim3(64), 0, stream, D, a, in_list, in_count, cursor);
}

}  // namespace kern
}  // namespace vbt
