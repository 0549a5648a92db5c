// Device-side pieces shared by the kernels of the MI355X tokenizer (gfx950, wave64): wave-level scans and minima, the packed keys
// of the fused fallback, the per-sentence regions of the workspace, the geometry of the sweep kernel's passes and the small helpers
// the generators share.  Included by gen.hip, lattice.hip, fused.hip and pack.hip; everything here is per translation unit
// (anonymous namespace).  DESIGN.md section 3 is the map.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "engine.hpp"
#include "kernels.hpp"

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
constexpr uint8_t kRouteInline = 0xFB;  // ... of a sentence the generator's own wave swept (gen_sweep): in no list

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

// Geometry of the sweep kernel (lattice_lds).  A sweep step -- all candidates of one start word x all nodes ending at its start node --
// is cut into PASSES of up to kRoundCands candidates x up to kRoundPreds predecessors.  In a pass lane = (candidate cl = lane >> 2,
// phase k = lane & 3): the lane walks the predecessors j = 4 i + k, i = 0 .. 3 ("units": one instruction stream per unit covers
// 4 predecessors x 16 candidates), keeps the minimum of its own phase in registers, and the four phases of a candidate are
// combined with two quad-permute levels at the end of the step.  VBT_DEPTH = passes whose connection costs are in flight.
#ifndef VBT_DEPTH
#define VBT_DEPTH 2
#endif
#ifndef VBT_NO_GATHER
#define VBT_NO_GATHER 0  // ceiling experiment (sweep_asm.hpp): the assembly loop without its matrix gather, WRONG RESULTS by design
#endif
#ifndef VBT_LOOP_PROF
#define VBT_LOOP_PROF 0  // developer aid (tools/phase_profile.py on a variant build): cycles parked at the assembly sweep loop's two waits
#endif
#ifndef VBT_GEN_RECORDS
#define VBT_GEN_RECORDS 1  // the bulk generator lays out the sweep's pass records for the sentences lattice_lds sweeps whole (gen_device.hpp)
#endif
// gen_one's expansion of the staged hits: rounds of 64 hits whose first entries are requested together, ahead of the first store (1 = round by round).
// Measured (round 6, tools/dbg/fill_ab.sh): 2 rounds neutral (0.559-0.565 vs 0.554-0.557 ms), 4 rounds 0.62-0.63 ms -- the expansion is not waiting for its own stores.
#ifndef VBT_FILL_ROUNDS
#define VBT_FILL_ROUNDS 1
#endif
#ifndef VBT_ABLATE  // timing probes of gen_one's expansion (results are wrong: run with VBT_SKIP_SWEEP=1); tools/dbg/gen_ablate.sh
#define VBT_ABLATE 0
#endif
#ifndef VBT_LEAN_UPFRONT  // lattice_whole: rounds of 64 candidate records requested before the first is used (2: the first form; 4 / 5: sweep 0.643-0.645 -> 0.629-0.634 ms)
#define VBT_LEAN_UPFRONT 5
#endif
#ifndef VBT_LEAN_REG_TRACE  // lattice_whole: the back-trace over registers (v_readlane) instead of dependent LDS reads -- measured slower (sweep 0.627-0.637 -> 0.646-0.659 ms): off
#define VBT_LEAN_REG_TRACE 0
#endif
#ifndef VBT_ABLATE_LEAN  // timing probes of lattice_whole (lattice.hip); tools/dbg/lean_ablate.py
#define VBT_ABLATE_LEAN 0
#endif
#ifndef VBT_CPINFO
#define VBT_CPINFO 1  // the generators read character class and trie codes of a code point in one 8-byte load (DevDict::cpinfo); 0: separate tables (A/B)
#endif
#ifndef VBT_GENLONG_PROF
#define VBT_GENLONG_PROF 0  // developer aid (tools/dbg/genlong_profile.py on a variant build): gen_long's wall cycles between its barriers
#endif
#ifndef VBT_ASM_LOOP
#define VBT_ASM_LOOP 1  // the sweep loop of the common build in assembly (sweep_asm.hpp; needs VBT_DEPTH 2); 0: the C++ loop everywhere
#endif
#ifndef VBT_LAT_WAVES
// waves per SIMD lattice_lds is compiled for.  4 (128 VGPRs: no VGPR spills around the assembly loop) with a 10 KiB tier beats 5 (96 VGPRs, 18 spilled)
// with an 8 KiB tier since round 5 took the loop's memory waits away: 0.830 vs 0.874 ms on the headline, 2.17 vs 2.72 ms on the dense law
// (profiles/EXPERIMENTS.md; round 4 measured the opposite, 1.064 vs 1.127 ms, when the loop still waited for its pass records)
#define VBT_LAT_WAVES 4
#endif
#ifndef VBT_LDS_REC
#define VBT_LDS_REC 1  // the assembly loop's 8-byte pass records in the sentence's LDS (sweep_asm.hpp); 0: round 4's records in global memory
#endif
#ifndef VBT_LDS_HITS
#define VBT_LDS_HITS 1  // gen_one stages its trie hits in what is left of its LDS (8-byte packed records) and only the overflow in global memory
#endif
#ifndef VBT_GUARD
#define VBT_GUARD 0  // developer aid: 64 guard bytes between the token path and the pass records, checked at four points (ctrl[20..23])
#endif
#ifndef VBT_C2B_LDS
#define VBT_C2B_LDS 1  // a whole sentence's character -> byte offsets are pulled into LDS with its candidates: emit's records need no second round trip
#endif
#ifndef VBT_EARLY_LOADS
#define VBT_EARLY_LOADS 4  // candidate records per lane requested in front of the reachability sweep and consumed behind it (0: load phase first, as in round 4)
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
// LDS bytes of the lattice arrays of lattice_lds for a (segment of a) sentence of n positions with C candidates, a window of
// E end-list slots and at most `passes` sweep passes: 8 bytes per slot, 8 per candidate, 2 + 2 per position (the token path, the
// byte offsets of the characters), 8 per pass record (+ the empty ones behind the last).  Must over-estimate the Arena carve there; gen_candidates routes sentences to LDS
// tiers with it.
__host__ __device__ __forceinline__ uint64_t lattice_fixed_bytes(uint32_t C, uint32_t n, uint32_t E, uint32_t passes) {
    return 8ull * (E + 2ull) + 8ull * (C + 2ull) + (VBT_C2B_LDS ? 4ull : 2ull) * (n + 4ull) + 48  // (+ alignment; VBT_LDS_REC=0: the first three pass records of the assembly loop)
           + (VBT_LDS_REC ? 8ull * (passes + 10ull) : 0ull) + (VBT_GUARD ? 72ull : 0ull);
}
// Cost word of a slot whose node was never inserted (its start position is never visited).  Biased cost 0xC0000000 = +2^30: with
// 16-bit connection and word costs a sentence of < 8000 characters keeps every live cost inside +-2^29, so such a predecessor
// loses every minimum without being tested for; lattice_sentence tests the slot's own field instead where that bound does not
// hold (i32 matrix cells, longer sentences).
constexpr uint32_t kDeadHi = 0xC0000000u;

extern __shared__ __attribute__((aligned(16))) char g_smem[];

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

// A staged trie hit in LDS: 8 bytes {first entry (21 bits) | entries (9) | lexicon (2), start position (14) | length - 1 (6) | candidates of
// the start position before the hit (12)}; the 16-byte form in global memory is {first entry, entries | lexicon << 16, end | start << 16,
// candidates before}.  An all-ones second word marks "this hit is in global memory" (start 16383 + length 64 + 4095 candidates: never packed).
__device__ __forceinline__ bool hit_packs(uint32_t first, uint32_t c, uint32_t start, uint32_t len, uint32_t before) {
    return first < (1u << 21) && c < 512u && start < (1u << 14) && len <= 64u && before < 4095u;
}
__device__ __forceinline__ uint2 pack_hit(uint32_t first, uint32_t c, uint32_t lex, uint32_t start, uint32_t len, uint32_t before) {
    return make_uint2(first | (c << 21) | (lex << 30), start | ((len - 1u) << 14) | (before << 20));
}
__device__ __forceinline__ uint4 unpack_hit(uint2 q) {
    const uint32_t pos = q.y & 0x3FFFu;
    return make_uint4(q.x & 0x1FFFFFu, ((q.x >> 21) & 0x1FFu) | ((q.x >> 30) << 16), (pos + ((q.y >> 14) & 63u) + 1u) | (pos << 16), q.y >> 20);
}

// gen_one's per-character working arrays in its wavefront's LDS (~26 bytes per character).  `ok` = they fit: the test by which
// gen_one files what does not fit for gen_long.
struct GenOneLds {
    uint64_t* lens;
    uint32_t *ci, *cand_off;
    uint16_t *code, *ucode, *grp;
    uint32_t *endc, *hcount;
    uint2* lhits;     // what is left of the wavefront's LDS: staged trie hits, 8 bytes each (gen_one)
    uint32_t lcap;    // hits it holds
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
    L.lhits = ar.take<uint2>(0);
    L.lcap = VBT_LDS_HITS && ar.ok && lds_bytes > ar.used ? (uint32_t)((lds_bytes - ar.used) / sizeof(uint2)) : 0u;
    return L;
}
// the smallest level of gen_long whose LDS holds the sentence
__device__ __forceinline__ uint32_t gen_long_level(const BatchArgs& A, uint32_t n, uint32_t nb, bool has_user) {
    const uint64_t need = gen_long_bytes(n, nb, has_user);
    uint32_t lv = 0;
    while (lv + 1 < (uint32_t)kGenLevels && need > A.gen_level_bytes[lv]) ++lv;
    return lv;
}

}  // namespace
}  // namespace vbt
