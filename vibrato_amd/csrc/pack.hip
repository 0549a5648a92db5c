// Token compaction (tokens from the per-sentence staging regions into sentence order) and the one-off expansion of a compact
// connector into the dense matrix the sweep reads.
#include "device_common.hpp"

namespace vbt {
namespace {

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
// Header of a packed result slot (Workspace::set_packed_output): {n_sentences u64, n_tokens u32, error flags u32, 0 x 4}.  A slot that
// is too small for this batch reports the records it holds (never more than were written) and says so in the flags word -- a consumer
// of the gathered slots on another rank sees nothing else of this rank's state (round-5 advisor).
__device__ __forceinline__ void write_out_header(const BatchArgs& A, uint32_t all) {
    if (!A.out_header) return;
    const bool over = all > A.tok_cap;
    A.out_header[0] = A.n; A.out_header[1] = 0u; A.out_header[2] = over ? A.tok_cap : all;
    A.out_header[3] = A.ctrl[kError] | (over ? (uint32_t)kErrTokCap : 0u);
    A.out_header[4] = 0u; A.out_header[5] = 0u; A.out_header[6] = 0u; A.out_header[7] = 0u;
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
    if (threadIdx.x == 0) {
        A.ctrl[kTotal] = running;
        write_out_header(A, running);
    }
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
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            A.ctrl[kTotal] = all;
            write_out_header(A, all);
        }
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
        if (to >= A.tok_cap) { atomicOr(&A.ctrl[kError], (uint32_t)kErrTokCap); break; }  // (a caller's slot that is too small for this batch)
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

// The connection matrix under a renumbering of the ids (Tokenizer::renumbered_image; the reference's MatrixConnector::
// map_connection_ids, matrix_connector.rs:99-116, on the device): one thread per destination cell, stores coalesced, the loads of a
// workgroup gather from one source row (31 KB for unidic: it stays in L2 while the row is being read).
template <typename CellT>
__global__ void __launch_bounds__(256) permute_matrix(const CellT* __restrict__ src, CellT* __restrict__ dst, const uint16_t* __restrict__ inv_left,
                                                      const uint16_t* __restrict__ inv_right, uint32_t num_right) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
    if (r >= num_right) return;
    dst[(size_t)l * num_right + r] = src[(size_t)inv_left[l] * num_right + inv_right[r]];
}


// The calibration sample of Tokenizer::maybe_calibrate (engine.hip): up to `want` sentences, spread evenly over the caller's batch
// (sentence j of the sample = sentence floor(j n / ns) of the batch -- an ordered corpus, headlines first and body later, is sampled
// over its whole length, not on its prefix), copied into the tokenizer's own buffers on the caller's stream: the counting sweep then
// runs on a side stream from there, after the caller's buffers may be gone.  sample_plan (one workgroup): lengths, their prefix =
// the sample's offsets, the number of sentences whose text fits `cap_bytes`; info = {sentences, bytes}.  Offsets that decrease or leave
// the declared window give an empty sample (the batch's own run reports them).
__global__ void __launch_bounds__(1024) sample_plan(const uint64_t* __restrict__ offsets, uint64_t n, uint64_t total_bytes, uint32_t want, uint64_t cap_bytes,
                                                    uint64_t* __restrict__ s_offs, uint32_t* __restrict__ s_src, uint32_t* __restrict__ info) {
    __shared__ uint32_t ws[16];
    __shared__ uint32_t bad_any, unfit;  // unfit: the first sample sentence that does not fit any more
    const uint64_t o0 = offsets[0], oN = offsets[n];
    const uint32_t ns = (uint32_t)(n < want ? n : want);
    if (threadIdx.x == 0) { bad_any = (oN < o0 || oN - o0 > total_bytes) ? 1u : 0u; unfit = ns; }
    __syncthreads();
    uint64_t running = 0;
    for (uint32_t j0 = 0; j0 < ns; j0 += 1024) {
        const uint32_t j = j0 + threadIdx.x;
        uint32_t len = 0, src = 0;
        bool huge = false;  // (a sentence of 4 MiB or more ends the sample; clipped, so that 1024 lengths add up in 32 bits)
        if (j < ns) {
            src = (uint32_t)(((unsigned __int128)j * n) / ns);
            const uint64_t a = offsets[src], b = offsets[src + 1];
            if (b < a || a < o0 || b > oN || b - a > 0xFFFFFFFFull) atomicOr(&bad_any, 1u);
            else { huge = b - a > 0x3FFFFFull; len = huge ? 0x3FFFFFu : (uint32_t)(b - a); }
            s_src[j] = src;
        }
        uint32_t tot;
        const uint32_t ex = block_exscan(len, ws, tot);
        if (j < ns) {
            s_offs[j] = running + ex;
            if (huge || running + ex + len > cap_bytes) atomicMin(&unfit, j);
        }
        running += tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t k = bad_any ? 0u : unfit;
        info[0] = k;  // (s_offs[k] = start of the first sentence that does not fit = end of the last one that does)
        if (k == ns) s_offs[ns] = running;
    }
}
// one wavefront per sample sentence: its bytes into the sample text
__global__ void __launch_bounds__(64) sample_copy(const uint8_t* __restrict__ text, const uint64_t* __restrict__ offsets, const uint64_t* __restrict__ s_offs,
                                                  const uint32_t* __restrict__ s_src, const uint32_t* __restrict__ info, uint8_t* __restrict__ s_text) {
    const uint32_t j = blockIdx.x;
    if (j >= info[0]) return;
    const uint32_t src = s_src[j];
    const uint64_t a = offsets[src], len = offsets[src + 1] - a, to = s_offs[j];
    for (uint64_t i = threadIdx.x; i < len; i += 64) s_text[to + i] = text[a + i];
}

}  // namespace

namespace kern {

void permute_matrix(hipStream_t stream, const void* src, void* dst, bool wide, const uint16_t* inv_left, const uint16_t* inv_right, uint32_t num_left, uint32_t num_right) {
    const dim3 grid((num_right + 255) / 256, num_left);
    if (wide) hipLaunchKernelGGL(vbt::permute_matrix<int32_t>, grid, dim3(256), 0, stream, static_cast<const int32_t*>(src), static_cast<int32_t*>(dst), inv_left, inv_right, num_right);
    else hipLaunchKernelGGL(vbt::permute_matrix<int16_t>, grid, dim3(256), 0, stream, static_cast<const int16_t*>(src), static_cast<int16_t*>(dst), inv_left, inv_right, num_right);
}

void sample_batch(hipStream_t stream, const uint8_t* text, const uint64_t* offsets, uint64_t n, uint64_t total_bytes, uint32_t want, uint64_t cap_bytes,
                  uint8_t* s_text, uint64_t* s_offs, uint32_t* s_src, uint32_t* info) {
    hipLaunchKernelGGL(vbt::sample_plan, dim3(1), dim3(1024), 0, stream, offsets, n, total_bytes, want, cap_bytes, s_offs, s_src, info);
    hipLaunchKernelGGL(vbt::sample_copy, dim3(want), dim3(64), 0, stream, text, offsets, s_offs, s_src, info, s_text);
}

uint32_t pack_split() { return kPackSplit; }
void tok_tile_scan(hipStream_t stream, const BatchArgs& a, uint32_t* tile_sums, uint32_t n_tiles) {
    hipLaunchKernelGGL(vbt::tok_tile_scan, dim3(1), dim3(1024), 0, stream, a, tile_sums, n_tiles);
}
void compact_tokens(hipStream_t stream, const BatchArgs& a, const uint32_t* tile_sums, uint32_t n_tiles, uint32_t scanned) {
    hipLaunchKernelGGL(vbt::compact_tokens, dim3(n_tiles * kPackSplit), dim3(kScanBlock), 0, stream, a, tile_sums, n_tiles, scanned);
}
void compact_tokens_out(uint32_t workgroups, hipStream_t stream, const BatchArgs& a, const uint32_t* tile_sums, uint32_t n_tiles, vbt_token_rec* out_tokens,
                        uint32_t* out_off, uint32_t* out_cnt) {
    hipLaunchKernelGGL(vbt::compact_tokens_out, dim3(workgroups), dim3(kScanBlock), 0, stream, a, tile_sums, n_tiles, out_tokens, out_off, out_cnt);
}
void expand_connector_i16(dim3 grid, const DevConnector& c, int16_t* out, uint32_t num_right, uint32_t num_left, uint32_t* range_flag) {
    hipLaunchKernelGGL(vbt::expand_connector<int16_t>, grid, dim3(256), 0, nullptr, c, out, num_right, num_left, range_flag);
}
void expand_connector_i32(dim3 grid, const DevConnector& c, int32_t* out, uint32_t num_right, uint32_t num_left, uint32_t* range_flag) {
    hipLaunchKernelGGL(vbt::expand_connector<int32_t>, grid, dim3(256), 0, nullptr, c, out, num_right, num_left, range_flag);
}

}  // namespace kern
}  // namespace vbt
