// Launchers of the device kernels (one translation unit per kernel family: gen.hip, lattice.hip, fused.hip, pack.hip), as plain
// functions the host side (engine.hip: Tokenizer, Workspace) calls.  Every launcher enqueues on `stream` and returns.
#pragma once
#include <hip/hip_runtime.h>

#include "engine.hpp"

namespace vbt {

// Compact connector (Raw / Dual) as expand_connector reads it (pack.hip); all pointers are device pointers.
struct DevConnector {
    const uint32_t* bases; const uint32_t* checks; const int32_t* costs;
    uint32_t n_bases, n_checks;
    const uint32_t* right_feats; const uint32_t* left_feats;
    uint32_t width;
    const int16_t* m; const uint16_t* right_map; const uint16_t* left_map;  // dual only (m == nullptr: raw)
    uint32_t m_num_right;
};

namespace kern {

// ---- gen.hip: input contract, candidate generation, work lists
void validate_batch(uint32_t blocks, hipStream_t stream, const BatchArgs& a, uint64_t total_bytes);
void gen_candidates(uint32_t n, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a);
void build_lists(uint32_t blocks, hipStream_t stream, const BatchArgs& a, int only_list);
void gen_candidates_large(uint32_t workgroups, uint32_t waves, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a, uint32_t level);
void gen_set_max_lds(int bytes);  // hipFuncAttributeMaxDynamicSharedMemorySize of gen_candidates_large

// ---- lattice.hip: the sweep (one instance per {ignore_space, i32 matrix cells}) and the resident Worker kernel
void lattice_lds(uint32_t workgroups, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a, uint32_t tier, uint32_t persistent);
void lattice_set_max_lds(int bytes);
// the lean instance for whole sentences that arrive with the generator's pass records (built for 5 waves per SIMD); the common build only
bool lattice_has_lean();
void gen_sweep(uint32_t n, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a);  // generator + lean sweep in one wave (needs lattice_has_lean())
void lattice_lean(uint32_t workgroups, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a, uint32_t tier);
// the segment tier's instance without the C++ loop, the `exact` mode and connection-id counting (i16 cells, < 8000 characters, tier <= 64 KiB)
void lattice_slim(uint32_t workgroups, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a, uint32_t tier);
void tokenize_serve(uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a, const uint8_t* h_text, uint32_t* ctl, uint32_t last_seq,
                    uint32_t idle_polls, uint32_t max_served);

// ---- fused.hip: the single-kernel fallback (global-memory lattice) and its LDS form (VBT_FUSED=1)
void fused_lds(uint32_t workgroups, uint32_t lds_bytes, hipStream_t stream, const DevDict& D, const BatchArgs& a, const uint32_t* in_list, const uint32_t* in_count,
               uint32_t* cursor, uint32_t* out_list, uint32_t* out_count);
void fused_global(uint32_t workgroups, hipStream_t stream, const DevDict& D, const BatchArgs& a, const uint32_t* in_list, const uint32_t* in_count, uint32_t* cursor);

// ---- pack.hip: token compaction, connector expansion
uint32_t pack_split();  // workgroups per packing tile
void tok_tile_scan(hipStream_t stream, const BatchArgs& a, uint32_t* tile_sums, uint32_t n_tiles);
void compact_tokens(hipStream_t stream, const BatchArgs& a, const uint32_t* tile_sums, uint32_t n_tiles, uint32_t scanned);
void compact_tokens_out(uint32_t workgroups, hipStream_t stream, const BatchArgs& a, const uint32_t* tile_sums, uint32_t n_tiles, vbt_token_rec* out_tokens,
                        uint32_t* out_off, uint32_t* out_cnt);
void expand_connector_i16(dim3 grid, const DevConnector& c, int16_t* out, uint32_t num_right, uint32_t num_left, uint32_t* range_flag);
void expand_connector_i32(dim3 grid, const DevConnector& c, int32_t* out, uint32_t num_right, uint32_t num_left, uint32_t* range_flag);
// dst[l][r] = src[inv_left[l]][inv_right[r]] (i16 cells, or i32 when `wide`): the matrix under a renumbering of the connection ids
void permute_matrix(hipStream_t stream, const void* src, void* dst, bool wide, const uint16_t* inv_left, const uint16_t* inv_right, uint32_t num_left, uint32_t num_right);

// the calibration sample (Tokenizer::maybe_calibrate): up to `want` sentences spread evenly over the batch, copied into s_text / s_offs
// (s_offs[0 .. k], k = info[0] = the sentences whose text fits cap_bytes; s_src: want words of scratch), on `stream`
void sample_batch(hipStream_t stream, const uint8_t* text, const uint64_t* offsets, uint64_t n, uint64_t total_bytes, uint32_t want, uint64_t cap_bytes,
                  uint8_t* s_text, uint64_t* s_offs, uint32_t* s_src, uint32_t* info);

}  // namespace kern
}  // namespace vbt
