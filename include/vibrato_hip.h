/*
 * vibrato_hip.h -- C ABI of libvibrato_hip.so: an MI355X (gfx950) drop-in for the
 * tokenize() hot path of daac-tools/vibrato 0.5.2.
 *
 * The reference has no FFI layer; its boundary is the public Rust API of crate
 * `vibrato` (vibrato/src/lib.rs:52-78).  Each entry point below names the Rust item
 * it replaces (paths relative to /root/reference/vibrato/src).  INTEGRATION.md shows
 * the Rust shim (`extern "C"` block + safe wrappers) a maintainer would add.
 *
 * Conventions
 *   - every fallible call returns a vbt_status; vbt_last_error() returns the
 *     thread-local message of the last failure (mirrors VibratoError, errors.rs:7-42)
 *   - handles are opaque; strings are {ptr,len} UTF-8, never NUL-terminated
 *   - tokenization never silently falls back to a CPU path: without a usable
 *     gfx950 device every vbt_tokenizer_* / vbt_*tokenize* call fails with VBT_ERR_DEVICE
 */
#ifndef VIBRATO_HIP_H
#define VIBRATO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VBT_API __attribute__((visibility("default")))

typedef enum vbt_status {
    VBT_OK = 0,
    VBT_ERR_INVALID_ARGUMENT = 1, /* VibratoError::InvalidArgument, errors.rs:13 */
    VBT_ERR_INVALID_FORMAT = 2,   /* VibratoError::InvalidFormat,   errors.rs:16 */
    VBT_ERR_INVALID_STATE = 3,    /* VibratoError::InvalidState,    errors.rs:19 */
    VBT_ERR_PARSE_INT = 4,        /* VibratoError::ParseInt,        errors.rs:25 */
    VBT_ERR_UTF8 = 5,             /* VibratoError::Utf8,            errors.rs:37 */
    VBT_ERR_DEVICE = 100,         /* HIP runtime / no gfx950 device (new: the reference is CPU-only) */
    VBT_ERR_UNSUPPORTED = 101     /* input outside what the device image can represent */
} vbt_status;

/* LexType, dictionary.rs:30-40 */
typedef enum vbt_lex_type { VBT_LEX_SYSTEM = 0, VBT_LEX_USER = 1, VBT_LEX_UNKNOWN = 2 } vbt_lex_type;

/* Output modes of the `tokenize` CLI, tokenize/src/main.rs:83-127 */
typedef enum vbt_format_mode { VBT_FORMAT_MECAB = 0, VBT_FORMAT_WAKATI = 1, VBT_FORMAT_DETAIL = 2 } vbt_format_mode;

typedef struct vbt_dict vbt_dict;
typedef struct vbt_tokenizer vbt_tokenizer;
typedef struct vbt_worker vbt_worker;
typedef struct vbt_batch vbt_batch;
typedef struct vbt_workspace vbt_workspace;

/* Fixed 24-byte token record written by the device (one per best-path node).
 * Fields are what Token::{range_char, range_byte, word_idx, total_cost} return
 * (token.rs:21-92); positions are relative to the sentence. */
typedef struct vbt_token_rec {
    uint32_t start_char, end_char; /* Token::range_char, token.rs:21-24 */
    uint32_t start_byte, end_byte; /* Token::range_byte, token.rs:28-32 */
    uint32_t word_idx;             /* WordIdx: lex_type << 30 | word_id (word_idx.rs:5-11) */
    int32_t total_cost;            /* Token::total_cost, token.rs:89-92 */
} vbt_token_rec;

/* Full token view (Token accessors, token.rs:21-92). surface points into the
 * caller's sentence bytes / batch copy, feature into dictionary memory. */
typedef struct vbt_token {
    const char* surface; size_t surface_len; /* Token::surface, token.rs:36-39 */
    const char* feature; size_t feature_len; /* Token::feature, token.rs:49-54 */
    uint32_t start_char, end_char, start_byte, end_byte;
    uint32_t lex_type, word_id;              /* Token::lex_type / word_idx */
    uint16_t left_id, right_id;              /* Token::left_id / right_id, token.rs:64-75 */
    int16_t word_cost;                       /* Token::word_cost, token.rs:78-87 */
    int32_t total_cost;
} vbt_token;

VBT_API const char* vbt_last_error(void);
/* 1 when the bytes are a Rust `str` (strict UTF-8), else 0: what every text-taking entry point requires. */
VBT_API int vbt_utf8_valid(const char* utf8, size_t len);

/* ---- Dictionary ------------------------------------------------------------ */

/* SystemDictionaryBuilder::from_readers(lex.csv, matrix.def, char.def, unk.def), builder.rs:64-89 */
VBT_API int vbt_dict_from_sources(const char* lex, size_t lex_len, const char* matrix_def, size_t matrix_len,
                                  const char* char_def, size_t char_len, const char* unk_def, size_t unk_len,
                                  vbt_dict** out);
/* Same, with the connection matrix given in binary: data[left*num_right+right] (the layout of
 * MatrixConnector, connector/matrix_connector.rs:11-15,47). A 459 MiB matrix has no sane text form. */
VBT_API int vbt_dict_from_sources_binmatrix(const char* lex, size_t lex_len, const int16_t* matrix,
                                            uint32_t num_right, uint32_t num_left, const char* char_def,
                                            size_t char_len, const char* unk_def, size_t unk_len, vbt_dict** out);
/* SystemDictionaryBuilder::from_readers_with_bigram_info(lex.csv, bigram.right, bigram.left, bigram.cost, char.def, unk.def,
 * dual_connector), builder.rs:111-160: the connection costs come from the compact bigram model (RawConnector
 * connector/raw_connector.rs; dual != 0: DualConnector connector/dual_connector.rs, a small matrix over classes of connection
 * ids + an 8-wide raw part, built as the reference builds it except that ties of its greedy template choice -- which the
 * reference resolves by hash-set iteration order -- go to the highest template index).  The device image materialises either as
 * a dense matrix when the tokenizer is created: i16 cells when every cost fits, else i32 cells (the reference's Raw / Dual cost is
 * an i32 sum, raw_connector.rs:153-161, dual_connector.rs:267-279); VBT_ERR_UNSUPPORTED there only for a matrix of 4 GiB or more. */
VBT_API int vbt_dict_from_sources_bigram(const char* lex, size_t lex_len, const char* bigram_right, size_t right_len,
                                         const char* bigram_left, size_t left_len, const char* bigram_cost, size_t cost_len,
                                         const char* char_def, size_t char_len, const char* unk_def, size_t unk_len, int dual,
                                         vbt_dict** out);
/* Dictionary::read(rdr), dictionary.rs:173-197: a dictionary written by Dictionary::write ("VibratoTokenizer 0.5\n" + bincode,
 * common.rs:5-9).  A zstd frame around it -- the released `system.dic.zst`, which the reference's CLIs unwrap with
 * zstd::Decoder (tokenize/src/main.rs:59-60) -- is detected by its magic and decompressed first (libzstd.so.1 of the system).
 * VBT_ERR_INVALID_ARGUMENT: wrong magic (dictionary.rs:188-193); VBT_ERR_INVALID_FORMAT: anything that does not decode. */
VBT_API int vbt_dict_read(const uint8_t* data, size_t len, vbt_dict** out);
/* Dictionary::write(wtr), dictionary.rs:142-150.  zstd_level < 0: the plain `system.dic` bytes; otherwise wrapped in a zstd
 * frame of that level like compile/src/main.rs:98 (level 19 there).  *out is released with vbt_free. */
VBT_API int vbt_dict_write(const vbt_dict* dict, int zstd_level, uint8_t** out, size_t* len);
/* ConnectorWrapper variant, connector.rs:30-35: 0 Matrix, 1 Raw, 2 Dual (-1: null / consumed handle) */
VBT_API int vbt_dict_connector_kind(const vbt_dict* dict);
/* Dictionary::reset_user_lexicon_from_reader(Some(csv) | None), dictionary.rs:209-229 */
VBT_API int vbt_dict_set_user_lexicon(vbt_dict* dict, const char* csv, size_t len);
/* Dictionary::map_connection_ids_from_iter(lmap, rmap), dictionary.rs:245-259: the i-th item (1-origin) is the
 * old id mapped to new id i; id 0 (BOS/EOS) is fixed. Permutes lexicon params, unknown entries and the matrix. */
VBT_API int vbt_dict_map_connection_ids(vbt_dict* dict, const uint16_t* lmap, size_t n_lmap, const uint16_t* rmap, size_t n_rmap);
VBT_API void vbt_dict_free(vbt_dict* dict);

VBT_API uint32_t vbt_dict_num_words(const vbt_dict* dict, uint32_t lex_type);
VBT_API uint32_t vbt_dict_num_left(const vbt_dict* dict);  /* Connector::num_left,  connector.rs:14 */
VBT_API uint32_t vbt_dict_num_right(const vbt_dict* dict); /* Connector::num_right, connector.rs:17 */
/* Dictionary::word_feature, dictionary.rs:108-114 */
VBT_API int vbt_dict_word_feature(const vbt_dict* dict, uint32_t lex_type, uint32_t word_id, const char** ptr, size_t* len);
/* Dictionary::word_param, dictionary.rs:98-104: out = {left_id, right_id, word_cost} */
VBT_API int vbt_dict_word_param(const vbt_dict* dict, uint32_t lex_type, uint32_t word_id, int32_t out[3]);
/* ConnectorCost::cost(right_id, left_id), connector.rs:25-28 */
VBT_API int vbt_dict_conn_cost(const vbt_dict* dict, uint32_t right_id, uint32_t left_id, int32_t* out);
/* CharProperty::char_info, character.rs:112-116 (packed CharInfo u32, character.rs:10-24) */
VBT_API uint32_t vbt_dict_char_info(const vbt_dict* dict, uint32_t code_point);
/* CharProperty::cate_id, character.rs:119-124; -1 when undefined */
VBT_API int vbt_dict_cate_id(const vbt_dict* dict, const char* name, size_t len);
/* Lexicon::common_prefix_iterator, lexicon.rs:33-46 (host walk of the device trie image).
 * out rows = {word_id, end_char}; returns the number of matches (may exceed cap). */
VBT_API uint32_t vbt_dict_common_prefix(const vbt_dict* dict, uint32_t lex_type, const uint32_t* code_points,
                                        uint32_t n, uint32_t* out, uint32_t cap);

/* ---- Tokenizer ------------------------------------------------------------- */

/* Tokenizer::new(dict).ignore_space(b)?.max_grouping_len(n), tokenizer.rs:26-74.
 * Consumes `dict` on success (Tokenizer::new moves it). Uploads the device image to
 * HIP device `device` (-1 = current device). max_grouping_len 0 = unlimited. */
VBT_API int vbt_tokenizer_new(vbt_dict* dict, int ignore_space, uint32_t max_grouping_len, int device,
                              vbt_tokenizer** out);
/* The same over several GPUs of the node (new; SURVEY.md 8(b) `device_mask`, as an explicit list): one replica of the device
 * image per listed HIP device.  The reference leaves parallelism to the caller -- one Worker per thread over a shared &Tokenizer
 * (worker.rs:9-19, tokenizer.rs:82-84); here one tokenizer spans the devices and vbt_tokenize_batch splits every batch into
 * `n_devices` contiguous shards balanced by bytes, runs them side by side (a pooled workspace + stream per device) and has every
 * device's DMA engines write its shard of the results into the ONE pinned result block at the shard's offset: a gather to the
 * caller, no collective.  Results are identical to a single-device tokenizer's.  A device may be listed more than once (two
 * shards in flight on one GPU; how the path is tested on a one-GPU box).  Worker and the vbt_workspace_* device API use the
 * first listed device. */
VBT_API int vbt_tokenizer_new_multi(vbt_dict* dict, int ignore_space, uint32_t max_grouping_len, const int* devices,
                                    uint32_t n_devices, vbt_tokenizer** out);
VBT_API uint32_t vbt_tokenizer_num_devices(const vbt_tokenizer* tok);
/* Connection-id locality, built in (the reference's `reorder` + `map` workflow -- map/src/reorder.rs:34-63,
 * MatrixConnector::map_connection_ids matrix_connector.rs:99-116, Dictionary::map_connection_ids_from_iter dictionary.rs:245-259,
 * docs/map.md -- without a caller action): the first batch of at least VBT_CONNID_MIN_SENTENCES (2048) sentences a tokenizer sees
 * has up to VBT_CONNID_SAMPLE (16384) of its sentences, spread evenly over the batch, copied aside by two small kernels on the call's
 * stream -- the call itself stays asynchronous: nothing is allocated, synchronised or waited for on the caller's side -- and a
 * background thread sweeps that sample once more with the connection-id counters on, sorts the ids by count and publishes a device
 * image whose matrix rows / columns and entry id pairs are renumbered hot ids first; launches enqueued from then on read it.
 * Results are bit-identical (a pure permutation) and nothing visible is in device ids: Token::left_id / right_id, vbt_dict_conn_cost,
 * vbt_dict_write, the connection-id counters (vbt_workspace_connid_counts, vbt_worker_*connid*) all stay in the dictionary's
 * numbering, and a mapping applied with vbt_dict_map_connection_ids before the tokenizer was created is kept underneath.
 * VBT_CONNID_REORDER=0 turns it off.  out[8] = {epoch of the image in use (0 = the dictionary's numbering, 1 = renumbered),
 * state (0 waiting for a large batch, 1 running, 2 done, 3 off), sentences sampled, the minimum batch size, microseconds the
 * calibration took, left ids moved, right ids moved, 0}; of the tokenizer's first device. */
VBT_API int vbt_tokenizer_connid_reorder_info(const vbt_tokenizer* tok, uint64_t out[8]);
/* The same calibration up front and synchronously, from host text (`n` sentences, `offsets[n + 1]`; at most VBT_CONNID_SAMPLE of them,
 * spread evenly, are used), on every device of the tokenizer: for callers who do not want the image to change under their first
 * batches (map/src/reorder.rs:34-63 run as a step of its own).  A no-op once a calibration has happened or when it is switched off. */
VBT_API int vbt_tokenizer_calibrate(const vbt_tokenizer* tok, const uint8_t* text, const uint64_t* offsets, uint64_t n);
/* Returns once no calibration is running on any device of the tokenizer, or after timeout_ms (< 0: no limit); *idle = 1 when none
 * is running.  (Benchmarks wait here behind their warm-up so that the counting sweep does not share the GPU with the timed region.) */
VBT_API int vbt_tokenizer_connid_reorder_wait(const vbt_tokenizer* tok, int64_t timeout_ms, int* idle);
/* Lattice density of the text the tokenizer has been seeing: candidates (lattice nodes before the sweep) per input byte of the last batch
 * that reported, on the tokenizer's first device; 0 = no batch yet.  ~2.0 on running Japanese text over a unidic-sized lexicon, 4.4 on the
 * dense synthetic law.  The library picks the LDS tiers of the sweep by it (> 3 = 10 KiB segments, else 8 KiB ones and
 * one more wave per SIMD); exposed for observability.  No reference counterpart.  VBT_TIER_ADAPT=0: not measured, always 0. */
VBT_API int vbt_tokenizer_lattice_density(const vbt_tokenizer* tok, double* candidates_per_byte);
VBT_API void vbt_tokenizer_free(vbt_tokenizer* tok);
VBT_API const vbt_dict* vbt_tokenizer_dictionary(const vbt_tokenizer* tok); /* Tokenizer::dictionary, tokenizer.rs:77 */

/* ---- Worker (per-sentence API, worker.rs:34-75) --------------------------------- */

VBT_API int vbt_worker_new(const vbt_tokenizer* tok, vbt_worker** out);          /* Tokenizer::new_worker */
VBT_API void vbt_worker_free(vbt_worker* w);
VBT_API int vbt_worker_reset_sentence(vbt_worker* w, const char* utf8, size_t len); /* Worker::reset_sentence */
VBT_API int vbt_worker_tokenize(vbt_worker* w);                                   /* Worker::tokenize */
VBT_API uint32_t vbt_worker_num_tokens(const vbt_worker* w);                      /* Worker::num_tokens */
VBT_API int vbt_worker_token(const vbt_worker* w, uint32_t i, vbt_token* out);    /* Worker::token(i) */
/* vbt_worker_reset_sentence fails with VBT_ERR_UTF8 when the bytes are not a Rust `str` (the reference takes &str,
 * worker.rs:34); the worker then holds the empty sentence. */
/* vbt_worker_tokenize costs no launch in steady state: a resident one-wavefront kernel polls a doorbell in the worker's pinned host block,
 * reads the text out of it and writes the token records back into it (no copy engine, no allocation); the call returns when the
 * kernel's status word has landed.  The kernel leaves after ~2 ms without a call (VBT_WORKER_IDLE_POLLS) and, so that it never holds
 * the device against other threads' synchronising calls, after VBT_WORKER_MAX_SERVED (4096) sentences; the next call starts it again.  A kernel
 * that answers late (a busy GPU, a profiler) is waited for up to 5 s, then told to leave; only a failing stream is an error.
 * Sentences a single wavefront cannot take (longer than ~2500 characters, a dictionary word of more
 * than 64 characters, ...) and workers that count connection ids go through the batch pipeline instead.
 * vbt_worker_path_stats: sentences served by the single launch / by the batch pipeline so far. */
VBT_API int vbt_worker_path_stats(const vbt_worker* w, uint64_t* fast, uint64_t* slow);
/* The reference's calling pattern, timed inside the library (tokenize/src/main.rs:78-82, benchmark/src/main.rs:57-61): for each
 * of the n sentences reset_sentence -> tokenize -> num_tokens -> token(i) for every token, `rounds` passes; wall seconds and the
 * number of tokens seen. */
VBT_API int vbt_worker_loop_benchmark(vbt_worker* w, const uint8_t* text, const uint64_t* offsets, uint64_t n, uint32_t rounds,
                                      double* seconds, uint64_t* tokens);
/* Worker::init_connid_counter, worker.rs:77-84 (ConnIdCounter::new, mapper.rs:94-99): zeroed counters */
VBT_API int vbt_worker_init_connid_counter(vbt_worker* w);
/* Worker::update_connid_counts, worker.rs:86-93: adds Lattice::add_connid_counts (lattice.rs:170-183) of the last
 * vbt_worker_tokenize call.  VBT_ERR_INVALID_STATE where the reference panics (no init_connid_counter). */
VBT_API int vbt_worker_update_connid_counts(vbt_worker* w);
/* The raw counters: lid[num_left], rid[num_right]. */
VBT_API int vbt_worker_connid_counts(const vbt_worker* w, uint64_t* lid, uint64_t* rid);
/* Worker::compute_connid_probs, worker.rs:95-103 (ConnIdCounter::compute_probs, mapper.rs:108-146): per side the
 * (id, count / sum) pairs without id 0, sorted by probability descending, then id ascending -- the content of the
 * reference's *.lmap / *.rmap files.  lid_* hold num_left - 1 entries, rid_* num_right - 1. */
VBT_API int vbt_worker_compute_connid_probs(const vbt_worker* w, uint32_t* lid_ids, double* lid_probs, uint32_t* rid_ids,
                                            double* rid_probs);
/* The same computation for one side of any counter array (e.g. vbt_workspace_connid_counts): n - 1 entries out. */
VBT_API int vbt_connid_probs(const uint64_t* counts, size_t n, uint32_t* ids, double* probs);

/* ---- Batched host API (new: many sentences per call) ---------------------------- */

/* The 3-call loop of tokenize/src/main.rs:78-82 over n sentences: sentence s is
 * text[offsets[s] .. offsets[s+1]). Copies text to the device, runs the kernels,
 * copies token records back (the GPU's SDMA engines, into a pinned block sized to the result). The batch keeps its own copy of the text.
 * Every sentence must be valid UTF-8 (a Rust `str`): otherwise VBT_ERR_UTF8 and no batch.
 * ONE call is a pipeline (tokenize/src/main.rs:76-95 is a single-threaded caller): on a single-device tokenizer a lone caller's batch of
 * 4 MiB or more is cut at sentence borders into VBT_H2H_CHUNKS (default 5, 1 = off) chunks that rotate through three chunk-sized
 * workspaces -- the copy in and the copy out of neighbouring chunks run on the SDMA engines next to the kernels of the chunk in
 * between, into ONE result block (40 M sentences/s for one call from one thread on the headline workload; 20 M unpipelined).  The
 * tokenizer's first batch runs unpipelined (it sets the tokens-per-KiB estimate the result block of a pipelined call is sized by; a
 * batch that outgrows the estimate is redone unpipelined).  Results are the same arrays either way.
 * Thread-safe per tokenizer: each call takes workspaces (device scratch + staging + stream) from the tokenizer's
 * pool and returns them, so steady-state calls do no device allocation (vbt_tokenizer_pool_stats); batches pushed from
 * several host threads overlap their copies with each other's kernels (4-8 threads reach 62-66 M sentences/s; such calls are not
 * cut into chunks: only a caller that has been alone for its last two calls is).
 * The pools keep idle workspaces (~400 B of device memory per byte of text of the batch they were sized for) and pinned
 * blocks for reuse, at most VBT_POOL_MAX_MB=<device MB>[,<pinned MB>] of each (default: a quarter of the GPU's memory -- 72 GiB on
 * an MI355X -- and 8192 MB of pinned memory; what would exceed it is released instead of pooled, and a caller whose workspace
 * never fits re-allocates it on every call); vbt_tokenizer_trim_pool releases everything idle now. */
VBT_API int vbt_tokenize_batch(const vbt_tokenizer* tok, const uint8_t* text, const uint64_t* offsets, uint64_t n,
                               vbt_batch** out);
VBT_API int vbt_tokenizer_trim_pool(const vbt_tokenizer* tok);
VBT_API int vbt_tokenizer_pool_stats(const vbt_tokenizer* tok, uint64_t* created, uint64_t* reused, uint64_t* idle);
VBT_API void vbt_batch_free(vbt_batch* b);
VBT_API uint64_t vbt_batch_num_sentences(const vbt_batch* b);
VBT_API uint64_t vbt_batch_total_tokens(const vbt_batch* b);
VBT_API uint32_t vbt_batch_num_tokens(const vbt_batch* b, uint64_t sentence);
VBT_API int vbt_batch_token(const vbt_batch* b, uint64_t sentence, uint32_t i, vbt_token* out);
/* Raw records of one sentence (num_tokens of them, in sentence order). */
VBT_API const vbt_token_rec* vbt_batch_records(const vbt_batch* b, uint64_t sentence);
/* Whole-batch views: sentence s owns tokens[tok_off[s] .. tok_off[s] + tok_cnt[s]). */
VBT_API int vbt_batch_arrays(const vbt_batch* b, const vbt_token_rec** tokens, const uint32_t** tok_off,
                             const uint32_t** tok_cnt);
/* Byte-identical output of the `tokenize` CLI for the whole batch (tokenize/src/main.rs:83-127).
 * *out belongs to the library: release it ONLY with vbt_free (never free(): the pointer does not start a malloc block).  Rendered by up
 * to VBT_FORMAT_THREADS host threads (default: the host's cores, at most 64) in two passes over chunks of sentences: exact sizes,
 * then every chunk in place. */
VBT_API int vbt_batch_format(const vbt_batch* b, int mode, char** out, size_t* len);
/* Releases a buffer handed out by vbt_batch_format / vbt_dict_write.  A pointer that is not a live buffer of this library (freed
 * already, or from somewhere else) is ignored.  Big buffers are kept for the next call (at most 4 / 512 MiB per process) until
 * vbt_tokenizer_trim_pool or vbt_tokenizer_free. */
VBT_API void vbt_free(void* p);

/* ---- Device-resident API (zero-copy; what bench.py times) ------------------------ */

/* A workspace owns the device scratch + output buffers for batches of up to
 * max_sentences sentences / max_bytes bytes of text, and is reused across calls.
 * Footprint: about 400 bytes of device memory per byte of text (candidate records and staged trie hits at 8 slots of
 * 16 bytes per input byte each, token staging + compact tokens at 24 bytes each, per-character records, the fused
 * fallback's scratch) plus ~7 KB per sentence (head room of its node region): ~6.5 GB for the 14 MB / 100k-sentence headline batch, and at most
 * ~0.6 GB of text per workspace on a 288 GB part -- larger corpora are fed as a sequence of batches (the host entry
 * point vbt_tokenize_batch refuses a batch of 4 GiB or more outright; VBT_ERR_DEVICE when the allocation fails). */
VBT_API int vbt_workspace_new(const vbt_tokenizer* tok, uint64_t max_sentences, uint64_t max_bytes, vbt_workspace** out);
VBT_API void vbt_workspace_free(vbt_workspace* ws);
/* Enqueue the tokenization of n sentences on `hip_stream` (a hipStream_t, NULL = default
 * stream). d_text / d_offsets are DEVICE pointers (offsets: n+1 x u64, bytes into d_text;
 * offsets[0] need not be 0: the batch may be a window into a larger text buffer).
 * Asynchronous; results are valid once the stream has been synchronized.
 * Input contract, checked on the device by the first kernel of the call: offsets non-decreasing,
 * offsets[n] - offsets[0] <= total_bytes (<= the workspace's max_bytes, checked on the host), text valid UTF-8
 * with every sentence starting on a character boundary.  A violation skips the batch and shows up as
 * error_flags 8 (offsets) / 16 (UTF-8) in vbt_workspace_stats; results of that call are undefined. */
VBT_API int vbt_tokenize_batch_device(vbt_workspace* ws, const uint8_t* d_text, const uint64_t* d_offsets, uint64_t n,
                                      uint64_t total_bytes, void* hip_stream);
/* Device pointers to the results of the last call:
 *   tokens   : vbt_token_rec[], sentence s occupies [tok_off[s], tok_off[s] + tok_cnt[s])
 *   tok_off  : u32[n]  (allocation order is unspecified; content is deterministic)
 *   tok_cnt  : u32[n]
 *   total    : u32[1]  total tokens written */
VBT_API int vbt_workspace_results(const vbt_workspace* ws, const vbt_token_rec** d_tokens, const uint32_t** d_tok_off,
                                  const uint32_t** d_tok_cnt, const uint32_t** d_total);
/* The final exchange of a multi-GPU job without copy kernels: every later vbt_tokenize_batch_device call of this workspace leaves
 * its results in ONE caller-owned device buffer laid out as a rank's slot of the gather (vibrato_amd/sharding.py; north_star:
 * "RCCL over xGMI only for the final gather"):
 *   [32-byte header {n_sentences u64, n_tokens u32, error flags u32, 0 x 16 bytes}] [tok_off u32 x max_sentences] [tok_cnt u32 x max_sentences] [vbt_token_rec x n_tokens]
 * -- tok_cnt written by the sweep, tok_off and the records by the packing kernel, the header with its total -- so the collective
 * (ncclSend / all-gather of the slot) starts from what the tokenizer wrote.  d_slot: 8-byte aligned device memory of slot_bytes >=
 * 32 + 8 max_sentences + 24 x (tokens of the largest batch); a batch whose tokens do not fit sets error flag 1 -- in the workspace's
 * statistics AND in the header's flags word, whose n_tokens is then the number of records the slot holds (a consumer of gathered slots
 * sees nothing else of the rank that wrote one).  d_slot = NULL:
 * back to the workspace's own buffers.  vbt_workspace_results then points into the slot.  The caller alternates two slots to
 * overlap the gather of batch k with the kernels of batch k + 1. */
VBT_API int vbt_workspace_set_packed_output(vbt_workspace* ws, void* d_slot, uint64_t slot_bytes, uint64_t max_sentences);
/* Per-call statistics (synchronizes the stream of the last call): how many sentences were filed
 * in the LDS tiers sentences are routed to up front -- the lean tier and the segment tier (n_tier0) --, in the escape tiers behind
 * them (n_tier1: what a sweep passed on) and in the list of the global-memory fallback kernel (n_tier2; a re-routed sentence is
 * counted in every list it was filed in), tokens
 * written, device error flags (1 = token buffer full, 2 = scratch exhausted, 4 = sentence too
 * long, 8 = bad offsets, 16 = invalid UTF-8) and, with timing enabled, the hipEvent-measured durations (ms, on the launch stream) of the
 * input check and the candidate generators (ms_tier0: validate_batch, gen_candidates, build_lists, the gen_long levels), of the
 * lattice sweeps from their fork to their join (ms_tier12) and of what follows -- the global-memory fallback launch and the packing
 * of the token records (ms_pack). */
typedef struct vbt_call_stats {
    uint64_t n_sentences, n_tier0, n_tier1, n_tier2, n_tokens;
    uint32_t error_flags;
    float ms_tier0, ms_tier12, ms_pack;
} vbt_call_stats;
VBT_API int vbt_workspace_set_timing(vbt_workspace* ws, int enabled);
/* Worker::init_connid_counter / update_connid_counts (worker.rs:77-93, Lattice::add_connid_counts lattice.rs:170-183):
 * while enabled, every batch adds, for each adjacent (left node, right node) pair of each lattice, 1 to
 * lid[right_node.left_id] and rid[left_node.right_id] (every sentence exactly once, whichever kernel ends up sweeping
 * it).  vbt_workspace_connid_counts copies the totals (num_left / num_right u64). */
VBT_API int vbt_workspace_count_connids(vbt_workspace* ws, int enabled);
VBT_API int vbt_workspace_connid_counts(vbt_workspace* ws, uint64_t* lid, uint64_t* rid, int reset);
/* Developer aid: per-phase shader-clock cycles summed over all sentences since the last reset
 * (enabled by VBT_PROFILE=1 in the environment when the workspace is created). out[0..7] =
 * decode, count, fill, end lists, pre-pass, pass records, recurrence, emit; out[8] = sentences,
 * out[9] = lattice steps, out[10] = lattice passes, out[11] = lattice candidates. */
VBT_API int vbt_workspace_profile(vbt_workspace* ws, uint64_t out[12], int reset);
VBT_API int vbt_workspace_stats(vbt_workspace* ws, vbt_call_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* VIBRATO_HIP_H */
