//! Raw bindings of `include/vibrato_hip.h` (one declaration per `VBT_API` entry point, same order).
//! Reference items each entry point replaces are cited in the header.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

macro_rules! opaque {
    ($($name:ident),*) => { $(#[repr(C)] pub struct $name { _private: [u8; 0] })* };
}
opaque!(vbt_dict, vbt_tokenizer, vbt_worker, vbt_batch, vbt_workspace);

pub const VBT_OK: c_int = 0;
pub const VBT_ERR_INVALID_ARGUMENT: c_int = 1;
pub const VBT_ERR_INVALID_FORMAT: c_int = 2;
pub const VBT_ERR_INVALID_STATE: c_int = 3;
pub const VBT_ERR_PARSE_INT: c_int = 4;
pub const VBT_ERR_UTF8: c_int = 5;
pub const VBT_ERR_DEVICE: c_int = 100;
pub const VBT_ERR_UNSUPPORTED: c_int = 101;

pub const VBT_LEX_SYSTEM: u32 = 0;
pub const VBT_LEX_USER: u32 = 1;
pub const VBT_LEX_UNKNOWN: u32 = 2;

pub const VBT_FORMAT_MECAB: c_int = 0;
pub const VBT_FORMAT_WAKATI: c_int = 1;
pub const VBT_FORMAT_DETAIL: c_int = 2;

/// 24-byte record the device writes per best-path node.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct vbt_token_rec {
    pub start_char: u32,
    pub end_char: u32,
    pub start_byte: u32,
    pub end_byte: u32,
    /// `lex_type << 30 | word_id`
    pub word_idx: u32,
    pub total_cost: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct vbt_token {
    pub surface: *const c_char,
    pub surface_len: usize,
    pub feature: *const c_char,
    pub feature_len: usize,
    pub start_char: u32,
    pub end_char: u32,
    pub start_byte: u32,
    pub end_byte: u32,
    pub lex_type: u32,
    pub word_id: u32,
    pub left_id: u16,
    pub right_id: u16,
    pub word_cost: i16,
    pub total_cost: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct vbt_call_stats {
    pub n_sentences: u64,
    pub n_tier0: u64,
    pub n_tier1: u64,
    pub n_tier2: u64,
    pub n_tokens: u64,
    pub error_flags: u32,
    pub ms_tier0: f32,
    pub ms_tier12: f32,
    pub ms_pack: f32,
}

extern "C" {
    pub fn vbt_last_error() -> *const c_char;
    pub fn vbt_utf8_valid(utf8: *const c_char, len: usize) -> c_int;

    // ---- Dictionary
    pub fn vbt_dict_from_sources(lex: *const c_char, lex_len: usize, matrix_def: *const c_char, matrix_len: usize,
                                 char_def: *const c_char, char_len: usize, unk_def: *const c_char, unk_len: usize,
                                 out: *mut *mut vbt_dict) -> c_int;
    pub fn vbt_dict_from_sources_binmatrix(lex: *const c_char, lex_len: usize, matrix: *const i16, num_right: u32, num_left: u32,
                                           char_def: *const c_char, char_len: usize, unk_def: *const c_char, unk_len: usize,
                                           out: *mut *mut vbt_dict) -> c_int;
    pub fn vbt_dict_from_sources_bigram(lex: *const c_char, lex_len: usize, bigram_right: *const c_char, right_len: usize,
                                        bigram_left: *const c_char, left_len: usize, bigram_cost: *const c_char, cost_len: usize,
                                        char_def: *const c_char, char_len: usize, unk_def: *const c_char, unk_len: usize,
                                        dual: c_int, out: *mut *mut vbt_dict) -> c_int;
    pub fn vbt_dict_read(data: *const u8, len: usize, out: *mut *mut vbt_dict) -> c_int;
    pub fn vbt_dict_write(dict: *const vbt_dict, zstd_level: c_int, out: *mut *mut u8, len: *mut usize) -> c_int;
    pub fn vbt_dict_connector_kind(dict: *const vbt_dict) -> c_int;
    pub fn vbt_dict_set_user_lexicon(dict: *mut vbt_dict, csv: *const c_char, len: usize) -> c_int;
    pub fn vbt_dict_map_connection_ids(dict: *mut vbt_dict, lmap: *const u16, n_lmap: usize, rmap: *const u16, n_rmap: usize) -> c_int;
    pub fn vbt_dict_free(dict: *mut vbt_dict);
    pub fn vbt_dict_num_words(dict: *const vbt_dict, lex_type: u32) -> u32;
    pub fn vbt_dict_num_left(dict: *const vbt_dict) -> u32;
    pub fn vbt_dict_num_right(dict: *const vbt_dict) -> u32;
    pub fn vbt_dict_word_feature(dict: *const vbt_dict, lex_type: u32, word_id: u32, ptr: *mut *const c_char, len: *mut usize) -> c_int;
    pub fn vbt_dict_word_param(dict: *const vbt_dict, lex_type: u32, word_id: u32, out: *mut i32) -> c_int;
    pub fn vbt_dict_conn_cost(dict: *const vbt_dict, right_id: u32, left_id: u32, out: *mut i32) -> c_int;
    pub fn vbt_dict_char_info(dict: *const vbt_dict, code_point: u32) -> u32;
    pub fn vbt_dict_cate_id(dict: *const vbt_dict, name: *const c_char, len: usize) -> c_int;
    pub fn vbt_dict_common_prefix(dict: *const vbt_dict, lex_type: u32, code_points: *const u32, n: u32, out: *mut u32, cap: u32) -> u32;

    // ---- Tokenizer
    pub fn vbt_tokenizer_new(dict: *mut vbt_dict, ignore_space: c_int, max_grouping_len: u32, device: c_int,
                             out: *mut *mut vbt_tokenizer) -> c_int;
    pub fn vbt_tokenizer_new_multi(dict: *mut vbt_dict, ignore_space: c_int, max_grouping_len: u32, devices: *const c_int, n_devices: u32,
                                   out: *mut *mut vbt_tokenizer) -> c_int;
    pub fn vbt_tokenizer_num_devices(tok: *const vbt_tokenizer) -> u32;
    pub fn vbt_tokenizer_connid_reorder_info(tok: *const vbt_tokenizer, out: *mut u64) -> c_int;
    pub fn vbt_tokenizer_calibrate(tok: *const vbt_tokenizer, text: *const u8, offsets: *const u64, n: u64) -> c_int;
    pub fn vbt_tokenizer_connid_reorder_wait(tok: *const vbt_tokenizer, timeout_ms: i64, idle: *mut c_int) -> c_int;
    pub fn vbt_tokenizer_lattice_density(tok: *const vbt_tokenizer, candidates_per_byte: *mut f64) -> c_int;
    pub fn vbt_tokenizer_free(tok: *mut vbt_tokenizer);
    pub fn vbt_tokenizer_dictionary(tok: *const vbt_tokenizer) -> *const vbt_dict;

    // ---- Worker
    pub fn vbt_tokenizer_trim_pool(tok: *const vbt_tokenizer) -> c_int;
    pub fn vbt_worker_new(tok: *const vbt_tokenizer, out: *mut *mut vbt_worker) -> c_int;
    pub fn vbt_worker_free(w: *mut vbt_worker);
    pub fn vbt_worker_reset_sentence(w: *mut vbt_worker, utf8: *const c_char, len: usize) -> c_int;
    pub fn vbt_worker_tokenize(w: *mut vbt_worker) -> c_int;
    pub fn vbt_worker_num_tokens(w: *const vbt_worker) -> u32;
    pub fn vbt_worker_token(w: *const vbt_worker, i: u32, out: *mut vbt_token) -> c_int;
    pub fn vbt_worker_path_stats(w: *const vbt_worker, fast: *mut u64, slow: *mut u64) -> c_int;
    pub fn vbt_worker_loop_benchmark(w: *mut vbt_worker, text: *const u8, offsets: *const u64, n: u64, rounds: u32, seconds: *mut f64,
                                     tokens: *mut u64) -> c_int;
    pub fn vbt_worker_init_connid_counter(w: *mut vbt_worker) -> c_int;
    pub fn vbt_worker_update_connid_counts(w: *mut vbt_worker) -> c_int;
    pub fn vbt_worker_connid_counts(w: *const vbt_worker, lid: *mut u64, rid: *mut u64) -> c_int;
    pub fn vbt_worker_compute_connid_probs(w: *const vbt_worker, lid_ids: *mut u32, lid_probs: *mut f64, rid_ids: *mut u32,
                                           rid_probs: *mut f64) -> c_int;
    pub fn vbt_connid_probs(counts: *const u64, n: usize, ids: *mut u32, probs: *mut f64) -> c_int;

    // ---- Batched host API
    pub fn vbt_tokenize_batch(tok: *const vbt_tokenizer, text: *const u8, offsets: *const u64, n: u64, out: *mut *mut vbt_batch) -> c_int;
    pub fn vbt_tokenizer_pool_stats(tok: *const vbt_tokenizer, created: *mut u64, reused: *mut u64, idle: *mut u64) -> c_int;
    pub fn vbt_batch_free(b: *mut vbt_batch);
    pub fn vbt_batch_num_sentences(b: *const vbt_batch) -> u64;
    pub fn vbt_batch_total_tokens(b: *const vbt_batch) -> u64;
    pub fn vbt_batch_num_tokens(b: *const vbt_batch, sentence: u64) -> u32;
    pub fn vbt_batch_token(b: *const vbt_batch, sentence: u64, i: u32, out: *mut vbt_token) -> c_int;
    pub fn vbt_batch_records(b: *const vbt_batch, sentence: u64) -> *const vbt_token_rec;
    pub fn vbt_batch_arrays(b: *const vbt_batch, tokens: *mut *const vbt_token_rec, tok_off: *mut *const u32, tok_cnt: *mut *const u32) -> c_int;
    pub fn vbt_batch_format(b: *const vbt_batch, mode: c_int, out: *mut *mut c_char, len: *mut usize) -> c_int;
    pub fn vbt_free(p: *mut c_void);

    // ---- Device-resident API
    pub fn vbt_workspace_new(tok: *const vbt_tokenizer, max_sentences: u64, max_bytes: u64, out: *mut *mut vbt_workspace) -> c_int;
    pub fn vbt_workspace_free(ws: *mut vbt_workspace);
    pub fn vbt_tokenize_batch_device(ws: *mut vbt_workspace, d_text: *const u8, d_offsets: *const u64, n: u64, total_bytes: u64,
                                     hip_stream: *mut c_void) -> c_int;
    pub fn vbt_workspace_results(ws: *const vbt_workspace, d_tokens: *mut *const vbt_token_rec, d_tok_off: *mut *const u32,
                                 d_tok_cnt: *mut *const u32, d_total: *mut *const u32) -> c_int;
    pub fn vbt_workspace_set_packed_output(ws: *mut vbt_workspace, d_slot: *mut c_void, slot_bytes: u64, max_sentences: u64) -> c_int;
    pub fn vbt_workspace_set_timing(ws: *mut vbt_workspace, enabled: c_int) -> c_int;
    pub fn vbt_workspace_count_connids(ws: *mut vbt_workspace, enabled: c_int) -> c_int;
    pub fn vbt_workspace_connid_counts(ws: *mut vbt_workspace, lid: *mut u64, rid: *mut u64, reset: c_int) -> c_int;
    pub fn vbt_workspace_profile(ws: *mut vbt_workspace, out: *mut u64, reset: c_int) -> c_int;
    pub fn vbt_workspace_stats(ws: *mut vbt_workspace, out: *mut vbt_call_stats) -> c_int;
}
