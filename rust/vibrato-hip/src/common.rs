//! Common settings (reference: `vibrato/src/common.rs`).

/// The maximum length of an input sentence in bytes for one device batch entry (the reference allows `usize::MAX`;
/// the device path handles sentences below 4 GiB, longer ones are rejected with an error).
pub const MAX_SENTENCE_LENGTH: usize = u32::MAX as usize;

/// The fixed connection id of BOS/EOS (`common.rs:18`).
pub const BOS_EOS_CONNECTION_ID: u16 = 0;
