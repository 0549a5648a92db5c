//! # vibrato over HIP
//!
//! The public API of crate `vibrato` for the `tokenize()` path (reference: `vibrato/src/lib.rs:52-78`), implemented over
//! the C ABI of `libvibrato_hip.so` (`include/vibrato_hip.h`): the lattice construction, Viterbi search and connection
//! lookups run as HIP kernels on an MI355X; there is no CPU fallback.
//!
//! ```no_run
//! use std::fs::File;
//! use vibrato::{Dictionary, Tokenizer};
//!
//! let dict = Dictionary::read(File::open("system.dic.zst")?)?; // zstd frames are unwrapped
//! let tokenizer = Tokenizer::new(dict).ignore_space(true)?.max_grouping_len(24);
//! let mut worker = tokenizer.new_worker();
//! worker.reset_sentence("京都東京都");
//! worker.tokenize();
//! for t in worker.token_iter() {
//!     println!("{}\t{}", t.surface(), t.feature());
//! }
//! // throughput path: many sentences per call
//! let batch = tokenizer.tokenize_batch(["京都東京都", "外国人参政権"])?;
//! print!("{}", batch.format(vibrato::batch::OutputMode::Mecab)?);
//! # Ok::<(), Box<dyn std::error::Error>>(())
//! ```
//!
//! Out of scope of this crate (unchanged in the reference): training (`trainer`, `mecab`), `Sentence` internals.
pub mod batch;
pub mod common;
pub mod dictionary;
pub mod errors;
pub mod token;
pub mod tokenizer;

pub use dictionary::{Dictionary, SystemDictionaryBuilder};
pub use tokenizer::{BatchSentence, LineBatches, Tokenizer};

/// Version number of this library (the reference API version it mirrors).
pub const VERSION: &str = env!("CARGO_PKG_VERSION");
