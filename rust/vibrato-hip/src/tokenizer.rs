//! Viterbi-based tokenizer (reference: `vibrato/src/tokenizer.rs:13-92`).
pub mod worker;

use std::ptr;
use std::sync::OnceLock;

use vibrato_hip_sys as sys;

use crate::batch::Batch;
use crate::dictionary::Dictionary;
use crate::errors::{check, Result, VibratoError};
use crate::tokenizer::worker::Worker;

struct Device(*mut sys::vbt_tokenizer);
// Safety: vbt_tokenizer is immutable after creation apart from its internally locked workspace pool (SURVEY.md 8(b)
// "threading"): concurrent vbt_worker_new / vbt_tokenize_batch calls on one handle are supported by the library.
unsafe impl Send for Device {}
unsafe impl Sync for Device {}

/// Tokenizer. `new`, `ignore_space` and `max_grouping_len` only record options (as in the reference, `tokenizer.rs:26-74`);
/// the device image (trie, connection matrix, ...) is uploaded once, by the first `new_worker` / `tokenize_batch` call.
pub struct Tokenizer {
    dict: Dictionary,
    ignore_space: bool,
    max_grouping_len: usize,
    device_index: i32,
    device_list: Vec<i32>,
    device: OnceLock<Device>,
}

impl Tokenizer {
    /// Creates a new instance (`tokenizer.rs:26-33`); the dictionary is moved in.
    pub const fn new(dict: Dictionary) -> Self {
        Self { dict, ignore_space: false, max_grouping_len: 0, device_index: -1, device_list: Vec::new(), device: OnceLock::new() }
    }

    /// Enables MeCab compatible mode: ignores spaces (`tokenizer.rs:42-58`).
    ///
    /// # Errors
    ///
    /// [`VibratoError`] is returned when category `SPACE` is not defined in the input dictionary.
    pub fn ignore_space(mut self, yes: bool) -> Result<Self> {
        if yes && !self.dict.has_category("SPACE") {
            return Err(VibratoError::invalid_argument("dict", "SPACE is not defined in the input dictionary (i.e., char.def)."));
        }
        self.ignore_space = yes;
        Ok(self)
    }

    /// Specifies the maximum grouping length for unknown words; `0` = infinity (`tokenizer.rs:67-74`). `24` gives MeCab's results.
    pub fn max_grouping_len(mut self, max_grouping_len: usize) -> Self {
        self.max_grouping_len = max_grouping_len;
        self
    }

    /// HIP device that will hold the dictionary image (new; default: the current device). One tokenizer per GPU.
    pub fn device(mut self, index: i32) -> Self {
        self.device_index = index;
        self
    }

    /// Several GPUs of the node (new): one replica of the dictionary image per listed HIP device; `tokenize_batch` splits every
    /// batch into contiguous shards balanced by bytes, runs them side by side and gathers the results into one host block. The
    /// reference leaves parallelism to the caller (one `Worker` per thread, `worker.rs:9-19`); this is the same for one batch call.
    /// Workers use the first listed device.
    pub fn devices(mut self, indices: &[i32]) -> Self {
        self.device_list = indices.to_vec();
        self
    }

    /// Number of devices the batches of this tokenizer are split over.
    pub fn num_devices(&self) -> usize {
        self.device_list.len().max(1)
    }

    /// Gets the reference to the dictionary (`tokenizer.rs:77-79`).
    pub const fn dictionary(&self) -> &Dictionary {
        &self.dict
    }

    /// Uploads the device image if that has not happened yet. `new_worker` calls this and panics on failure (it is infallible in
    /// the reference); call it directly to handle a missing device as an error.
    pub fn build_device_image(&self) -> Result<()> {
        self.raw().map(|_| ())
    }

    pub(crate) fn raw(&self) -> Result<*mut sys::vbt_tokenizer> {
        if let Some(d) = self.device.get() {
            return Ok(d.0);
        }
        // std's OnceLock has no fallible initialiser on stable: serialise the build so that the C handle is consumed once
        static BUILD: std::sync::Mutex<()> = std::sync::Mutex::new(());
        let _guard = BUILD.lock().unwrap();
        if let Some(d) = self.device.get() {
            return Ok(d.0);
        }
        let mgl = u32::try_from(self.max_grouping_len).unwrap_or(0); // lengths beyond u32 never limit anything
        let mut raw = ptr::null_mut();
        // Safety: on success the library takes the dictionary handle over (Tokenizer::new moves it); on failure we keep it.
        if self.device_list.is_empty() {
            check(unsafe { sys::vbt_tokenizer_new(self.dict.raw(), self.ignore_space as i32, mgl, self.device_index, &mut raw) })?;
        } else {
            let n = u32::try_from(self.device_list.len()).map_err(|_| VibratoError::invalid_argument("devices", "too many devices"))?;
            check(unsafe {
                sys::vbt_tokenizer_new_multi(self.dict.raw(), self.ignore_space as i32, mgl, self.device_list.as_ptr(), n, &mut raw)
            })?;
        }
        self.dict.rebind_borrowed(unsafe { sys::vbt_tokenizer_dictionary(raw) });
        let _ = self.device.set(Device(raw));
        Ok(raw)
    }

    /// Creates a new worker (`tokenizer.rs:82-84`).
    ///
    /// # Panics
    ///
    /// When no gfx950 device is available or the upload fails (there is no CPU fallback).
    pub fn new_worker(&self) -> Worker<'_> {
        Worker::new(self)
    }

    /// Tokenizes many sentences in one device batch (new): the loop `reset_sentence; tokenize; token(i)...` of
    /// `tokenize/src/main.rs:78-82` for every sentence, as a handful of kernel launches. This is the throughput path.
    pub fn tokenize_batch<I, S>(&self, sentences: I) -> Result<Batch<'_>>
    where
        I: IntoIterator<Item = S>,
        S: AsRef<str>,
    {
        let mut text = Vec::new();
        let mut offsets = vec![0u64];
        for s in sentences {
            text.extend_from_slice(s.as_ref().as_bytes());
            offsets.push(text.len() as u64);
        }
        self.tokenize_batch_raw(&text, &offsets)
    }

    /// The same on concatenated UTF-8 text: sentence `s` is `text[offsets[s]..offsets[s + 1]]`.
    pub fn tokenize_batch_raw(&self, text: &[u8], offsets: &[u64]) -> Result<Batch<'_>> {
        if offsets.is_empty() || *offsets.last().unwrap() as usize > text.len() {
            return Err(VibratoError::invalid_argument("offsets", "offsets must hold n + 1 positions inside text"));
        }
        let mut raw = ptr::null_mut();
        check(unsafe { sys::vbt_tokenize_batch(self.raw()?, text.as_ptr(), offsets.as_ptr(), (offsets.len() - 1) as u64, &mut raw) })?;
        Ok(Batch::from_raw(raw, self))
    }

    /// The per-line loop of the reference's callers (`tokenize/src/main.rs:78-95`) over an iterator of lines, batched behind the
    /// scenes (new): lines are collected until `batch_bytes` of text (or `batch_lines` lines) are in hand, tokenized as ONE device
    /// batch, and handed back one sentence at a time in input order. `Worker::tokenize` costs ~35 us per call on the GPU (one
    /// wavefront walks the whole chain alone); a batch of 100 k lines costs ~17 ns per line -- a caller that has more than a
    /// handful of lines in hand wants this, not a `Worker`. An `Err` item ends the iteration (an invalid line fails its whole batch).
    ///
    /// ```ignore
    /// for sent in tokenizer.tokenize_lines(stdin.lock().lines().map(|l| l.unwrap()), 16 << 20, 100_000) {
    ///     let sent = sent?;
    ///     for t in sent.tokens() { println!("{}\t{}", t.surface(), t.feature()); }
    ///     println!("EOS");
    /// }
    /// ```
    pub fn tokenize_lines<I, S>(&self, lines: I, batch_bytes: usize, batch_lines: usize) -> LineBatches<'_, I::IntoIter>
    where
        I: IntoIterator<Item = S>,
        S: AsRef<str>,
    {
        LineBatches { tokenizer: self, lines: lines.into_iter(), batch_bytes: batch_bytes.max(1), batch_lines: batch_lines.max(1), current: None, next: 0, failed: false }
    }

    /// The internal renumbering of the connection ids by measured usage (the reference's `reorder` + `map` workflow,
    /// `map/src/reorder.rs:34-63`, applied to the DEVICE image only: nothing visible changes), done up front and synchronously on a
    /// sample of `lines` (at most 16 384 of them, spread evenly).  Without this call the tokenizer does the same in the background behind
    /// its first batch of 2048 or more sentences; a second call is a no-op.
    pub fn calibrate<I, S>(&self, lines: I) -> Result<()>
    where
        I: IntoIterator<Item = S>,
        S: AsRef<str>,
    {
        let mut text: Vec<u8> = Vec::new();
        let mut offsets: Vec<u64> = vec![0];
        for l in lines {
            text.extend_from_slice(l.as_ref().as_bytes());
            offsets.push(text.len() as u64);
        }
        let tok = self.raw()?;
        // Safety: `text` / `offsets` outlive the call; the library copies what it samples.
        check(unsafe { sys::vbt_tokenizer_calibrate(tok, text.as_ptr(), offsets.as_ptr(), (offsets.len() - 1) as u64) })
    }

    /// Blocks while a background calibration is running (`timeout_ms < 0`: no limit); `Ok(true)` when none is running any more.
    pub fn wait_for_calibration(&self, timeout_ms: i64) -> Result<bool> {
        let tok = self.raw()?;
        let mut idle: i32 = 0;
        check(unsafe { sys::vbt_tokenizer_connid_reorder_wait(tok, timeout_ms, &mut idle) })?;
        Ok(idle != 0)
    }

    /// Candidates (lattice nodes) per input byte of the last batch that reported: what the library picks the sweep's LDS tiers by
    /// (`vbt_tokenizer_lattice_density`); 0.0 before the first batch.
    pub fn lattice_density(&self) -> Result<f64> {
        let tok = self.raw()?;
        let mut d: f64 = 0.0;
        check(unsafe { sys::vbt_tokenizer_lattice_density(tok, &mut d) })?;
        Ok(d)
    }

    /// Releases the idle device workspaces and pinned blocks `tokenize_batch` keeps for reuse (about 400 bytes of device memory per
    /// byte of text of every batch that was in flight at once; at most a quarter of the GPU's memory, `VBT_POOL_MAX_MB`). They are
    /// created again on demand. Thread-safe; a no-op before the first batch.
    pub fn trim_pool(&self) -> Result<()> {
        match self.device.get() {
            Some(d) => check(unsafe { sys::vbt_tokenizer_trim_pool(d.0) }),
            None => Ok(()),
        }
    }
}

impl Drop for Tokenizer {
    fn drop(&mut self) {
        if let Some(d) = self.device.take() {
            // Safety: workers and batches borrow `self`, so none is alive; the borrowed dictionary view dies with the handle.
            unsafe { sys::vbt_tokenizer_free(d.0) };
        }
    }
}

/// Iterator returned by [`Tokenizer::tokenize_lines`]: the sentences of the input, in order, each with its tokens.
pub struct LineBatches<'t, I> {
    tokenizer: &'t Tokenizer,
    lines: I,
    batch_bytes: usize,
    batch_lines: usize,
    current: Option<std::rc::Rc<Batch<'t>>>,
    next: usize,
    failed: bool,
}

/// One sentence of a batch behind [`LineBatches`]; keeps its batch alive.
pub struct BatchSentence<'t> {
    batch: std::rc::Rc<Batch<'t>>,
    index: usize,
}

impl<'t> BatchSentence<'t> {
    /// `Worker::num_tokens`.
    pub fn num_tokens(&self) -> usize {
        self.batch.num_tokens(self.index)
    }

    /// `Worker::token(i)`.
    pub fn token<'b>(&'b self, i: usize) -> crate::token::Token<'b, 't> {
        self.batch.token(self.index, i)
    }

    /// `Worker::token_iter`.
    pub fn tokens<'b>(&'b self) -> impl Iterator<Item = crate::token::Token<'b, 't>> + 'b {
        (0..self.num_tokens()).map(move |i| self.token(i))
    }
}

impl<'t, I, S> Iterator for LineBatches<'t, I>
where
    I: Iterator<Item = S>,
    S: AsRef<str>,
{
    type Item = Result<BatchSentence<'t>>;

    fn next(&mut self) -> Option<Self::Item> {
        if self.failed {
            return None;
        }
        if let Some(b) = &self.current {
            if self.next < b.len() {
                self.next += 1;
                return Some(Ok(BatchSentence { batch: b.clone(), index: self.next - 1 }));
            }
            self.current = None;
        }
        // collect the next batch: concatenated text + n + 1 offsets, as vbt_tokenize_batch takes them
        let mut text = Vec::new();
        let mut offsets = vec![0u64];
        while text.len() < self.batch_bytes && offsets.len() <= self.batch_lines {
            match self.lines.next() {
                Some(l) => {
                    text.extend_from_slice(l.as_ref().as_bytes());
                    offsets.push(text.len() as u64);
                }
                None => break,
            }
        }
        if offsets.len() == 1 {
            return None;
        }
        match self.tokenizer.tokenize_batch_raw(&text, &offsets) {
            Ok(b) => {
                self.current = Some(std::rc::Rc::new(b));
                self.next = 1;
                Some(Ok(BatchSentence { batch: self.current.as_ref().unwrap().clone(), index: 0 }))
            }
            Err(e) => {
                self.failed = true;
                Some(Err(e))
            }
        }
    }
}
