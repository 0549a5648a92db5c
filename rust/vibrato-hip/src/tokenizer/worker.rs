//! The per-sentence interface: one sentence in, its tokens out (API of `vibrato/src/tokenizer/worker.rs:13-103`).
use std::marker::PhantomData;
use std::mem::MaybeUninit;
use std::os::raw::c_char;
use std::ptr;

use vibrato_hip_sys as sys;

use crate::errors::check;
use crate::token::{Token, TokenIter};
use crate::tokenizer::Tokenizer;

/// `(connection id, probability)` pairs in descending order of probability, id 0 left out (`dictionary/mapper.rs:84`).
pub type ConnIdProbs = Vec<(usize, f64)>;

/// Holds one sentence and, after `tokenize()`, its tokens.  `tokenize()` costs NO kernel launch in steady state: the worker's first
/// call starts a resident one-wavefront kernel that polls a doorbell in the worker's pinned host block, reads the text out of it
/// and writes the token records back into it (it leaves after ~2 ms without a call, or after 4096 sentences, and the next call
/// starts it again).  ~39 us for a 47-character sentence on an MI355X (21 us up to 13 characters), against ~12.5 us on one CPU
/// core for the reference -- the reference's calling pattern keeps working, but it is 3x SLOWER than the CPU here; the throughput
/// path is [`Tokenizer::tokenize_batch`] / [`Tokenizer::tokenize_lines`] (three orders of magnitude above this loop).
pub struct Worker<'t> {
    raw: *mut sys::vbt_worker,
    tokenizer: &'t Tokenizer,
    _not_sync: PhantomData<std::cell::Cell<()>>, // used from one thread at a time, as in the reference (`&mut self` methods)
}

// Safety: a worker may move to another thread (the reference's Worker is Send); the library keeps no thread affinity.
unsafe impl Send for Worker<'_> {}

impl<'t> Worker<'t> {
    pub(crate) fn new(tokenizer: &'t Tokenizer) -> Self {
        let tok = tokenizer.raw().expect("vibrato-hip: cannot create the device tokenizer");
        let mut raw = ptr::null_mut();
        check(unsafe { sys::vbt_worker_new(tok, &mut raw) }).expect("vibrato-hip: vbt_worker_new");
        Self { raw, tokenizer, _not_sync: PhantomData }
    }

    /// Replaces the sentence the worker holds; the previous tokens are gone (`worker.rs:34-45`).
    pub fn reset_sentence<S: AsRef<str>>(&mut self, input: S) {
        let s = input.as_ref();
        // a &str is valid UTF-8, the only error this call has
        check(unsafe { sys::vbt_worker_reset_sentence(self.raw, s.as_ptr() as *const c_char, s.len()) }).expect("vbt_worker_reset_sentence");
    }

    /// Runs the lattice construction and the Viterbi search for the held sentence on the device (`worker.rs:49-55`).
    ///
    /// # Panics
    ///
    /// If the HIP runtime reports an error: the reference's signature has no error to return, and there is no CPU path.
    pub fn tokenize(&mut self) {
        check(unsafe { sys::vbt_worker_tokenize(self.raw) }).expect("vibrato-hip: device error in tokenize()");
    }

    /// How many tokens the last `tokenize()` produced (`worker.rs:59-61`).
    pub fn num_tokens(&self) -> usize {
        unsafe { sys::vbt_worker_num_tokens(self.raw) as usize }
    }

    /// Token number `i` of the sentence, counted from its start (`worker.rs:65-68`).
    pub fn token<'w>(&'w self, i: usize) -> Token<'w, 't> {
        let mut t = MaybeUninit::<sys::vbt_token>::uninit();
        check(unsafe { sys::vbt_worker_token(self.raw, i as u32, t.as_mut_ptr()) }).expect("token index out of range");
        // Safety: filled by the library on success; surface points into the worker's copy of the sentence ('w), feature into
        // dictionary memory owned by the tokenizer ('t)
        Token::new(unsafe { t.assume_init() })
    }

    /// All tokens of the last `tokenize()`, in sentence order (`worker.rs:72-74`).
    pub fn token_iter<'w>(&'w self) -> TokenIter<'w, 't> {
        TokenIter::over(self)
    }

    /// Starts (or restarts) counting how often each connection id is used by the lattices this worker builds (`worker.rs:77-84`).
    pub fn init_connid_counter(&mut self) {
        check(unsafe { sys::vbt_worker_init_connid_counter(self.raw) }).expect("vbt_worker_init_connid_counter");
    }

    /// Adds the lattice of the last `tokenize()` to the counters (`worker.rs:90-93`).
    ///
    /// # Panics
    ///
    /// Without a preceding [`Self::init_connid_counter()`], like the reference.
    pub fn update_connid_counts(&mut self) {
        check(unsafe { sys::vbt_worker_update_connid_counts(self.raw) }).expect("init_connid_counter() has never been called");
    }

    /// The counters as probabilities: `(left ids, right ids)`, what the reference's `reorder` tool writes to `*.lmap` / `*.rmap`
    /// (`worker.rs:101-103`).
    ///
    /// # Panics
    ///
    /// Without a preceding [`Self::init_connid_counter()`], like the reference.
    pub fn compute_connid_probs(&self) -> (ConnIdProbs, ConnIdProbs) {
        let (nl, nr) = self.tokenizer.dictionary().num_connection_ids();
        let (mut li, mut lp) = (vec![0u32; nl.saturating_sub(1)], vec![0f64; nl.saturating_sub(1)]);
        let (mut ri, mut rp) = (vec![0u32; nr.saturating_sub(1)], vec![0f64; nr.saturating_sub(1)]);
        check(unsafe { sys::vbt_worker_compute_connid_probs(self.raw, li.as_mut_ptr(), lp.as_mut_ptr(), ri.as_mut_ptr(), rp.as_mut_ptr()) })
            .expect("init_connid_counter() has never been called");
        let zip = |ids: Vec<u32>, ps: Vec<f64>| ids.into_iter().map(|i| i as usize).zip(ps).collect::<ConnIdProbs>();
        (zip(li, lp), zip(ri, rp))
    }
}

impl Drop for Worker<'_> {
    fn drop(&mut self) {
        unsafe { sys::vbt_worker_free(self.raw) };
    }
}
