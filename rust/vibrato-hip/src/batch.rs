//! Result of one device batch (new: the reference tokenizes one sentence per call).
use std::marker::PhantomData;
use std::mem::MaybeUninit;
use std::os::raw::c_char;
use std::ptr;

use vibrato_hip_sys as sys;

use crate::errors::{check, Result};
use crate::token::Token;
use crate::tokenizer::Tokenizer;

/// Output modes of the `tokenize` CLI (`tokenize/src/main.rs:15-29`).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum OutputMode {
    /// `surface\tfeature` lines + `EOS`
    Mecab = 0,
    /// surfaces joined by one space
    Wakati = 1,
    /// MeCab lines + lex_type / ids / costs
    Detail = 2,
}

/// Tokens of every sentence of one [`Tokenizer::tokenize_batch`] call; owns its copy of the text.
pub struct Batch<'t> {
    raw: *mut sys::vbt_batch,
    _t: PhantomData<&'t Tokenizer>,
}

// Safety: a finished batch is immutable host memory.
unsafe impl Send for Batch<'_> {}
unsafe impl Sync for Batch<'_> {}

impl<'t> Batch<'t> {
    pub(crate) fn from_raw(raw: *mut sys::vbt_batch, _tokenizer: &'t Tokenizer) -> Self {
        Self { raw, _t: PhantomData }
    }

    /// Number of sentences.
    pub fn len(&self) -> usize {
        unsafe { sys::vbt_batch_num_sentences(self.raw) as usize }
    }

    /// No sentences at all.
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }

    /// `Worker::num_tokens` of sentence `s`.
    pub fn num_tokens(&self, s: usize) -> usize {
        unsafe { sys::vbt_batch_num_tokens(self.raw, s as u64) as usize }
    }

    /// `Worker::token(i)` of sentence `s`.
    pub fn token<'b>(&'b self, s: usize, i: usize) -> Token<'b, 't> {
        let mut t = MaybeUninit::<sys::vbt_token>::uninit();
        check(unsafe { sys::vbt_batch_token(self.raw, s as u64, i as u32, t.as_mut_ptr()) }).expect("token index out of range");
        Token::new(unsafe { t.assume_init() })
    }

    /// The raw 24-byte records of sentence `s`, as written by the device.
    pub fn records(&self, s: usize) -> &[sys::vbt_token_rec] {
        let n = self.num_tokens(s);
        let p = unsafe { sys::vbt_batch_records(self.raw, s as u64) };
        if n == 0 || p.is_null() {
            &[]
        } else {
            unsafe { std::slice::from_raw_parts(p, n) }
        }
    }

    /// The exact bytes the reference's `tokenize` CLI prints for these sentences (`tokenize/src/main.rs:83-127`).
    pub fn format(&self, mode: OutputMode) -> Result<String> {
        let (mut p, mut n): (*mut c_char, usize) = (ptr::null_mut(), 0);
        check(unsafe { sys::vbt_batch_format(self.raw, mode as i32, &mut p, &mut n) })?;
        // Safety: malloc'ed UTF-8 (surfaces and features are UTF-8, the rest ASCII)
        let s = unsafe { std::str::from_utf8_unchecked(std::slice::from_raw_parts(p as *const u8, n)) }.to_owned();
        unsafe { sys::vbt_free(p as *mut _) };
        Ok(s)
    }
}

impl Drop for Batch<'_> {
    fn drop(&mut self) {
        unsafe { sys::vbt_batch_free(self.raw) };
    }
}
