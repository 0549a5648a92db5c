//! Tokens of a finished tokenization.
//!
//! API-compatible with `vibrato::token` (`vibrato/src/token.rs:8-139`): the same accessor names and return types, the same two
//! lifetimes.  A token here is a plain copy of the C ABI's `vbt_token` view -- the device wrote the 24-byte record, the library
//! resolved `word_idx` against the host-side dictionary -- so every accessor is a field read.
use std::fmt;
use std::marker::PhantomData;
use std::ops::Range;

use vibrato_hip_sys as sys;

use crate::dictionary::{LexType, WordIdx};
use crate::tokenizer::worker::Worker;

/// One token of the best path.
///
/// * `'w` -- the holder of the sentence text (a [`Worker`] or a [`crate::batch::Batch`]): `surface()` borrows from it.
/// * `'t` -- the tokenizer, i.e. the dictionary: `feature()` borrows from it.
#[derive(Clone, Copy)]
pub struct Token<'w, 't> {
    view: sys::vbt_token,
    holder: PhantomData<&'w [u8]>,
    dictionary: PhantomData<&'t [u8]>,
}

/// `&str` over memory owned by the library.
///
/// # Safety
/// `(p, n)` must describe UTF-8 bytes that outlive `'a`; the library guarantees both for the two pointer pairs of a `vbt_token`
/// (a slice of a validated sentence cut at character boundaries; a dictionary feature string).
unsafe fn borrowed_str<'a>(p: *const std::os::raw::c_char, n: usize) -> &'a str {
    std::str::from_utf8_unchecked(std::slice::from_raw_parts(p as *const u8, n))
}

impl<'w, 't> Token<'w, 't> {
    pub(crate) fn new(view: sys::vbt_token) -> Self {
        Self { view, holder: PhantomData, dictionary: PhantomData }
    }

    /// Surface form: the slice of the input sentence this token covers (skipped spaces belong to no token).
    pub fn surface(&self) -> &'w str {
        unsafe { borrowed_str(self.view.surface, self.view.surface_len) }
    }

    /// Feature string of the word, exactly as it stands in the lexicon (or in `unk.def` for an unknown word).
    pub fn feature(&self) -> &'t str {
        unsafe { borrowed_str(self.view.feature, self.view.feature_len) }
    }

    /// Half-open range of the token in characters of the sentence.
    pub fn range_char(&self) -> Range<usize> {
        (self.view.start_char as usize)..(self.view.end_char as usize)
    }

    /// Half-open range of the token in bytes of the sentence.
    pub fn range_byte(&self) -> Range<usize> {
        (self.view.start_byte as usize)..(self.view.end_byte as usize)
    }

    /// Which lexicon the word comes from.
    pub fn lex_type(&self) -> LexType {
        LexType::from_u32(self.view.lex_type)
    }

    /// `(lex_type, word_id)` of the word.
    pub fn word_idx(&self) -> WordIdx {
        WordIdx::new(self.lex_type(), self.view.word_id)
    }

    /// Left connection id of the word.
    pub fn left_id(&self) -> u16 {
        self.view.left_id
    }

    /// Right connection id of the word.
    pub fn right_id(&self) -> u16 {
        self.view.right_id
    }

    /// Cost of the word itself.
    pub fn word_cost(&self) -> i16 {
        self.view.word_cost
    }

    /// Cost of the best path from BOS up to and including this token (connection to the next token excluded).
    pub fn total_cost(&self) -> i32 {
        self.view.total_cost
    }
}

impl fmt::Debug for Token<'_, '_> {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        // same field names and order as the reference prints
        let mut d = f.debug_struct("Token");
        d.field("surface", &self.surface()).field("range_char", &self.range_char()).field("range_byte", &self.range_byte());
        d.field("feature", &self.feature()).field("lex_type", &self.lex_type());
        d.field("left_id", &self.left_id()).field("right_id", &self.right_id());
        d.field("word_cost", &self.word_cost()).field("total_cost", &self.total_cost());
        d.finish()
    }
}

/// Iterator over the tokens of a worker, first token first (`Worker::token_iter`).
pub struct TokenIter<'w, 't> {
    worker: &'w Worker<'t>,
    pending: Range<usize>,
}

impl<'w, 't> TokenIter<'w, 't> {
    pub(crate) fn over(worker: &'w Worker<'t>) -> Self {
        Self { worker, pending: 0..worker.num_tokens() }
    }
}

impl<'w, 't> Iterator for TokenIter<'w, 't> {
    type Item = Token<'w, 't>;

    fn next(&mut self) -> Option<Self::Item> {
        self.pending.next().map(|i| self.worker.token(i))
    }

    fn size_hint(&self) -> (usize, Option<usize>) {
        self.pending.size_hint()
    }
}

impl ExactSizeIterator for TokenIter<'_, '_> {}

impl DoubleEndedIterator for TokenIter<'_, '_> {
    fn next_back(&mut self) -> Option<Self::Item> {
        self.pending.next_back().map(|i| self.worker.token(i))
    }
}
