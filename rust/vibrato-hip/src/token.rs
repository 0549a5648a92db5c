//! Container of resultant tokens (reference: `vibrato/src/token.rs:8-139`).
use std::marker::PhantomData;
use std::ops::Range;

use vibrato_hip_sys as sys;

use crate::dictionary::{LexType, WordIdx};
use crate::tokenizer::worker::Worker;

/// Resultant token. `'w`: the worker (or batch) that holds the sentence text, `'t`: the tokenizer that holds the dictionary.
pub struct Token<'w, 't> {
    t: sys::vbt_token,
    _w: PhantomData<&'w ()>,
    _t: PhantomData<&'t ()>,
}

impl<'w, 't> Token<'w, 't> {
    #[inline(always)]
    pub(crate) fn new(t: sys::vbt_token) -> Self {
        Self { t, _w: PhantomData, _t: PhantomData }
    }

    /// Gets the position range of the token in characters (`token.rs:21-24`).
    #[inline(always)]
    pub fn range_char(&self) -> Range<usize> {
        self.t.start_char as usize..self.t.end_char as usize
    }

    /// Gets the position range of the token in bytes (`token.rs:28-32`).
    #[inline(always)]
    pub fn range_byte(&self) -> Range<usize> {
        self.t.start_byte as usize..self.t.end_byte as usize
    }

    /// Gets the surface string of the token (`token.rs:36-39`).
    #[inline(always)]
    pub fn surface(&self) -> &'w str {
        // Safety: a slice of a validated UTF-8 sentence cut at character boundaries, alive for 'w
        unsafe { std::str::from_utf8_unchecked(std::slice::from_raw_parts(self.t.surface as *const u8, self.t.surface_len)) }
    }

    /// Gets the word index of the token (`token.rs:42-45`).
    #[inline(always)]
    pub fn word_idx(&self) -> WordIdx {
        WordIdx::new(self.lex_type(), self.t.word_id)
    }

    /// Gets the feature string of the token (`token.rs:49-54`).
    #[inline(always)]
    pub fn feature(&self) -> &'t str {
        // Safety: dictionary memory, alive for 't
        unsafe { std::str::from_utf8_unchecked(std::slice::from_raw_parts(self.t.feature as *const u8, self.t.feature_len)) }
    }

    /// Gets the lexicon type where the token is from (`token.rs:58-60`).
    #[inline(always)]
    pub fn lex_type(&self) -> LexType {
        LexType::from_u32(self.t.lex_type)
    }

    /// Gets the left id of the token's node (`token.rs:64-67`).
    #[inline(always)]
    pub fn left_id(&self) -> u16 {
        self.t.left_id
    }

    /// Gets the right id of the token's node (`token.rs:71-74`).
    #[inline(always)]
    pub fn right_id(&self) -> u16 {
        self.t.right_id
    }

    /// Gets the word cost of the token's node (`token.rs:78-87`).
    #[inline(always)]
    pub fn word_cost(&self) -> i16 {
        self.t.word_cost
    }

    /// Gets the total cost from BOS to the token's node (`token.rs:89-92`).
    #[inline(always)]
    pub fn total_cost(&self) -> i32 {
        self.t.total_cost
    }
}

impl std::fmt::Debug for Token<'_, '_> {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("Token")
            .field("surface", &self.surface())
            .field("range_char", &self.range_char())
            .field("range_byte", &self.range_byte())
            .field("feature", &self.feature())
            .field("lex_type", &self.lex_type())
            .field("left_id", &self.left_id())
            .field("right_id", &self.right_id())
            .field("word_cost", &self.word_cost())
            .field("total_cost", &self.total_cost())
            .finish()
    }
}

/// Iterator of tokens (`token.rs:112-139`).
pub struct TokenIter<'w, 't> {
    worker: &'w Worker<'t>,
    i: usize,
}

impl<'w, 't> TokenIter<'w, 't> {
    #[inline(always)]
    pub(crate) const fn new(worker: &'w Worker<'t>, i: usize) -> Self {
        Self { worker, i }
    }
}

impl<'w, 't> Iterator for TokenIter<'w, 't> {
    type Item = Token<'w, 't>;

    #[inline(always)]
    fn next(&mut self) -> Option<Self::Item> {
        if self.i < self.worker.num_tokens() {
            let t = self.worker.token(self.i);
            self.i += 1;
            Some(t)
        } else {
            None
        }
    }
}
