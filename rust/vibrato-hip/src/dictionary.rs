//! Dictionary for tokenization (reference: `vibrato/src/dictionary.rs`, `dictionary/builder.rs`, `dictionary/word_idx.rs`).
use std::io::{Read, Write};
use std::os::raw::c_char;
use std::ptr;
use std::sync::atomic::{AtomicBool, AtomicPtr, Ordering};

use vibrato_hip_sys as sys;

use crate::errors::{check, Result};

/// Type of a lexicon that contains the word (`dictionary.rs:30-40`).
#[derive(Clone, Copy, Default, Eq, PartialEq, Debug, Hash)]
#[repr(u8)]
pub enum LexType {
    /// System lexicon.
    #[default]
    System = 0,
    /// User lexicon.
    User = 1,
    /// Unknown words.
    Unknown = 2,
}

impl LexType {
    pub(crate) fn from_u32(x: u32) -> Self {
        match x {
            0 => Self::System,
            1 => Self::User,
            _ => Self::Unknown,
        }
    }
}

/// Identifier of a word (`dictionary/word_idx.rs:5-25`).
#[derive(Debug, Eq, PartialEq, Hash, Clone, Copy, Default)]
pub struct WordIdx {
    /// Type of a lexicon that contains this word.
    pub lex_type: LexType,
    /// Id of this word.
    pub word_id: u32,
}

impl WordIdx {
    /// Creates a new instance.
    #[inline(always)]
    pub const fn new(lex_type: LexType, word_id: u32) -> Self {
        Self { lex_type, word_id }
    }
}

/// Dictionary for tokenization: a handle to the host-side dictionary model of `libvibrato_hip.so`.
///
/// `Tokenizer::new` moves the dictionary in (`tokenizer.rs:26`); when the tokenizer builds its device image the C handle is
/// consumed and this object turns into a borrowed view of the same dictionary (the memory `feature()` strings point to
/// does not move), which is why the raw pointer lives in an atomic.
pub struct Dictionary {
    raw: AtomicPtr<sys::vbt_dict>,
    owned: AtomicBool,
}

// Safety: the C library never mutates a dictionary through a `*const vbt_dict`, and the mutating entry points take `self`.
unsafe impl Send for Dictionary {}
unsafe impl Sync for Dictionary {}

impl Dictionary {
    pub(crate) fn from_raw(raw: *mut sys::vbt_dict) -> Self {
        Self { raw: AtomicPtr::new(raw), owned: AtomicBool::new(true) }
    }

    #[inline(always)]
    pub(crate) fn raw(&self) -> *mut sys::vbt_dict {
        self.raw.load(Ordering::Acquire)
    }

    /// The tokenizer consumed the C handle: from now on this is a view of `view` (owned by the tokenizer).
    pub(crate) fn rebind_borrowed(&self, view: *const sys::vbt_dict) {
        self.owned.store(false, Ordering::Release);
        self.raw.store(view as *mut _, Ordering::Release);
    }

    /// Gets the feature string of a word (`dictionary.rs:108-114`).
    pub fn word_feature(&self, word_idx: WordIdx) -> &str {
        let (mut p, mut n): (*const c_char, usize) = (ptr::null(), 0);
        // Safety: valid handle; on success (p, n) is a UTF-8 string that lives as long as the dictionary.
        check(unsafe { sys::vbt_dict_word_feature(self.raw(), word_idx.lex_type as u32, word_idx.word_id, &mut p, &mut n) })
            .expect("word_idx out of range");
        unsafe { std::str::from_utf8_unchecked(std::slice::from_raw_parts(p as *const u8, n)) }
    }

    /// Exports the dictionary data (`dictionary.rs:142-150`): magic + bincode, the bytes `Dictionary::read` takes.
    /// Wrap the writer in a zstd encoder for `system.dic.zst` like `compile/src/main.rs:98`.
    pub fn write<W: Write>(&self, mut wtr: W) -> Result<usize> {
        let (mut p, mut n): (*mut u8, usize) = (ptr::null_mut(), 0);
        check(unsafe { sys::vbt_dict_write(self.raw(), -1, &mut p, &mut n) })?;
        // Safety: the library returned a malloc'ed buffer of n bytes.
        let r = wtr.write_all(unsafe { std::slice::from_raw_parts(p, n) });
        unsafe { sys::vbt_free(p as *mut _) };
        r?;
        Ok(n)
    }

    /// Creates a dictionary from raw dictionary data (`dictionary.rs:173-197`). A zstd frame around the data is detected and
    /// unwrapped, so both `system.dic` and `system.dic.zst` readers can be passed directly.
    pub fn read<R: Read>(mut rdr: R) -> Result<Self> {
        let mut buf = vec![];
        rdr.read_to_end(&mut buf)?;
        let mut raw = ptr::null_mut();
        check(unsafe { sys::vbt_dict_read(buf.as_ptr(), buf.len(), &mut raw) })?;
        Ok(Self::from_raw(raw))
    }

    /// Resets the user dictionary from a reader (`dictionary.rs:209-229`); `None` removes it.
    pub fn reset_user_lexicon_from_reader<R: Read>(self, user_lexicon_rdr: Option<R>) -> Result<Self> {
        match user_lexicon_rdr {
            Some(mut rdr) => {
                let mut buf = vec![];
                rdr.read_to_end(&mut buf)?;
                check(unsafe { sys::vbt_dict_set_user_lexicon(self.raw(), buf.as_ptr() as *const c_char, buf.len()) })?;
            }
            None => check(unsafe { sys::vbt_dict_set_user_lexicon(self.raw(), ptr::null(), 0) })?,
        }
        Ok(self)
    }

    /// Edits connection ids with the given mappings (`dictionary.rs:245-259`): the i-th item (1-origin) is the old id that
    /// becomes new id i.
    pub fn map_connection_ids_from_iter<L, R>(self, lmap: L, rmap: R) -> Result<Self>
    where
        L: IntoIterator<Item = u16>,
        R: IntoIterator<Item = u16>,
    {
        let l: Vec<u16> = lmap.into_iter().collect();
        let r: Vec<u16> = rmap.into_iter().collect();
        check(unsafe { sys::vbt_dict_map_connection_ids(self.raw(), l.as_ptr(), l.len(), r.as_ptr(), r.len()) })?;
        Ok(self)
    }

    /// `Connector::num_left` / `num_right` (`connector.rs:14-17`).
    pub fn num_connection_ids(&self) -> (usize, usize) {
        unsafe { (sys::vbt_dict_num_left(self.raw()) as usize, sys::vbt_dict_num_right(self.raw()) as usize) }
    }

    pub(crate) fn has_category(&self, name: &str) -> bool {
        unsafe { sys::vbt_dict_cate_id(self.raw(), name.as_ptr() as *const c_char, name.len()) >= 0 }
    }
}

impl Drop for Dictionary {
    fn drop(&mut self) {
        if self.owned.load(Ordering::Acquire) {
            // Safety: an owned handle is freed exactly once.
            unsafe { sys::vbt_dict_free(self.raw()) };
        }
    }
}

/// Builder for [`Dictionary`] from MeCab-format sources (`dictionary/builder.rs`).
pub struct SystemDictionaryBuilder {}

fn slurp<R: Read>(mut r: R) -> Result<Vec<u8>> {
    let mut buf = vec![];
    r.read_to_end(&mut buf)?;
    Ok(buf)
}

impl SystemDictionaryBuilder {
    /// Creates a new instance from readers in the MeCab format: `lex.csv`, `matrix.def`, `char.def`, `unk.def`
    /// (`builder.rs:64-89`).
    pub fn from_readers<S, C, P, U>(system_lexicon_rdr: S, connector_rdr: C, char_prop_rdr: P, unk_handler_rdr: U) -> Result<Dictionary>
    where
        S: Read,
        C: Read,
        P: Read,
        U: Read,
    {
        let (a, b, c, d) = (slurp(system_lexicon_rdr)?, slurp(connector_rdr)?, slurp(char_prop_rdr)?, slurp(unk_handler_rdr)?);
        let mut raw = ptr::null_mut();
        check(unsafe {
            sys::vbt_dict_from_sources(a.as_ptr() as _, a.len(), b.as_ptr() as _, b.len(), c.as_ptr() as _, c.len(), d.as_ptr() as _,
                                       d.len(), &mut raw)
        })?;
        Ok(Dictionary::from_raw(raw))
    }

    /// Creates a new instance from `lex.csv`, `bigram.right`, `bigram.left`, `bigram.cost`, `char.def`, `unk.def`
    /// (`builder.rs:111-160`): the compact connectors -- a `RawConnector`, or with `dual_connector` a `DualConnector` (small matrix
    /// over classes of connection ids + an 8-wide raw part).  On the device every connector is expanded into the dense matrix when
    /// the tokenizer is created.
    #[allow(clippy::too_many_arguments)]
    pub fn from_readers_with_bigram_info<S, R, L, C, P, U>(system_lexicon_rdr: S, bigram_right_rdr: R, bigram_left_rdr: L,
                                                           bigram_cost_rdr: C, char_prop_rdr: P, unk_handler_rdr: U,
                                                           dual_connector: bool) -> Result<Dictionary>
    where
        S: Read,
        R: Read,
        L: Read,
        C: Read,
        P: Read,
        U: Read,
    {
        let (a, r, l, c) = (slurp(system_lexicon_rdr)?, slurp(bigram_right_rdr)?, slurp(bigram_left_rdr)?, slurp(bigram_cost_rdr)?);
        let (p, u) = (slurp(char_prop_rdr)?, slurp(unk_handler_rdr)?);
        let mut raw = ptr::null_mut();
        check(unsafe {
            sys::vbt_dict_from_sources_bigram(a.as_ptr() as _, a.len(), r.as_ptr() as _, r.len(), l.as_ptr() as _, l.len(),
                                              c.as_ptr() as _, c.len(), p.as_ptr() as _, p.len(), u.as_ptr() as _, u.len(),
                                              dual_connector as i32, &mut raw)
        })?;
        Ok(Dictionary::from_raw(raw))
    }
}
