//! Definition of errors (reference: `vibrato/src/errors.rs:7-42`). The variants a caller can match on are kept; errors that
//! originate behind the C ABI carry the message of `vbt_last_error()`.
use std::error::Error;
use std::ffi::CStr;
use std::fmt;
use std::os::raw::c_int;

use vibrato_hip_sys as sys;

/// A specialized Result type for Vibrato.
pub type Result<T, E = VibratoError> = std::result::Result<T, E>;

/// The error type for Vibrato.
#[derive(Debug)]
pub enum VibratoError {
    /// Invalid argument (`VBT_ERR_INVALID_ARGUMENT`).
    InvalidArgument(InvalidArgumentError),
    /// Invalid format of an input file (`VBT_ERR_INVALID_FORMAT`).
    InvalidFormat(InvalidFormatError),
    /// An integer in an input file did not parse (`VBT_ERR_PARSE_INT`).
    ParseInt(String),
    /// Input text that is not UTF-8 (`VBT_ERR_UTF8`; unreachable through `&str` arguments).
    Utf8(String),
    /// `std::io::Error` of a reader handed to a constructor.
    StdIo(std::io::Error),
    /// HIP runtime failure, no gfx950 device, or an input the device image cannot represent
    /// (`VBT_ERR_DEVICE`, `VBT_ERR_UNSUPPORTED`, `VBT_ERR_INVALID_STATE`). New: the reference is CPU-only.
    Device(String),
}

/// Error used when the argument is invalid.
#[derive(Debug)]
pub struct InvalidArgumentError {
    pub(crate) arg: &'static str,
    pub(crate) msg: String,
}

/// Error used when the input format is invalid.
#[derive(Debug)]
pub struct InvalidFormatError {
    pub(crate) arg: &'static str,
    pub(crate) msg: String,
}

impl VibratoError {
    pub(crate) fn invalid_argument<S: Into<String>>(arg: &'static str, msg: S) -> Self {
        Self::InvalidArgument(InvalidArgumentError { arg, msg: msg.into() })
    }
}

impl fmt::Display for InvalidArgumentError {
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        write!(f, "InvalidArgumentError: {}: {}", self.arg, self.msg)
    }
}
impl fmt::Display for InvalidFormatError {
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        write!(f, "InvalidFormatError: {}: {}", self.arg, self.msg)
    }
}
impl Error for InvalidArgumentError {}
impl Error for InvalidFormatError {}

impl fmt::Display for VibratoError {
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        match self {
            Self::InvalidArgument(e) => e.fmt(f),
            Self::InvalidFormat(e) => e.fmt(f),
            Self::ParseInt(m) => write!(f, "ParseIntError: {m}"),
            Self::Utf8(m) => write!(f, "Utf8Error: {m}"),
            Self::StdIo(e) => e.fmt(f),
            Self::Device(m) => write!(f, "DeviceError: {m}"),
        }
    }
}
impl Error for VibratoError {}

impl From<std::io::Error> for VibratoError {
    fn from(e: std::io::Error) -> Self {
        Self::StdIo(e)
    }
}

/// Maps a `vbt_status` to `Result`, reading the thread-local message of the library.
pub(crate) fn check(rc: c_int) -> Result<()> {
    if rc == sys::VBT_OK {
        return Ok(());
    }
    // Safety: vbt_last_error returns a NUL-terminated string owned by the library (thread-local, valid until the next call).
    let msg = unsafe { CStr::from_ptr(sys::vbt_last_error()) }.to_string_lossy().into_owned();
    Err(match rc {
        sys::VBT_ERR_INVALID_ARGUMENT => VibratoError::InvalidArgument(InvalidArgumentError { arg: "ffi", msg }),
        sys::VBT_ERR_INVALID_FORMAT => VibratoError::InvalidFormat(InvalidFormatError { arg: "ffi", msg }),
        sys::VBT_ERR_PARSE_INT => VibratoError::ParseInt(msg),
        sys::VBT_ERR_UTF8 => VibratoError::Utf8(msg),
        _ => VibratoError::Device(msg),
    })
}
