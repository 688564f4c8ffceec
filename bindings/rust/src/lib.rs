//! Rust side of the drop-in boundary: `libconstriction_amd.so` (include/constriction_amd.h) behind wrappers that keep
//! the method names of the reference's `stream::stack::AnsCoder` and `stream::queue::{RangeEncoder, RangeDecoder}`.
//!
//! The reference's coders are single host objects (`AnsCoder::encode_iid_symbols_reverse`, src/stream/stack.rs:835-849;
//! `encode_symbols_reverse` :784-797; `into_compressed` :891-895; `from_compressed` :299-318; `decode_iid_symbols`
//! src/stream/mod.rs:1016-1031; `RangeEncoder::encode_*` src/stream/queue.rs:612-705; `RangeDecoder` :847-868, 968-1033).
//! Here every call codes MANY independent coders at once -- one per stream of a batch, on an AMD MI355X -- and each
//! stream's compressed words are bit for bit what the reference coder produces for that stream alone.
//!
//! Layers: [`ffi`] is the raw `extern "C"` block (generated from the header, every entry point); [`hip`] the few HIP
//! runtime calls needed for device buffers and streams; this file the safe wrappers.  A maintainer of the reference
//! would add this crate as an optional dependency (feature `mi355x`) and re-export [`BatchedAnsCoder`] from
//! `stream::stack`.
//!
//! No Rust toolchain exists in the image this repository is built in: the crate is checked there by
//! tests/test_rust_binding.py (every declaration of [`ffi`] against the header: names, arity, types; every `ffi::` call
//! below against its declaration), not by `cargo build`.

pub mod ffi;
pub mod hip;

use core::ffi::c_void;
use core::ops::RangeInclusive;

pub use ffi::{CstChainHeads as ChainHeads, CstCoderConfig as CoderConfig, CstRangeState as RangeState};
pub use hip::{DeviceBuffer, HipError, Stream};

/// `DefaultAnsCoder` / the whole Python API: `AnsCoder<u32, u64>` with 24-bit models (src/stream/stack.rs:139).
pub const DEFAULT: CoderConfig = CoderConfig { word_bits: 32, state_bits: 64, precision: 24 };
/// The lookup-model preset of benches/lookup.rs:32-34 and of BASELINE configs C2 / C3.
pub const LOOKUP: CoderConfig = CoderConfig { word_bits: 32, state_bits: 64, precision: 12 };
/// `SmallAnsCoder` = `AnsCoder<u16, u32>` with 12-bit models (src/stream/stack.rs:153).
pub const SMALL: CoderConfig = CoderConfig { word_bits: 16, state_bits: 32, precision: 12 };

/// Call-level errors (`cst_status`).
#[derive(Clone, Debug, PartialEq, Eq)]
pub enum Error {
    InvalidArgument,
    /// A HIP runtime call inside the library failed; the text is `cst_last_hip_error()`.
    Hip(String),
    /// No gfx950 device: the library has no CPU fallback.
    NoDevice,
    /// The model cannot be built (`Err(())` of the reference's constructors, the `assert!` on `std > 0`).
    Model,
    OutOfMemory,
    /// A HIP call of this crate's own buffer handling failed.
    Runtime(HipError),
    Unknown(i32),
}

impl From<HipError> for Error {
    fn from(e: HipError) -> Self {
        Error::Runtime(e)
    }
}

pub type Result<T> = core::result::Result<T, Error>;

fn check(status: ffi::CstStatus) -> Result<()> {
    match status {
        ffi::CST_OK => Ok(()),
        ffi::CST_ERR_INVALID_ARGUMENT => Err(Error::InvalidArgument),
        ffi::CST_ERR_HIP => {
            let text = unsafe { std::ffi::CStr::from_ptr(ffi::cst_last_hip_error()) };
            Err(Error::Hip(text.to_string_lossy().into_owned()))
        }
        ffi::CST_ERR_NO_DEVICE => Err(Error::NoDevice),
        ffi::CST_ERR_MODEL => Err(Error::Model),
        ffi::CST_ERR_OUT_OF_MEMORY => Err(Error::OutOfMemory),
        other => Err(Error::Unknown(other)),
    }
}

/// Per-stream outcome (`cst_stream_status`): what the reference returns from the corresponding single coder.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum StreamStatus {
    Ok,
    /// `DefaultEncoderFrontendError::ImpossibleSymbol` (src/lib.rs:376-385).
    ImpossibleSymbol,
    /// The output slab was too small (the reference: a backend `WriteError`).
    Capacity,
    /// Trailing zero word (`from_compressed`, stack.rs:299-318) or `RangeDecoder` `InvalidData` (queue.rs:989-993).
    InvalidData,
    /// Chain coder ran out of compressed data / remainders (chain.rs:854-890).
    OutOfData,
    Other(i32),
}

impl From<i32> for StreamStatus {
    fn from(code: i32) -> Self {
        match code {
            ffi::CST_STREAM_OK => StreamStatus::Ok,
            ffi::CST_STREAM_IMPOSSIBLE_SYMBOL => StreamStatus::ImpossibleSymbol,
            ffi::CST_STREAM_CAPACITY => StreamStatus::Capacity,
            ffi::CST_STREAM_INVALID_DATA => StreamStatus::InvalidData,
            ffi::CST_STREAM_OUT_OF_DATA => StreamStatus::OutOfData,
            other => StreamStatus::Other(other),
        }
    }
}

/// Memory layout of the `i32` symbol matrix of a batch (`cst_layout`).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Layout {
    /// `symbols[stream][t]`
    StreamMajor,
    /// `symbols[t][stream]`
    SymbolMajor,
}

impl Layout {
    fn raw(self) -> ffi::CstLayout {
        match self {
            Layout::StreamMajor => ffi::CST_LAYOUT_STREAM_MAJOR,
            Layout::SymbolMajor => ffi::CST_LAYOUT_SYMBOL_MAJOR,
        }
    }
}

/// Number of visible gfx950 devices.
pub fn device_count() -> Result<usize> {
    let n = unsafe { ffi::cst_device_count() };
    if n < 0 {
        check(n)?;
    }
    Ok(n as usize)
}

/// Symbol types narrower than `i32` that the batched coders take (ABI 4: `cst_ans_*_batch_sym`, `cst_symbols_widen` / `_narrow`).
pub trait NarrowSymbol: Copy {
    const BYTES: i32;
}
impl NarrowSymbol for i8 {
    const BYTES: i32 = 1;
}
impl NarrowSymbol for i16 {
    const BYTES: i32 = 2;
}

/// Widens a narrow symbol matrix to `i32` on the device (for the coders that take `i32` only: range, per-symbol, checkpointed calls).
pub fn widen_symbols<T: NarrowSymbol>(narrow: &DeviceBuffer<T>, wide: &mut DeviceBuffer<i32>, stream: &Stream) -> Result<()> {
    if wide.len() < narrow.len() {
        return Err(Error::InvalidArgument);
    }
    check(unsafe { ffi::cst_symbols_widen(narrow.as_ptr() as *const c_void, T::BYTES, narrow.len(), wide.as_mut_ptr(), stream.as_raw()) })
}

/// Narrows decoded `i32` symbols on the device (values outside the type are clamped: the caller checked the model's support).
pub fn narrow_symbols<T: NarrowSymbol>(wide: &DeviceBuffer<i32>, narrow: &mut DeviceBuffer<T>, stream: &Stream) -> Result<()> {
    if narrow.len() < wide.len() {
        return Err(Error::InvalidArgument);
    }
    check(unsafe { ffi::cst_symbols_narrow(wide.as_ptr(), wide.len(), narrow.as_mut_ptr() as *mut c_void, T::BYTES, stream.as_raw()) })
}

/// `true` if the loaded library was built from the header this crate was generated from.
pub fn abi_matches() -> bool {
    unsafe { ffi::cst_abi_version() == ffi::CST_ABI_VERSION }
}

/// The kernel family this thread's last batched coder call launched (diagnostics: the dispatcher picks by shape and flags).
pub fn last_kernel_name() -> String {
    unsafe { std::ffi::CStr::from_ptr(ffi::cst_last_kernel_name()) }.to_string_lossy().into_owned()
}

// ---------------------------------------------------------------------------------------------------------------------
// entropy models
// ---------------------------------------------------------------------------------------------------------------------

/// A device-resident entropy model over a contiguous `i32` support: the encoder side is
/// `EncoderModel::left_cumulative_and_probability` as a table (ContiguousCategoricalEntropyModel,
/// src/stream/model/categorical/contiguous.rs:673-700), the decoder side `ContiguousLookupDecoderModel`
/// (lookup_contiguous.rs:564-605).
pub struct DeviceModel {
    raw: *mut ffi::CstModel,
}

impl DeviceModel {
    /// Any tabulated model: `cdf[0] = 0 < cdf[1] < ... < cdf[n] = 2^precision`.
    pub fn from_cdf(precision: u32, min_symbol: i32, cdf: &[u32]) -> Result<Self> {
        if cdf.len() < 2 {
            return Err(Error::InvalidArgument);
        }
        let mut raw = core::ptr::null_mut();
        check(unsafe { ffi::cst_model_create_table(precision as i32, min_symbol, (cdf.len() - 1) as i32, cdf.as_ptr(), &mut raw) })?;
        Ok(DeviceModel { raw })
    }

    /// `LeakyQuantizer::<f64, i32, _, P>::new(support).quantize(Gaussian::new(mean, std))`
    /// (src/stream/model/quantize.rs:284-308, 525-568), tabulated on the device in bit-exact f64.
    pub fn quantized_gaussian(precision: u32, support: RangeInclusive<i32>, mean: f64, std: f64, stream: &Stream) -> Result<Self> {
        let mut raw = core::ptr::null_mut();
        check(unsafe {
            ffi::cst_model_create_gaussian(precision as i32, *support.start(), *support.end(), mean, std, stream.as_raw(), &mut raw)
        })?;
        Ok(DeviceModel { raw })
    }

    /// One quantized Gaussian PER STREAM (BASELINE config C3): stream `s` uses `Gaussian(means[s], stds[s])`.
    pub fn quantized_gaussian_per_stream(
        precision: u32,
        support: RangeInclusive<i32>,
        means: &DeviceBuffer<f64>,
        stds: &DeviceBuffer<f64>,
        stream: &Stream,
    ) -> Result<Self> {
        if means.len() != stds.len() {
            return Err(Error::InvalidArgument);
        }
        let mut raw = core::ptr::null_mut();
        check(unsafe {
            ffi::cst_model_create_gaussian_per_stream(
                precision as i32,
                *support.start(),
                *support.end(),
                means.as_ptr(),
                stds.as_ptr(),
                means.len(),
                stream.as_raw(),
                &mut raw,
            )
        })?;
        Ok(DeviceModel { raw })
    }

    /// A tabulated model over arbitrary distinct symbols (NonContiguousCategorical*Model); code with
    /// [`DeviceModel::symbols_to_indices`] / [`DeviceModel::indices_to_symbols`] around the coder calls.
    pub fn from_cdf_noncontiguous(precision: u32, symbols: &[i32], cdf: &[u32]) -> Result<Self> {
        if symbols.len() + 1 != cdf.len() {
            return Err(Error::InvalidArgument);
        }
        let mut raw = core::ptr::null_mut();
        check(unsafe {
            ffi::cst_model_create_table_noncontiguous(precision as i32, symbols.len() as i32, symbols.as_ptr(), cdf.as_ptr(), &mut raw)
        })?;
        Ok(DeviceModel { raw })
    }

    /// `Categorical(probabilities, perfect=True)`: `perfectly_quantized_probabilities` (categorical.rs:56-177) + cumulation.
    pub fn perfect_categorical_cdf(probabilities: &[f64], precision: u32) -> Result<Vec<u32>> {
        let mut cdf = vec![0u32; probabilities.len() + 1];
        check(unsafe { ffi::cst_categorical_perfect_cdf(probabilities.as_ptr(), probabilities.len(), precision as i32, cdf.as_mut_ptr()) })?;
        Ok(cdf)
    }

    pub fn precision(&self) -> u32 {
        unsafe { ffi::cst_model_precision(self.raw) as u32 }
    }

    pub fn min_symbol(&self) -> i32 {
        unsafe { ffi::cst_model_min_symbol(self.raw) }
    }

    pub fn n_symbols(&self) -> usize {
        unsafe { ffi::cst_model_n_symbols(self.raw) as usize }
    }

    pub fn n_tables(&self) -> usize {
        unsafe { ffi::cst_model_n_tables(self.raw) }
    }

    /// The cumulatives of table `index` (`n_symbols + 1` entries); synchronises `stream`.
    pub fn cdf(&self, index: usize, stream: &Stream) -> Result<Vec<u32>> {
        let mut cdf = vec![0u32; self.n_symbols() + 1];
        check(unsafe { ffi::cst_model_get_cdf(self.raw, index, cdf.as_mut_ptr(), stream.as_raw()) })?;
        Ok(cdf)
    }

    pub fn symbols_to_indices(&self, symbols: &DeviceBuffer<i32>, indices: &mut DeviceBuffer<i32>, stream: &Stream) -> Result<()> {
        if indices.len() < symbols.len() {
            return Err(Error::InvalidArgument);
        }
        check(unsafe { ffi::cst_symbols_to_indices(self.raw, symbols.as_ptr(), symbols.len(), indices.as_mut_ptr(), stream.as_raw()) })
    }

    pub fn indices_to_symbols(&self, indices: &DeviceBuffer<i32>, symbols: &mut DeviceBuffer<i32>, stream: &Stream) -> Result<()> {
        if symbols.len() < indices.len() {
            return Err(Error::InvalidArgument);
        }
        check(unsafe { ffi::cst_indices_to_symbols(self.raw, indices.as_ptr(), indices.len(), symbols.as_mut_ptr(), stream.as_raw()) })
    }

    pub fn as_raw(&self) -> *const ffi::CstModel {
        self.raw
    }
}

impl Drop for DeviceModel {
    fn drop(&mut self) {
        unsafe { ffi::cst_model_destroy(self.raw) };
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// compressed batches
// ---------------------------------------------------------------------------------------------------------------------

/// What a batched encoder leaves behind: one fixed-stride slab of words per stream (emission order, final state words
/// last, exactly `into_compressed()` of the reference coder), the word count and the status of every stream.
pub struct EncodedBatch {
    pub words: DeviceBuffer<u32>,
    pub n_words: DeviceBuffer<u32>,
    pub status: DeviceBuffer<i32>,
    pub stride_words: usize,
    pub n_streams: usize,
    pub config: CoderConfig,
}

impl EncodedBatch {
    fn allocate(n_streams: usize, stride_words: usize, config: CoderConfig) -> Result<Self> {
        Ok(EncodedBatch {
            words: DeviceBuffer::new(n_streams.checked_mul(stride_words).ok_or(Error::InvalidArgument)?)?,
            n_words: DeviceBuffer::new(n_streams)?,
            status: DeviceBuffer::new(n_streams)?,
            stride_words,
            n_streams,
            config,
        })
    }

    /// Per-stream results on the host (synchronises the device).
    pub fn statuses(&self) -> Result<Vec<StreamStatus>> {
        Ok(self.status.to_vec()?.into_iter().map(StreamStatus::from).collect())
    }

    /// `into_compressed()` of stream `s` on the host (synchronises the device).
    pub fn stream_to_vec(&self, s: usize) -> Result<Vec<u32>> {
        let counts = self.n_words.to_vec()?;
        let all = self.words.to_vec_prefix((s + 1) * self.stride_words)?;
        let begin = s * self.stride_words;
        Ok(all[begin..begin + counts[s] as usize].to_vec())
    }
}

/// The concatenation of every stream's `into_compressed()` result plus `offsets[n_streams + 1]` (the container layout
/// of src/pybindings/stream/stack.rs:149-166); what the gather over RCCL moves.
pub struct PackedBatch {
    pub words: DeviceBuffer<u32>,
    pub offsets: DeviceBuffer<u64>,
    pub n_words: DeviceBuffer<u32>,
    pub n_streams: usize,
    pub config: CoderConfig,
}

/// `EncodedBatch` -> `PackedBatch`: one asynchronous kernel (prefix sum of the counts fused with the gather).
pub fn compact(encoded: EncodedBatch, stream: &Stream) -> Result<PackedBatch> {
    let n = encoded.n_streams;
    let capacity = n * encoded.stride_words;
    let mut packed: DeviceBuffer<u32> = DeviceBuffer::new(capacity)?;
    let mut offsets: DeviceBuffer<u64> = DeviceBuffer::new(n + 1)?;
    let mut scratch: DeviceBuffer<u8> = DeviceBuffer::new(unsafe { ffi::cst_compact_scratch_bytes(n) })?;
    check(unsafe {
        ffi::cst_compact_words(
            encoded.words.as_ptr(),
            encoded.stride_words,
            encoded.n_words.as_ptr(),
            n,
            offsets.as_mut_ptr(),
            packed.as_mut_ptr(),
            capacity,
            scratch.as_mut_ptr() as *mut c_void,
            stream.as_raw(),
        )
    })?;
    stream.synchronize()?; // (the scratch buffer is dropped on return)
    Ok(PackedBatch { words: packed, offsets, n_words: encoded.n_words, n_streams: n, config: encoded.config })
}

/// Every stream's words in the opposite order, in place (its own inverse): between the reference's default order and the one
/// `AnsCoder::from_reversed_compressed` / `Cursor::into_reversed` use (src/stream/stack.rs:734-748, src/backends.rs:1424-1448).
pub fn reverse_words(encoded: &mut EncodedBatch, stream: &Stream) -> Result<()> {
    let words = encoded.words.as_mut_ptr();
    check(unsafe {
        ffi::cst_words_reverse(
            words as *const u32,
            std::ptr::null(),
            encoded.stride_words,
            encoded.n_words.as_ptr(),
            encoded.n_streams,
            words,
            std::ptr::null(),
            encoded.stride_words,
            stream.as_raw(),
        )
    })
}

/// Slabs of PACKED 16-bit words (`CST_FLAG_PACKED_W16`: the `(16, 32)` preset with its words as the reference's `Vec<u16>`,
/// src/stream/stack.rs:153): every count, stride and offset is in 16-bit words.
pub struct EncodedBatch16 {
    pub words: DeviceBuffer<u16>,
    pub n_words: DeviceBuffer<u32>,
    pub status: DeviceBuffer<i32>,
    pub n_streams: usize,
    pub stride_words: usize,
    pub config: CoderConfig,
}

/// `EncodedBatch16` -> the concatenation of every stream's `Vec<u16>` plus `offsets[n_streams + 1]` (in 16-bit words).
pub fn compact16(encoded: &EncodedBatch16, stream: &Stream) -> Result<(DeviceBuffer<u16>, DeviceBuffer<u64>)> {
    let n = encoded.n_streams;
    let capacity = n.checked_mul(encoded.stride_words).ok_or(Error::InvalidArgument)?;
    let mut packed: DeviceBuffer<u16> = DeviceBuffer::new(capacity)?;
    let mut offsets: DeviceBuffer<u64> = DeviceBuffer::new(n + 1)?;
    let mut scratch: DeviceBuffer<u8> = DeviceBuffer::new(unsafe { ffi::cst_compact_scratch_bytes(n) })?;
    check(unsafe {
        ffi::cst_compact_words16(
            encoded.words.as_ptr(),
            encoded.stride_words,
            encoded.n_words.as_ptr(),
            n,
            offsets.as_mut_ptr(),
            packed.as_mut_ptr(),
            capacity,
            scratch.as_mut_ptr() as *mut c_void,
            stream.as_raw(),
        )
    })?;
    stream.synchronize()?; // (the scratch buffer is dropped on return)
    Ok((packed, offsets))
}

/// Where a decoder finds the words of stream `s`.
enum WordSource<'a> {
    Slabs(&'a EncodedBatch),
    Packed(&'a PackedBatch),
}

impl<'a> WordSource<'a> {
    fn words(&self) -> *const u32 {
        match self {
            WordSource::Slabs(e) => e.words.as_ptr(),
            WordSource::Packed(p) => p.words.as_ptr(),
        }
    }
    fn offsets(&self) -> *const u64 {
        match self {
            WordSource::Slabs(_) => core::ptr::null(),
            WordSource::Packed(p) => p.offsets.as_ptr(),
        }
    }
    fn stride(&self) -> usize {
        match self {
            WordSource::Slabs(e) => e.stride_words,
            WordSource::Packed(_) => 0,
        }
    }
    fn capacity(&self) -> usize {
        match self {
            WordSource::Slabs(e) => e.words.len(),
            WordSource::Packed(p) => p.words.len(),
        }
    }
    fn n_words(&self) -> *const u32 {
        match self {
            WordSource::Slabs(e) => e.n_words.as_ptr(),
            WordSource::Packed(p) => p.n_words.as_ptr(),
        }
    }
    fn n_streams(&self) -> usize {
        match self {
            WordSource::Slabs(e) => e.n_streams,
            WordSource::Packed(p) => p.n_streams,
        }
    }
}

/// Decoded symbols of a batch and the status of every stream.
pub struct DecodedBatch {
    pub symbols: DeviceBuffer<i32>,
    pub status: DeviceBuffer<i32>,
}

impl DecodedBatch {
    /// The per-chunk status of a `*_from_checkpoints` decode (`n_streams * n_chunks` entries) as one status per stream -- the worst of
    /// its chunks, what the plain decoder of the whole stream reports (`cst_ckpt_status_per_stream`).
    pub fn status_per_stream(&self, n_streams: usize, n_chunks: usize, stream: &Stream) -> Result<DeviceBuffer<i32>> {
        if n_chunks == 0 || self.status.len() < n_streams.checked_mul(n_chunks).ok_or(Error::InvalidArgument)? {
            return Err(Error::InvalidArgument);
        }
        let mut out: DeviceBuffer<i32> = DeviceBuffer::new(n_streams)?;
        check(unsafe { ffi::cst_ckpt_status_per_stream(self.status.as_ptr(), n_streams, n_chunks, out.as_mut_ptr(), stream.as_raw()) })?;
        Ok(out)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ANS: many `AnsCoder<Word, State>` at once
// ---------------------------------------------------------------------------------------------------------------------

/// One independent `AnsCoder` per stream of a batch.  Methods are named after the reference's (`stream::stack::AnsCoder`).
#[derive(Clone, Copy, Debug)]
pub struct BatchedAnsCoder {
    pub config: CoderConfig,
    pub layout: Layout,
}

impl BatchedAnsCoder {
    pub fn new(config: CoderConfig) -> Self {
        BatchedAnsCoder { config, layout: Layout::StreamMajor }
    }

    pub fn with_layout(self, layout: Layout) -> Self {
        BatchedAnsCoder { layout, ..self }
    }

    /// Upper bound on `num_words()` of one stream of `n_symbols` symbols: the slab stride.
    pub fn max_words(&self, n_symbols: usize) -> usize {
        unsafe { ffi::cst_ans_max_words(n_symbols, self.config) }
    }

    /// The library's own choice of jump points for a batch about to be encoded with `encode_iid_symbols_reverse*` (ABI 5,
    /// `cst_jump_points_auto`): the `interval` to hand to `encode_iid_symbols_reverse_with_checkpoints[_narrow]` and to the matching
    /// `decode_iid_symbols_from_checkpoints*`, or 0 = the plain calls are as fast.  Jump points (`Pos` / `Seek`,
    /// src/stream/stack.rs:1107-1139) never change the words.  `symbol_bytes` = 4 for `i32` matrices, 1 / 2 for `i8` / `i16`.
    pub fn auto_jump_interval(&self, symbol_bytes: usize, n_streams: usize, n_per_stream: usize, model: &DeviceModel) -> usize {
        unsafe {
            ffi::cst_jump_points_auto(
                model.as_raw(),
                self.config,
                ffi::CST_CODER_ANS,
                symbol_bytes as i32,
                core::ptr::null(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                core::ptr::null(),
                self.max_words(n_per_stream),
            )
        }
    }

    /// ... for `encode_symbols_reverse_with_checkpoints` (every symbol its own mean and std; `cst_jump_points_auto_gaussian`).
    pub fn auto_jump_interval_gaussian(&self, n_streams: usize, n_per_stream: usize) -> usize {
        unsafe { ffi::cst_jump_points_auto_gaussian(self.config, ffi::CST_CODER_ANS, n_streams, n_per_stream, self.layout.raw()) }
    }

    /// Per stream: `AnsCoder::new()`, `encode_iid_symbols_reverse(symbols[s], &model)?`, `into_compressed()`
    /// (src/stream/stack.rs:249, 835-849, 891-895).
    pub fn encode_iid_symbols_reverse(
        &self,
        symbols: &DeviceBuffer<i32>,
        n_streams: usize,
        n_per_stream: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<EncodedBatch> {
        if symbols.len() < n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)? {
            return Err(Error::InvalidArgument);
        }
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        check(unsafe {
            ffi::cst_ans_encode_batch(
                model.as_raw(),
                self.config,
                symbols.as_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    /// `SmallAnsCoder::encode_iid_symbols_reverse` + `into_compressed()` with the words PACKED as the reference holds them -- a
    /// `Vec<u16>` per stream (src/stream/stack.rs:153), two words per 32-bit slot (`CST_FLAG_PACKED_W16`; `self.config` must be
    /// the `(16, 32)` preset).  Half the word bytes of the unpacked form cross HBM.
    pub fn encode_iid_symbols_reverse_packed16(
        &self,
        symbols: &DeviceBuffer<i32>,
        n_streams: usize,
        n_per_stream: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<EncodedBatch16> {
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if symbols.len() < count || self.config.word_bits != 16 {
            return Err(Error::InvalidArgument);
        }
        let stride = self.max_words(n_per_stream);
        let mut out = EncodedBatch16 {
            words: DeviceBuffer::new(n_streams.checked_mul(stride).ok_or(Error::InvalidArgument)?)?,
            n_words: DeviceBuffer::new(n_streams)?,
            status: DeviceBuffer::new(n_streams)?,
            n_streams,
            stride_words: stride,
            config: self.config,
        };
        check(unsafe {
            ffi::cst_ans_encode_batch(
                model.as_raw(),
                self.config,
                symbols.as_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr() as *mut u32,
                stride,
                out.n_words.as_mut_ptr(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_PACKED_W16,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    /// `encode_iid_symbols_reverse_packed16` + `AnsCoder::pos()` (`Pos`, src/stream/stack.rs:1107-1139) in front of every chunk of
    /// `interval` symbols: positions count 16-bit words, the words are those of the plain call (chunks of whole 32-symbol tiles that
    /// divide the rows).  Decode from the points = `cst_ans_decode_batch` on the chunks as streams of their own with raw states
    /// (hand it a COPY of the states).
    pub fn encode_iid_symbols_reverse_packed16_with_checkpoints(
        &self,
        symbols: &DeviceBuffer<i32>,
        n_streams: usize,
        n_per_stream: usize,
        interval: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<(EncodedBatch16, Checkpoints)> {
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if symbols.len() < count || self.config.word_bits != 16 || interval == 0 || n_per_stream % interval != 0 {
            return Err(Error::InvalidArgument);
        }
        let n_points = n_streams.checked_mul(n_per_stream / interval).ok_or(Error::InvalidArgument)?;
        let stride = self.max_words(n_per_stream);
        let mut out = EncodedBatch16 {
            words: DeviceBuffer::new(n_streams.checked_mul(stride).ok_or(Error::InvalidArgument)?)?,
            n_words: DeviceBuffer::new(n_streams)?,
            status: DeviceBuffer::new(n_streams)?,
            n_streams,
            stride_words: stride,
            config: self.config,
        };
        let mut ckpt = Checkpoints { pos: DeviceBuffer::new(n_points)?, state: DeviceBuffer::new(n_points)?, interval };
        check(unsafe {
            ffi::cst_ans_encode_batch_ckpt_packed16(
                model.as_raw(),
                self.config,
                symbols.as_ptr(),
                n_streams,
                n_per_stream,
                out.words.as_mut_ptr(),
                stride,
                out.n_words.as_mut_ptr(),
                interval,
                ckpt.pos.as_mut_ptr(),
                ckpt.state.as_mut_ptr(),
                out.status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok((out, ckpt))
    }

    /// `AnsCoder::seek(pos, state)` + `interval` symbols for every chunk of an `EncodedBatch16` (src/stream/stack.rs:1117-1128): the
    /// chunks are decoded as `n_streams * n_chunks` streams of their own by the plain batched decoder -- word offsets (in 16-bit words)
    /// = their stream's slab, counts = the jump points' word counts, raw states.  The states are COPIED first: with
    /// `CST_FLAG_RAW_STATE` the decoder leaves its final states in that array, and a jump table is side information that outlives
    /// a decode.  `status` has one entry per chunk (`DecodedBatch::status_per_stream` folds them); a jump point that claims more
    /// words than its stream holds is refused up front (`InvalidArgument` after a host read of the counts).
    pub fn decode_iid_symbols_packed16_from_checkpoints(
        &self,
        encoded: &EncodedBatch16,
        checkpoints: &Checkpoints,
        n_per_stream: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<DecodedBatch> {
        let n_streams = encoded.n_streams;
        let n_chunks = checkpoints.chunks_for(n_streams, n_per_stream)?;
        let n_virtual = n_streams.checked_mul(n_chunks).ok_or(Error::InvalidArgument)?;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        stream.synchronize()?;
        let pos = checkpoints.pos.to_vec_prefix(n_virtual)?;
        let n_words = encoded.n_words.to_vec_prefix(n_streams)?;
        let mut offsets: Vec<u64> = Vec::with_capacity(n_virtual);
        for s in 0..n_streams {
            for j in 0..n_chunks {
                if pos[s * n_chunks + j] > n_words[s] || pos[s * n_chunks + j] as usize > encoded.stride_words {
                    return Err(Error::InvalidArgument);
                }
                offsets.push((s * encoded.stride_words) as u64);
            }
        }
        let d_offsets = DeviceBuffer::from_slice(&offsets)?;
        let mut states = DeviceBuffer::from_slice(&checkpoints.state.to_vec_prefix(n_virtual)?)?; // a COPY: in and out
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(count)?, status: DeviceBuffer::new(n_virtual)? };
        check(unsafe {
            ffi::cst_ans_decode_batch(
                model.as_raw(),
                self.config,
                encoded.words.as_ptr() as *const u32,
                d_offsets.as_ptr(),
                0,
                encoded.words.len(),
                checkpoints.pos.as_ptr(),
                out.symbols.as_mut_ptr(),
                n_virtual,
                checkpoints.interval,
                ffi::CST_LAYOUT_STREAM_MAJOR,
                states.as_mut_ptr(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_PACKED_W16 | ffi::CST_FLAG_RAW_STATE,
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?; // (the offsets and the copied states are dropped on return)
        Ok(out)
    }

    /// The decoder of `encode_iid_symbols_reverse_packed16`.
    pub fn decode_iid_symbols_packed16(&self, encoded: &EncodedBatch16, n_per_stream: usize, model: &DeviceModel, stream: &Stream) -> Result<DecodedBatch> {
        let n_streams = encoded.n_streams;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(count)?, status: DeviceBuffer::new(n_streams)? };
        check(unsafe {
            ffi::cst_ans_decode_batch(
                model.as_raw(),
                self.config,
                encoded.words.as_ptr() as *const u32,
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                encoded.n_words.as_ptr(),
                out.symbols.as_mut_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                core::ptr::null_mut(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_PACKED_W16,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    /// `encode_iid_symbols_reverse` for a NARROW symbol type (`i8` / `i16`: the reference's coders are generic over `Symbol`,
    /// src/stream/model/quantize.rs:229-255).  The matrix is widened on the device next to the coder call; what it saves is the
    /// link to the host.  Words, counts and status are those of the `i32` call on the widened values.
    pub fn encode_iid_symbols_reverse_narrow<T: NarrowSymbol>(
        &self,
        symbols: &DeviceBuffer<T>,
        n_streams: usize,
        n_per_stream: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<EncodedBatch> {
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if symbols.len() < count {
            return Err(Error::InvalidArgument);
        }
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        let mut scratch: DeviceBuffer<u8> = DeviceBuffer::new(unsafe { ffi::cst_symbols_scratch_bytes(n_streams, n_per_stream, T::BYTES) })?;
        check(unsafe {
            ffi::cst_ans_encode_batch_sym(
                model.as_raw(),
                self.config,
                symbols.as_ptr() as *const c_void,
                T::BYTES,
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                scratch.as_mut_ptr() as *mut c_void,
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?; // (the scratch buffer is dropped on return)
        Ok(out)
    }

    /// `decode_iid_symbols` into a NARROW symbol type; the model's support must fit it (`InvalidArgument` otherwise).
    pub fn decode_iid_symbols_narrow<T: NarrowSymbol>(
        &self,
        encoded: &EncodedBatch,
        n_per_stream: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<(DeviceBuffer<T>, DeviceBuffer<i32>)> {
        let n_streams = encoded.n_streams;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        let mut symbols: DeviceBuffer<T> = DeviceBuffer::new(count)?;
        let mut status: DeviceBuffer<i32> = DeviceBuffer::new(n_streams)?;
        let mut scratch: DeviceBuffer<u8> = DeviceBuffer::new(unsafe { ffi::cst_symbols_scratch_bytes(n_streams, n_per_stream, T::BYTES) })?;
        check(unsafe {
            ffi::cst_ans_decode_batch_sym(
                model.as_raw(),
                self.config,
                encoded.words.as_ptr(),
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                encoded.n_words.as_ptr(),
                symbols.as_mut_ptr() as *mut c_void,
                T::BYTES,
                n_streams,
                n_per_stream,
                self.layout.raw(),
                core::ptr::null_mut(),
                core::ptr::null_mut(),
                status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                scratch.as_mut_ptr() as *mut c_void,
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?;
        Ok((symbols, status))
    }

    /// Per stream: `encode_symbols_reverse(symbols[s].zip(models))` with one leakily quantized Gaussian per SYMBOL --
    /// the reference's flagship call (src/stream/stack.rs:784-797; Python `encode_reverse(symbols, QuantizedGaussian(lo,
    /// hi), means, stds)`, src/pybindings/stream/stack.rs:567-588).  `means` / `stds` have the shape of `symbols`.
    pub fn encode_symbols_reverse(
        &self,
        symbols: &DeviceBuffer<i32>,
        support: RangeInclusive<i32>,
        means: &DeviceBuffer<f64>,
        stds: &DeviceBuffer<f64>,
        n_streams: usize,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<EncodedBatch> {
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if symbols.len() < count || means.len() < count || stds.len() < count {
            return Err(Error::InvalidArgument);
        }
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        check(unsafe {
            ffi::cst_ans_encode_gaussian_batch(
                self.config,
                *support.start(),
                *support.end(),
                symbols.as_ptr(),
                means.as_ptr(),
                stds.as_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    /// `encode_symbols_reverse` + `pos()` in front of every chunk of `interval` symbols (a multiple of 16 that divides the row
    /// length; batches of at least 16 384 streams): the jump table of the reference's flagship call.
    pub fn encode_symbols_reverse_with_checkpoints(
        &self,
        symbols: &DeviceBuffer<i32>,
        support: RangeInclusive<i32>,
        means: &DeviceBuffer<f64>,
        stds: &DeviceBuffer<f64>,
        n_streams: usize,
        n_per_stream: usize,
        interval: usize,
        stream: &Stream,
    ) -> Result<(EncodedBatch, Checkpoints)> {
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if interval == 0 || symbols.len() < count || means.len() < count || stds.len() < count {
            return Err(Error::InvalidArgument);
        }
        let n_points = n_streams.checked_mul((n_per_stream + interval - 1) / interval).ok_or(Error::InvalidArgument)?;
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        let mut ckpt = Checkpoints { pos: DeviceBuffer::new(n_points)?, state: DeviceBuffer::new(n_points)?, interval };
        check(unsafe {
            ffi::cst_ans_encode_gaussian_batch_ckpt(
                self.config,
                *support.start(),
                *support.end(),
                symbols.as_ptr(),
                means.as_ptr(),
                stds.as_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                interval,
                ckpt.pos.as_mut_ptr(),
                ckpt.state.as_mut_ptr(),
                out.status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok((out, ckpt))
    }

    /// `seek(pos, state)` + `decode_symbols` of one chunk's models, for every chunk at once: two resident waves per SIMD where
    /// the plain per-symbol decoder of a 65 536-stream batch has one.
    pub fn decode_symbols_from_checkpoints(
        &self,
        encoded: &EncodedBatch,
        checkpoints: &Checkpoints,
        support: RangeInclusive<i32>,
        means: &DeviceBuffer<f64>,
        stds: &DeviceBuffer<f64>,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<DecodedBatch> {
        let n_streams = encoded.n_streams;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        let n_chunks = checkpoints.chunks_for(n_streams, n_per_stream)?;
        if means.len() < count || stds.len() < count {
            return Err(Error::InvalidArgument);
        }
        let n_points = n_streams * n_chunks;
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(count)?, status: DeviceBuffer::new(n_points)? };
        let mut scratch: DeviceBuffer<u8> = DeviceBuffer::new(unsafe { ffi::cst_ckpt_scratch_bytes(n_streams, n_per_stream, checkpoints.interval) })?;
        check(unsafe {
            ffi::cst_ans_decode_gaussian_batch_ckpt(
                self.config,
                *support.start(),
                *support.end(),
                encoded.words.as_ptr(),
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                checkpoints.interval,
                checkpoints.pos.as_ptr(),
                checkpoints.state.as_ptr(),
                means.as_ptr(),
                stds.as_ptr(),
                out.symbols.as_mut_ptr(),
                n_streams,
                n_per_stream,
                scratch.as_mut_ptr() as *mut c_void,
                out.status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?; // (the scratch buffer is dropped on return)
        Ok(out)
    }

    /// `encode_symbols_reverse` with the models given as `(left_cumulative, probability)` per symbol (any family).
    pub fn encode_symbols_reverse_with_cp(
        &self,
        left: &DeviceBuffer<u32>,
        prob: &DeviceBuffer<u32>,
        n_streams: usize,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<EncodedBatch> {
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if left.len() < count || prob.len() < count {
            return Err(Error::InvalidArgument);
        }
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        check(unsafe {
            ffi::cst_ans_encode_cp_batch(
                self.config,
                left.as_ptr(),
                prob.as_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    fn decode_iid(&self, src: WordSource, n_per_stream: usize, model: &DeviceModel, flags: u32, stream: &Stream) -> Result<DecodedBatch> {
        let n_streams = src.n_streams();
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?)?, status: DeviceBuffer::new(n_streams)? };
        check(unsafe {
            ffi::cst_ans_decode_batch(
                model.as_raw(),
                self.config,
                src.words(),
                src.offsets(),
                src.stride(),
                src.capacity(),
                src.n_words(),
                out.symbols.as_mut_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                core::ptr::null_mut(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                flags,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    /// Per stream: `AnsCoder::from_compressed(words[s])?.decode_iid_symbols(n_per_stream, &model)`
    /// (src/stream/stack.rs:299-318, 440-462; src/stream/mod.rs:1016-1031).
    pub fn decode_iid_symbols(&self, encoded: &EncodedBatch, n_per_stream: usize, model: &DeviceModel, stream: &Stream) -> Result<DecodedBatch> {
        self.decode_iid(WordSource::Slabs(encoded), n_per_stream, model, ffi::CST_FLAG_NONE, stream)
    }

    /// The same from a packed batch (e.g. one that arrived through [`Communicator::scatter`] or from the host: words that are
    /// not in the GPU's caches, which is what `CST_FLAG_COLD_WORDS` tells the decoder).
    pub fn decode_iid_symbols_packed(&self, packed: &PackedBatch, n_per_stream: usize, model: &DeviceModel, stream: &Stream) -> Result<DecodedBatch> {
        self.decode_iid(WordSource::Packed(packed), n_per_stream, model, ffi::CST_FLAG_COLD_WORDS, stream)
    }

    /// Per stream: `decode_symbols(models)` with one quantized Gaussian per symbol (src/stream/mod.rs:893-908; Python
    /// `decode(QuantizedGaussian(lo, hi), means, stds)`, src/pybindings/stream/stack.rs:733-751).
    pub fn decode_symbols(
        &self,
        encoded: &EncodedBatch,
        support: RangeInclusive<i32>,
        means: &DeviceBuffer<f64>,
        stds: &DeviceBuffer<f64>,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<DecodedBatch> {
        let n_streams = encoded.n_streams;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if means.len() < count || stds.len() < count {
            return Err(Error::InvalidArgument);
        }
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(count)?, status: DeviceBuffer::new(n_streams)? };
        check(unsafe {
            ffi::cst_ans_decode_gaussian_batch(
                self.config,
                *support.start(),
                *support.end(),
                encoded.words.as_ptr(),
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                encoded.n_words.as_ptr(),
                means.as_ptr(),
                stds.as_ptr(),
                out.symbols.as_mut_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                core::ptr::null_mut(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    /// `decode_symbols` with one explicit cdf row (`n_symbols + 1` cumulatives) per coded symbol.
    pub fn decode_symbols_with_cdf_rows(
        &self,
        encoded: &EncodedBatch,
        cdf_rows: &DeviceBuffer<u32>,
        n_symbols: usize,
        min_symbol: i32,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<DecodedBatch> {
        let n_streams = encoded.n_streams;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if cdf_rows.len() < count * (n_symbols + 1) {
            return Err(Error::InvalidArgument);
        }
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(count)?, status: DeviceBuffer::new(n_streams)? };
        check(unsafe {
            ffi::cst_ans_decode_rows_batch(
                self.config,
                encoded.words.as_ptr(),
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                encoded.n_words.as_ptr(),
                cdf_rows.as_ptr(),
                n_symbols as i32,
                min_symbol,
                out.symbols.as_mut_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                core::ptr::null_mut(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    /// Streams of different lengths in one launch -- one `DefaultAnsCoder` per document with a shared model, the
    /// reference's tests/issue52.rs:27-60.  `sym_offsets[n_streams + 1]` delimits the symbols of every stream;
    /// `word_offsets[n_streams + 1]` its slab (`max_words(length)` words always suffice).
    ///
    /// # Safety
    /// The offsets are DEVICE memory: this wrapper cannot check them.  Every `sym_offsets` pair must lie inside `symbols`, every
    /// slab `[word_offsets[s], word_offsets[s + 1])` inside `words`, in ascending order.
    pub unsafe fn encode_ragged(
        &self,
        symbols: &DeviceBuffer<i32>,
        sym_offsets: &DeviceBuffer<u64>,
        word_offsets: &DeviceBuffer<u64>,
        words: &mut DeviceBuffer<u32>,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<(DeviceBuffer<u32>, DeviceBuffer<i32>)> {
        if sym_offsets.is_empty() || word_offsets.len() != sym_offsets.len() {
            return Err(Error::InvalidArgument);
        }
        let n_streams = sym_offsets.len() - 1;
        let mut n_words: DeviceBuffer<u32> = DeviceBuffer::new(n_streams)?;
        let mut status: DeviceBuffer<i32> = DeviceBuffer::new(n_streams)?;
        check(unsafe {
            ffi::cst_ans_encode_ragged(
                model.as_raw(),
                self.config,
                symbols.as_ptr(),
                sym_offsets.as_ptr(),
                n_streams,
                words.as_mut_ptr(),
                word_offsets.as_ptr(),
                0,
                n_words.as_mut_ptr(),
                status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok((n_words, status))
    }

    /// The decoder of [`BatchedAnsCoder::encode_ragged`] (tests/issue52.rs:63-80 with known lengths).
    ///
    /// # Safety
    /// As for `encode_ragged`: `sym_offsets` (device memory) must delimit ranges inside `symbols`.  The word slices ARE checked
    /// on the device against `words.len()` (`words_capacity` of the C ABI).
    pub unsafe fn decode_ragged(
        &self,
        words: &DeviceBuffer<u32>,
        word_offsets: &DeviceBuffer<u64>,
        n_words: &DeviceBuffer<u32>,
        sym_offsets: &DeviceBuffer<u64>,
        symbols: &mut DeviceBuffer<i32>,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<DeviceBuffer<i32>> {
        if sym_offsets.is_empty() || word_offsets.len() < sym_offsets.len() - 1 || n_words.len() < sym_offsets.len() - 1 {
            return Err(Error::InvalidArgument);
        }
        let n_streams = sym_offsets.len() - 1;
        let mut status: DeviceBuffer<i32> = DeviceBuffer::new(n_streams)?;
        check(unsafe {
            ffi::cst_ans_decode_ragged(
                model.as_raw(),
                self.config,
                words.as_ptr(),
                word_offsets.as_ptr(),
                0,
                words.len(),
                n_words.as_ptr(),
                symbols.as_mut_ptr(),
                sym_offsets.as_ptr(),
                n_streams,
                status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok(status)
    }

    /// `encode_ragged` that also notes `AnsCoder::pos()` in front of every `jump_interval` symbols of every stream (ABI 5,
    /// `cst_ans_encode_ragged_jump`; `Pos`, src/stream/stack.rs:1107-1116): chunk `j` of stream `s` is entry `chunk_offsets[s] + j` of
    /// the returned `(pos, state)` arrays, `chunk_offsets[n_streams + 1]` the exclusive prefix sum of `ceil(length / jump_interval)`
    /// (device memory, `n_chunks_total` its last entry).  The words are those of `encode_ragged`.
    ///
    /// # Safety
    /// As for `encode_ragged`; `chunk_offsets` must be that prefix sum.
    pub unsafe fn encode_ragged_with_jump_points(
        &self,
        symbols: &DeviceBuffer<i32>,
        sym_offsets: &DeviceBuffer<u64>,
        word_offsets: &DeviceBuffer<u64>,
        words: &mut DeviceBuffer<u32>,
        jump_interval: usize,
        chunk_offsets: &DeviceBuffer<u64>,
        n_chunks_total: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<(DeviceBuffer<u32>, DeviceBuffer<i32>, DeviceBuffer<u32>, DeviceBuffer<u64>)> {
        if sym_offsets.is_empty() || word_offsets.len() != sym_offsets.len() || chunk_offsets.len() != sym_offsets.len() || jump_interval == 0 || jump_interval % 8 != 0 {
            return Err(Error::InvalidArgument);
        }
        let n_streams = sym_offsets.len() - 1;
        let mut n_words: DeviceBuffer<u32> = DeviceBuffer::new(n_streams)?;
        let mut status: DeviceBuffer<i32> = DeviceBuffer::new(n_streams)?;
        let mut pos: DeviceBuffer<u32> = DeviceBuffer::new(n_chunks_total.max(1))?;
        let mut state: DeviceBuffer<u64> = DeviceBuffer::new(n_chunks_total.max(1))?;
        check(unsafe {
            ffi::cst_ans_encode_ragged_jump(
                model.as_raw(),
                self.config,
                symbols.as_ptr(),
                sym_offsets.as_ptr(),
                n_streams,
                core::ptr::null(),
                words.as_mut_ptr(),
                word_offsets.as_ptr(),
                0,
                n_words.as_mut_ptr(),
                jump_interval,
                chunk_offsets.as_ptr(),
                pos.as_mut_ptr(),
                state.as_mut_ptr(),
                status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok((n_words, status, pos, state))
    }

    /// The decoder of `encode_ragged_with_jump_points`: every chunk of every stream as a coder of its own (`AnsCoder::seek` + at most
    /// `jump_interval` symbols), so that a launch lasts as long as a chunk, not as its longest document.  One status per stream.
    ///
    /// # Safety
    /// As for `decode_ragged`.  The jump table is checked against the lengths on the device.
    pub unsafe fn decode_ragged_from_jump_points(
        &self,
        words: &DeviceBuffer<u32>,
        word_offsets: &DeviceBuffer<u64>,
        n_words: &DeviceBuffer<u32>,
        sym_offsets: &DeviceBuffer<u64>,
        symbols: &mut DeviceBuffer<i32>,
        jump_interval: usize,
        chunk_offsets: &DeviceBuffer<u64>,
        pos: &DeviceBuffer<u32>,
        state: &DeviceBuffer<u64>,
        n_chunks_total: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<DeviceBuffer<i32>> {
        if sym_offsets.is_empty() || word_offsets.len() < sym_offsets.len() - 1 || n_words.len() < sym_offsets.len() - 1 {
            return Err(Error::InvalidArgument);
        }
        if chunk_offsets.len() != sym_offsets.len() || pos.len() < n_chunks_total || state.len() < n_chunks_total || jump_interval == 0 || jump_interval % 8 != 0 {
            return Err(Error::InvalidArgument);
        }
        let n_streams = sym_offsets.len() - 1;
        let mut status: DeviceBuffer<i32> = DeviceBuffer::new(n_streams)?;
        let mut scratch: DeviceBuffer<u8> = DeviceBuffer::new(unsafe { ffi::cst_ragged_jump_scratch_bytes(n_chunks_total) })?;
        check(unsafe {
            ffi::cst_ans_decode_ragged_jump(
                model.as_raw(),
                self.config,
                words.as_ptr(),
                word_offsets.as_ptr(),
                0,
                words.len(),
                n_words.as_ptr(),
                symbols.as_mut_ptr(),
                sym_offsets.as_ptr(),
                n_streams,
                jump_interval,
                chunk_offsets.as_ptr(),
                n_chunks_total,
                pos.as_ptr(),
                state.as_ptr(),
                scratch.as_mut_ptr() as *mut c_void,
                status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?; // (the scratch buffer is dropped on return)
        Ok(status)
    }

    /// Encoding with jump tables: what `AnsCoder::pos()` returns in front of every chunk of `interval` symbols
    /// (`Pos`, src/stream/stack.rs:1107-1116); the words are those of `encode_iid_symbols_reverse`.
    pub fn encode_iid_symbols_reverse_with_checkpoints(
        &self,
        symbols: &DeviceBuffer<i32>,
        n_streams: usize,
        n_per_stream: usize,
        interval: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<(EncodedBatch, Checkpoints)> {
        if interval == 0 || symbols.len() < n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)? {
            return Err(Error::InvalidArgument);
        }
        let n_chunks = (n_per_stream + interval - 1) / interval;
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        let mut ckpt = Checkpoints { pos: DeviceBuffer::new(n_streams.checked_mul(n_chunks).ok_or(Error::InvalidArgument)?)?, state: DeviceBuffer::new(n_streams.checked_mul(n_chunks).ok_or(Error::InvalidArgument)?)?, interval };
        check(unsafe {
            ffi::cst_ans_encode_batch_ckpt(
                model.as_raw(),
                self.config,
                symbols.as_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                interval,
                ckpt.pos.as_mut_ptr(),
                ckpt.state.as_mut_ptr(),
                out.status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok((out, ckpt))
    }

    /// `AnsCoder::seek(pos, state)` + `interval` decoded symbols for EVERY chunk at once (`Seek`, stack.rs:1118-1139):
    /// one long stream decodes on as many lanes as it has chunks.
    pub fn decode_iid_symbols_from_checkpoints(
        &self,
        encoded: &EncodedBatch,
        checkpoints: &Checkpoints,
        n_per_stream: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<DecodedBatch> {
        let n_streams = encoded.n_streams;
        let n_chunks = checkpoints.chunks_for(n_streams, n_per_stream)?;
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?)?, status: DeviceBuffer::new(n_streams.checked_mul(n_chunks).ok_or(Error::InvalidArgument)?)? };
        let mut scratch: DeviceBuffer<u8> = DeviceBuffer::new(unsafe { ffi::cst_ckpt_scratch_bytes(n_streams, n_per_stream, checkpoints.interval) })?;
        check(unsafe {
            ffi::cst_ans_decode_batch_ckpt(
                model.as_raw(),
                self.config,
                encoded.words.as_ptr(),
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                checkpoints.interval,
                checkpoints.pos.as_ptr(),
                checkpoints.state.as_ptr(),
                out.symbols.as_mut_ptr(),
                n_streams,
                n_per_stream,
                scratch.as_mut_ptr() as *mut c_void,
                out.status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?; // (the scratch buffer is dropped on return)
        Ok(out)
    }

    /// `encode_iid_symbols_reverse_with_checkpoints` for a NARROW symbol type (round 5: `cst_ans_encode_batch_ckpt_sym`).  An `i8`
    /// matrix of whole 128-symbol lines is read by the encoder loops themselves, jump points noted on the way.
    pub fn encode_iid_symbols_reverse_with_checkpoints_narrow<T: NarrowSymbol>(
        &self,
        symbols: &DeviceBuffer<T>,
        n_streams: usize,
        n_per_stream: usize,
        interval: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<(EncodedBatch, Checkpoints)> {
        if interval == 0 || symbols.len() < n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)? {
            return Err(Error::InvalidArgument);
        }
        let n_chunks = (n_per_stream + interval - 1) / interval;
        let n_points = n_streams.checked_mul(n_chunks).ok_or(Error::InvalidArgument)?;
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        let mut ckpt = Checkpoints { pos: DeviceBuffer::new(n_points)?, state: DeviceBuffer::new(n_points)?, interval };
        let mut scratch: DeviceBuffer<u8> = DeviceBuffer::new(unsafe { ffi::cst_ckpt_sym_scratch_bytes(n_streams, n_per_stream, interval, T::BYTES) })?;
        check(unsafe {
            ffi::cst_ans_encode_batch_ckpt_sym(
                model.as_raw(),
                self.config,
                symbols.as_ptr() as *const c_void,
                T::BYTES,
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                interval,
                ckpt.pos.as_mut_ptr(),
                ckpt.state.as_mut_ptr(),
                out.status.as_mut_ptr(),
                scratch.as_mut_ptr() as *mut c_void,
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?; // (the scratch buffer is dropped on return)
        Ok((out, ckpt))
    }

    /// `decode_iid_symbols_from_checkpoints` into a NARROW symbol type (`cst_ans_decode_batch_ckpt_sym`): every chunk on its own
    /// lane, `i8` chunks of whole 128-symbol lines written by the decoder loops themselves.
    pub fn decode_iid_symbols_from_checkpoints_narrow<T: NarrowSymbol>(
        &self,
        encoded: &EncodedBatch,
        checkpoints: &Checkpoints,
        n_per_stream: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<(DeviceBuffer<T>, DeviceBuffer<i32>)> {
        let n_streams = encoded.n_streams;
        let n_chunks = checkpoints.chunks_for(n_streams, n_per_stream)?;
        let mut symbols: DeviceBuffer<T> = DeviceBuffer::new(n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?)?;
        let mut status: DeviceBuffer<i32> = DeviceBuffer::new(n_streams.checked_mul(n_chunks).ok_or(Error::InvalidArgument)?)?;
        let mut scratch: DeviceBuffer<u8> =
            DeviceBuffer::new(unsafe { ffi::cst_ckpt_sym_scratch_bytes(n_streams, n_per_stream, checkpoints.interval, T::BYTES) })?;
        check(unsafe {
            ffi::cst_ans_decode_batch_ckpt_sym(
                model.as_raw(),
                self.config,
                encoded.words.as_ptr(),
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                checkpoints.interval,
                checkpoints.pos.as_ptr(),
                checkpoints.state.as_ptr(),
                symbols.as_mut_ptr() as *mut c_void,
                T::BYTES,
                n_streams,
                n_per_stream,
                scratch.as_mut_ptr() as *mut c_void,
                status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?;
        Ok((symbols, status))
    }
}

/// `(pos, state)` of every stream in front of every chunk: the reference's jump table.
pub struct Checkpoints {
    pub pos: DeviceBuffer<u32>,
    pub state: DeviceBuffer<u64>,
    pub interval: usize,
}

impl Checkpoints {
    /// Jump points per stream if this table describes `n_streams` streams of `n_per_stream` symbols -- whole chunks, one entry per
    /// (stream, chunk) in both arrays -- else `InvalidArgument`: the C calls index `[n_streams][n_per_stream / interval]` and trust it.
    pub fn chunks_for(&self, n_streams: usize, n_per_stream: usize) -> Result<usize> {
        if self.interval == 0 || n_per_stream % self.interval != 0 {
            return Err(Error::InvalidArgument);
        }
        let n_chunks = n_per_stream / self.interval;
        let n_points = n_streams.checked_mul(n_chunks).ok_or(Error::InvalidArgument)?;
        if self.pos.len() < n_points || self.state.len() < n_points {
            return Err(Error::InvalidArgument);
        }
        Ok(n_chunks)
    }
}

/// `RangeEncoder::pos()` of every stream in front of every chunk (src/stream/queue.rs:172-196).
pub struct RangeCheckpoints {
    pub pos: DeviceBuffer<u32>,
    pub lower: DeviceBuffer<u64>,
    pub range: DeviceBuffer<u64>,
    pub interval: usize,
}

// ---------------------------------------------------------------------------------------------------------------------
// range coder: many `RangeEncoder` / `RangeDecoder` at once (src/stream/queue.rs)
// ---------------------------------------------------------------------------------------------------------------------

/// One `RangeEncoder<Word, State>` per stream; symbols are coded first to last (a queue).
#[derive(Clone, Copy, Debug)]
pub struct BatchedRangeEncoder {
    pub config: CoderConfig,
    pub layout: Layout,
}

impl BatchedRangeEncoder {
    pub fn new(config: CoderConfig) -> Self {
        BatchedRangeEncoder { config, layout: Layout::StreamMajor }
    }

    pub fn max_words(&self, n_symbols: usize) -> usize {
        unsafe { ffi::cst_range_max_words(n_symbols, self.config) }
    }

    /// The library's own choice of jump points (`RangeEncoder::pos`, src/stream/queue.rs:172-196) for a batch about to be encoded:
    /// the `interval` for `encode_iid_symbols_with_checkpoints` / `BatchedRangeDecoder::decode_iid_symbols_from_checkpoints`, or 0.
    pub fn auto_jump_interval(&self, n_streams: usize, n_per_stream: usize, model: &DeviceModel) -> usize {
        unsafe {
            ffi::cst_jump_points_auto(
                model.as_raw(),
                self.config,
                ffi::CST_CODER_RANGE,
                4,
                core::ptr::null(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                core::ptr::null(),
                self.max_words(n_per_stream),
            )
        }
    }

    /// Per stream: `RangeEncoder::new()`, `encode_iid_symbols(symbols[s], &model)?`, `into_compressed()`
    /// (src/stream/queue.rs:612-705, 458-523).
    pub fn encode_iid_symbols(
        &self,
        symbols: &DeviceBuffer<i32>,
        n_streams: usize,
        n_per_stream: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<EncodedBatch> {
        if symbols.len() < n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)? {
            return Err(Error::InvalidArgument);
        }
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        check(unsafe {
            ffi::cst_range_encode_batch(
                model.as_raw(),
                self.config,
                symbols.as_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    /// Encoding with jump tables: what `RangeEncoder::pos()` returns in front of every chunk of `interval` symbols
    /// (`Pos`, src/stream/queue.rs:172-196: words emitted so far including held-back ones, and `RangeCoderState`); the words are
    /// those of `encode_iid_symbols`.
    pub fn encode_iid_symbols_with_checkpoints(
        &self,
        symbols: &DeviceBuffer<i32>,
        n_streams: usize,
        n_per_stream: usize,
        interval: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<(EncodedBatch, RangeCheckpoints)> {
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if interval == 0 || symbols.len() < count {
            return Err(Error::InvalidArgument);
        }
        let n_chunks = (n_per_stream + interval - 1) / interval;
        let n_points = n_streams.checked_mul(n_chunks).ok_or(Error::InvalidArgument)?;
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        let mut ckpt = RangeCheckpoints {
            pos: DeviceBuffer::new(n_points)?,
            lower: DeviceBuffer::new(n_points)?,
            range: DeviceBuffer::new(n_points)?,
            interval,
        };
        check(unsafe {
            ffi::cst_range_encode_batch_ckpt(
                model.as_raw(),
                self.config,
                symbols.as_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                interval,
                ckpt.pos.as_mut_ptr(),
                ckpt.lower.as_mut_ptr(),
                ckpt.range.as_mut_ptr(),
                out.status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok((out, ckpt))
    }

    /// `encode_iid_symbols` for a NARROW symbol type (`i8` / `i16`: `RangeEncoder` is generic over the symbol type,
    /// src/stream/queue.rs:612; ABI 5 `cst_range_encode_batch_sym`).  An `i8` matrix of stream-major rows of whole 32-symbol tiles is
    /// read by the hand-scheduled encoder itself; other shapes are widened on the device next to the `i32` call.  Words, counts and
    /// status are those of the `i32` call on the widened values.
    pub fn encode_iid_symbols_narrow<T: NarrowSymbol>(
        &self,
        symbols: &DeviceBuffer<T>,
        n_streams: usize,
        n_per_stream: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<EncodedBatch> {
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if symbols.len() < count {
            return Err(Error::InvalidArgument);
        }
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        let mut scratch: DeviceBuffer<u8> = DeviceBuffer::new(unsafe { ffi::cst_range_sym_scratch_bytes(n_streams, n_per_stream, 0, T::BYTES) })?;
        check(unsafe {
            ffi::cst_range_encode_batch_sym(
                model.as_raw(),
                self.config,
                symbols.as_ptr() as *const c_void,
                T::BYTES,
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                scratch.as_mut_ptr() as *mut c_void,
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?; // (the scratch buffer is dropped on return)
        Ok(out)
    }

    /// `encode_iid_symbols_with_checkpoints` for a NARROW symbol type (`cst_range_encode_batch_ckpt_sym`): `RangeEncoder::pos()` in front of
    /// every `interval` symbols, noted by the `i8` encoder loop on its way where the chunks are whole tiles.
    pub fn encode_iid_symbols_with_checkpoints_narrow<T: NarrowSymbol>(
        &self,
        symbols: &DeviceBuffer<T>,
        n_streams: usize,
        n_per_stream: usize,
        interval: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<(EncodedBatch, RangeCheckpoints)> {
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if interval == 0 || symbols.len() < count {
            return Err(Error::InvalidArgument);
        }
        let n_chunks = (n_per_stream + interval - 1) / interval;
        let n_points = n_streams.checked_mul(n_chunks).ok_or(Error::InvalidArgument)?;
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        let mut ckpt = RangeCheckpoints {
            pos: DeviceBuffer::new(n_points)?,
            lower: DeviceBuffer::new(n_points)?,
            range: DeviceBuffer::new(n_points)?,
            interval,
        };
        let mut scratch: DeviceBuffer<u8> =
            DeviceBuffer::new(unsafe { ffi::cst_range_sym_scratch_bytes(n_streams, n_per_stream, interval, T::BYTES) })?;
        check(unsafe {
            ffi::cst_range_encode_batch_ckpt_sym(
                model.as_raw(),
                self.config,
                symbols.as_ptr() as *const c_void,
                T::BYTES,
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                interval,
                ckpt.pos.as_mut_ptr(),
                ckpt.lower.as_mut_ptr(),
                ckpt.range.as_mut_ptr(),
                out.status.as_mut_ptr(),
                scratch.as_mut_ptr() as *mut c_void,
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?; // (the scratch buffer is dropped on return)
        Ok((out, ckpt))
    }

    /// Per stream: `encode_symbols(symbols[s].zip(models))` with one quantized Gaussian per symbol (Python
    /// `RangeEncoder.encode(symbols, QuantizedGaussian(lo, hi), means, stds)`, src/pybindings/stream/queue.rs:343-410).
    pub fn encode_symbols(
        &self,
        symbols: &DeviceBuffer<i32>,
        support: RangeInclusive<i32>,
        means: &DeviceBuffer<f64>,
        stds: &DeviceBuffer<f64>,
        n_streams: usize,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<EncodedBatch> {
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if symbols.len() < count || means.len() < count || stds.len() < count {
            return Err(Error::InvalidArgument);
        }
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        check(unsafe {
            ffi::cst_range_encode_gaussian_batch(
                self.config,
                *support.start(),
                *support.end(),
                symbols.as_ptr(),
                means.as_ptr(),
                stds.as_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    /// `encode_symbols` that also notes `RangeEncoder::pos()` in front of every chunk of `interval` symbols (ABI 5,
    /// `cst_range_encode_gaussian_batch_ckpt`; src/stream/queue.rs:172-196): a multiple of 16 that divides `n_per_stream`, stream-major.
    /// The words are those of `encode_symbols`.  `auto_jump_interval_gaussian` is the library's own choice of `interval`.
    pub fn encode_symbols_with_checkpoints(
        &self,
        symbols: &DeviceBuffer<i32>,
        support: RangeInclusive<i32>,
        means: &DeviceBuffer<f64>,
        stds: &DeviceBuffer<f64>,
        n_streams: usize,
        n_per_stream: usize,
        interval: usize,
        stream: &Stream,
    ) -> Result<(EncodedBatch, RangeCheckpoints)> {
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if interval == 0 || n_per_stream % interval != 0 || symbols.len() < count || means.len() < count || stds.len() < count {
            return Err(Error::InvalidArgument);
        }
        let n_points = n_streams.checked_mul(n_per_stream / interval).ok_or(Error::InvalidArgument)?;
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        let mut ckpt = RangeCheckpoints {
            pos: DeviceBuffer::new(n_points)?,
            lower: DeviceBuffer::new(n_points)?,
            range: DeviceBuffer::new(n_points)?,
            interval,
        };
        check(unsafe {
            ffi::cst_range_encode_gaussian_batch_ckpt(
                self.config,
                *support.start(),
                *support.end(),
                symbols.as_ptr(),
                means.as_ptr(),
                stds.as_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                interval,
                ckpt.pos.as_mut_ptr(),
                ckpt.lower.as_mut_ptr(),
                ckpt.range.as_mut_ptr(),
                out.status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok((out, ckpt))
    }

    /// The library's own choice of jump points for `encode_symbols_with_checkpoints` (`cst_jump_points_auto_gaussian`), or 0.
    pub fn auto_jump_interval_gaussian(&self, n_streams: usize, n_per_stream: usize) -> usize {
        unsafe { ffi::cst_jump_points_auto_gaussian(self.config, ffi::CST_CODER_RANGE, n_streams, n_per_stream, self.layout.raw()) }
    }

    /// `encode_symbols` with explicit `(left_cumulative, probability)` per symbol.
    pub fn encode_symbols_with_cp(
        &self,
        left: &DeviceBuffer<u32>,
        prob: &DeviceBuffer<u32>,
        n_streams: usize,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<EncodedBatch> {
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if left.len() < count || prob.len() < count {
            return Err(Error::InvalidArgument);
        }
        let mut out = EncodedBatch::allocate(n_streams, self.max_words(n_per_stream), self.config)?;
        check(unsafe {
            ffi::cst_range_encode_cp_batch(
                self.config,
                left.as_ptr(),
                prob.as_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                out.words.as_mut_ptr(),
                out.stride_words,
                out.n_words.as_mut_ptr(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }
}

/// One `RangeDecoder<Word, State>` per stream.
#[derive(Clone, Copy, Debug)]
pub struct BatchedRangeDecoder {
    pub config: CoderConfig,
    pub layout: Layout,
}

impl BatchedRangeDecoder {
    pub fn new(config: CoderConfig) -> Self {
        BatchedRangeDecoder { config, layout: Layout::StreamMajor }
    }

    /// `decode_iid_symbols` into a NARROW symbol type (`cst_range_decode_batch_sym`; `RangeDecoder` is generic over the symbol type,
    /// src/stream/queue.rs:968).  The model's support must fit the type.  `i8` stream-major matrices are written by the decoder itself.
    pub fn decode_iid_symbols_narrow<T: NarrowSymbol>(
        &self,
        encoded: &EncodedBatch,
        n_per_stream: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<(DeviceBuffer<T>, DeviceBuffer<i32>)> {
        let n_streams = encoded.n_streams;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        let mut symbols: DeviceBuffer<T> = DeviceBuffer::new(count)?;
        let mut status: DeviceBuffer<i32> = DeviceBuffer::new(n_streams)?;
        let mut scratch: DeviceBuffer<u8> = DeviceBuffer::new(unsafe { ffi::cst_range_sym_scratch_bytes(n_streams, n_per_stream, 0, T::BYTES) })?;
        check(unsafe {
            ffi::cst_range_decode_batch_sym(
                model.as_raw(),
                self.config,
                encoded.words.as_ptr(),
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                encoded.n_words.as_ptr(),
                symbols.as_mut_ptr() as *mut c_void,
                T::BYTES,
                n_streams,
                n_per_stream,
                self.layout.raw(),
                core::ptr::null_mut(),
                status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                scratch.as_mut_ptr() as *mut c_void,
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?;
        Ok((symbols, status))
    }

    /// `decode_iid_symbols_from_checkpoints` into a NARROW symbol type (`cst_range_decode_batch_ckpt_sym`): every chunk on a lane of its
    /// own, `i8` matrices written by the sub-lane decoder itself.  Returns the symbols and one status per (stream, chunk).
    pub fn decode_iid_symbols_from_checkpoints_narrow<T: NarrowSymbol>(
        &self,
        encoded: &EncodedBatch,
        checkpoints: &RangeCheckpoints,
        n_per_stream: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<(DeviceBuffer<T>, DeviceBuffer<i32>)> {
        let n_streams = encoded.n_streams;
        if checkpoints.interval == 0 || n_per_stream % checkpoints.interval != 0 {
            return Err(Error::InvalidArgument); // whole chunks only: the C call indexes [n_streams][n_per_stream / interval]
        }
        let n_chunks = n_per_stream / checkpoints.interval;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        let n_points = n_streams.checked_mul(n_chunks).ok_or(Error::InvalidArgument)?;
        if checkpoints.pos.len() < n_points || checkpoints.lower.len() < n_points || checkpoints.range.len() < n_points {
            return Err(Error::InvalidArgument);
        }
        let mut symbols: DeviceBuffer<T> = DeviceBuffer::new(count)?;
        let mut status: DeviceBuffer<i32> = DeviceBuffer::new(n_points)?;
        let mut scratch: DeviceBuffer<u8> =
            DeviceBuffer::new(unsafe { ffi::cst_range_sym_scratch_bytes(n_streams, n_per_stream, checkpoints.interval, T::BYTES) })?;
        check(unsafe {
            ffi::cst_range_decode_batch_ckpt_sym(
                model.as_raw(),
                self.config,
                encoded.words.as_ptr(),
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                encoded.n_words.as_ptr(),
                checkpoints.interval,
                checkpoints.pos.as_ptr(),
                checkpoints.lower.as_ptr(),
                checkpoints.range.as_ptr(),
                symbols.as_mut_ptr() as *mut c_void,
                T::BYTES,
                n_streams,
                n_per_stream,
                scratch.as_mut_ptr() as *mut c_void,
                status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?;
        Ok((symbols, status))
    }

    /// Per stream: `RangeDecoder::from_compressed(words[s])?.decode_iid_symbols(n_per_stream, &model)`
    /// (src/stream/queue.rs:847-868, 968-1033); `InvalidData` is reported per stream.
    pub fn decode_iid_symbols(&self, encoded: &EncodedBatch, n_per_stream: usize, model: &DeviceModel, stream: &Stream) -> Result<DecodedBatch> {
        let n_streams = encoded.n_streams;
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?)?, status: DeviceBuffer::new(n_streams)? };
        check(unsafe {
            ffi::cst_range_decode_batch(
                model.as_raw(),
                self.config,
                encoded.words.as_ptr(),
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                encoded.n_words.as_ptr(),
                out.symbols.as_mut_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    /// `RangeDecoder::seek((pos, state))` + `interval` decoded symbols for EVERY chunk at once (`Seek`, src/stream/queue.rs:900-926):
    /// k lanes per stream -- two resident waves per SIMD where the plain decoder has one.
    pub fn decode_iid_symbols_from_checkpoints(
        &self,
        encoded: &EncodedBatch,
        checkpoints: &RangeCheckpoints,
        n_per_stream: usize,
        model: &DeviceModel,
        stream: &Stream,
    ) -> Result<DecodedBatch> {
        let n_streams = encoded.n_streams;
        if checkpoints.interval == 0 {
            return Err(Error::InvalidArgument);
        }
        if n_per_stream % checkpoints.interval != 0 {
            return Err(Error::InvalidArgument); // whole chunks only: the C call indexes [n_streams][n_per_stream / interval]
        }
        let n_chunks = n_per_stream / checkpoints.interval;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        let n_points = n_streams.checked_mul(n_chunks).ok_or(Error::InvalidArgument)?;
        if checkpoints.pos.len() < n_points || checkpoints.lower.len() < n_points || checkpoints.range.len() < n_points {
            return Err(Error::InvalidArgument);
        }
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(count)?, status: DeviceBuffer::new(n_points)? };
        let mut scratch: DeviceBuffer<u8> = DeviceBuffer::new(unsafe { ffi::cst_range_ckpt_scratch_bytes(n_streams, n_per_stream, checkpoints.interval) })?;
        check(unsafe {
            ffi::cst_range_decode_batch_ckpt(
                model.as_raw(),
                self.config,
                encoded.words.as_ptr(),
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                encoded.n_words.as_ptr(),
                checkpoints.interval,
                checkpoints.pos.as_ptr(),
                checkpoints.lower.as_ptr(),
                checkpoints.range.as_ptr(),
                out.symbols.as_mut_ptr(),
                n_streams,
                n_per_stream,
                scratch.as_mut_ptr() as *mut c_void,
                out.status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?; // (the scratch buffer is dropped on return)
        Ok(out)
    }

    /// `RangeDecoder::seek` + `decode_symbols` of one chunk's models, for every chunk at once (`cst_range_decode_gaussian_batch_ckpt`): two
    /// resident waves per SIMD where the plain per-symbol decoder of a 65 536-stream batch has one.  One status per (stream, chunk).
    pub fn decode_symbols_from_checkpoints(
        &self,
        encoded: &EncodedBatch,
        checkpoints: &RangeCheckpoints,
        support: RangeInclusive<i32>,
        means: &DeviceBuffer<f64>,
        stds: &DeviceBuffer<f64>,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<DecodedBatch> {
        let n_streams = encoded.n_streams;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if checkpoints.interval == 0 || n_per_stream % checkpoints.interval != 0 || means.len() < count || stds.len() < count {
            return Err(Error::InvalidArgument);
        }
        let n_points = n_streams.checked_mul(n_per_stream / checkpoints.interval).ok_or(Error::InvalidArgument)?;
        if checkpoints.pos.len() < n_points || checkpoints.lower.len() < n_points || checkpoints.range.len() < n_points {
            return Err(Error::InvalidArgument);
        }
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(count)?, status: DeviceBuffer::new(n_points)? };
        let mut scratch: DeviceBuffer<u8> =
            DeviceBuffer::new(unsafe { ffi::cst_range_gaussian_ckpt_scratch_bytes(n_streams, n_per_stream, checkpoints.interval) })?;
        check(unsafe {
            ffi::cst_range_decode_gaussian_batch_ckpt(
                self.config,
                *support.start(),
                *support.end(),
                encoded.words.as_ptr(),
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                encoded.n_words.as_ptr(),
                checkpoints.interval,
                checkpoints.pos.as_ptr(),
                checkpoints.lower.as_ptr(),
                checkpoints.range.as_ptr(),
                means.as_ptr(),
                stds.as_ptr(),
                out.symbols.as_mut_ptr(),
                n_streams,
                n_per_stream,
                scratch.as_mut_ptr() as *mut c_void,
                out.status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?; // (the scratch buffer is dropped on return)
        Ok(out)
    }

    /// Per stream: `decode_symbols(models)` with one quantized Gaussian per symbol
    /// (src/pybindings/stream/queue.rs:598-661).
    pub fn decode_symbols(
        &self,
        encoded: &EncodedBatch,
        support: RangeInclusive<i32>,
        means: &DeviceBuffer<f64>,
        stds: &DeviceBuffer<f64>,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<DecodedBatch> {
        let n_streams = encoded.n_streams;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if means.len() < count || stds.len() < count {
            return Err(Error::InvalidArgument);
        }
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(count)?, status: DeviceBuffer::new(n_streams)? };
        check(unsafe {
            ffi::cst_range_decode_gaussian_batch(
                self.config,
                *support.start(),
                *support.end(),
                encoded.words.as_ptr(),
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                encoded.n_words.as_ptr(),
                means.as_ptr(),
                stds.as_ptr(),
                out.symbols.as_mut_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    /// `decode_symbols` with one explicit cdf row per coded symbol.
    pub fn decode_symbols_with_cdf_rows(
        &self,
        encoded: &EncodedBatch,
        cdf_rows: &DeviceBuffer<u32>,
        n_symbols: usize,
        min_symbol: i32,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<DecodedBatch> {
        let n_streams = encoded.n_streams;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if cdf_rows.len() < count * (n_symbols + 1) {
            return Err(Error::InvalidArgument);
        }
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(count)?, status: DeviceBuffer::new(n_streams)? };
        check(unsafe {
            ffi::cst_range_decode_rows_batch(
                self.config,
                encoded.words.as_ptr(),
                core::ptr::null(),
                encoded.stride_words,
                encoded.words.len(),
                encoded.n_words.as_ptr(),
                cdf_rows.as_ptr(),
                n_symbols as i32,
                min_symbol,
                out.symbols.as_mut_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                core::ptr::null_mut(),
                out.status.as_mut_ptr(),
                ffi::CST_FLAG_NONE,
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// chain coder: the symbol loops of `ChainCoder` (src/stream/chain.rs:1044-1122, 1140-1209) for many chains
// ---------------------------------------------------------------------------------------------------------------------

/// The two stacks and heads of many `ChainCoder`s.  `pop_*` is the stack the call consumes (from the end), `push_*` the
/// words it produces; constructors and terminators (`from_binary`, `into_remainders`, ...) stay host code as in the reference.
pub struct ChainBatch {
    pub pop_words: DeviceBuffer<u32>,
    pub pop_stride: usize,
    pub n_pop: DeviceBuffer<u32>,
    pub push_words: DeviceBuffer<u32>,
    pub push_stride: usize,
    pub n_push: DeviceBuffer<u32>,
    pub heads: DeviceBuffer<ChainHeads>,
    pub n_streams: usize,
}

/// `ChainCoder::decode_symbols` / `encode_symbols_reverse` with one quantized Gaussian per symbol.
#[derive(Clone, Copy, Debug)]
pub struct BatchedChainCoder {
    pub config: CoderConfig,
    pub layout: Layout,
}

impl BatchedChainCoder {
    pub fn new(config: CoderConfig) -> Self {
        BatchedChainCoder { config, layout: Layout::StreamMajor }
    }

    /// `decode_symbols` (chain.rs:1044-1122): pops `precision` bits per symbol from `compressed`, pushes onto `remainders`.
    pub fn decode_symbols(
        &self,
        chains: &mut ChainBatch,
        support: RangeInclusive<i32>,
        means: &DeviceBuffer<f64>,
        stds: &DeviceBuffer<f64>,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<DecodedBatch> {
        let n_streams = chains.n_streams;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if means.len() < count || stds.len() < count {
            return Err(Error::InvalidArgument);
        }
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(count)?, status: DeviceBuffer::new(n_streams)? };
        check(unsafe {
            ffi::cst_chain_decode_gaussian_batch(
                self.config,
                *support.start(),
                *support.end(),
                chains.pop_words.as_ptr(),
                core::ptr::null(),
                chains.pop_stride,
                chains.n_pop.as_mut_ptr(),
                means.as_ptr(),
                stds.as_ptr(),
                out.symbols.as_mut_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                chains.push_words.as_mut_ptr(),
                chains.push_stride,
                chains.n_push.as_mut_ptr(),
                chains.heads.as_mut_ptr(),
                out.status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    /// `decode_iid_symbols` / `decode_symbols` with explicit cdf rows; `row_stride = 0` shares ONE row (a concrete model).
    pub fn decode_symbols_with_cdf_rows(
        &self,
        chains: &mut ChainBatch,
        cdf_rows: &DeviceBuffer<u32>,
        row_stride: usize,
        n_symbols: usize,
        min_symbol: i32,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<DecodedBatch> {
        let n_streams = chains.n_streams;
        let mut out = DecodedBatch { symbols: DeviceBuffer::new(n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?)?, status: DeviceBuffer::new(n_streams)? };
        check(unsafe {
            ffi::cst_chain_decode_rows_batch(
                self.config,
                chains.pop_words.as_ptr(),
                core::ptr::null(),
                chains.pop_stride,
                chains.n_pop.as_mut_ptr(),
                cdf_rows.as_ptr(),
                row_stride,
                n_symbols as i32,
                min_symbol,
                out.symbols.as_mut_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                chains.push_words.as_mut_ptr(),
                chains.push_stride,
                chains.n_push.as_mut_ptr(),
                chains.heads.as_mut_ptr(),
                out.status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok(out)
    }

    /// `encode_symbols_reverse` (chain.rs:1140-1209): pops from `remainders`, pushes `precision` bits per symbol onto `compressed`.
    pub fn encode_symbols_reverse(
        &self,
        chains: &mut ChainBatch,
        symbols: &DeviceBuffer<i32>,
        support: RangeInclusive<i32>,
        means: &DeviceBuffer<f64>,
        stds: &DeviceBuffer<f64>,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<DeviceBuffer<i32>> {
        let n_streams = chains.n_streams;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if symbols.len() < count || means.len() < count || stds.len() < count {
            return Err(Error::InvalidArgument);
        }
        let mut status: DeviceBuffer<i32> = DeviceBuffer::new(n_streams)?;
        check(unsafe {
            ffi::cst_chain_encode_gaussian_batch(
                self.config,
                *support.start(),
                *support.end(),
                symbols.as_ptr(),
                means.as_ptr(),
                stds.as_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                chains.pop_words.as_ptr(),
                core::ptr::null(),
                chains.pop_stride,
                chains.n_pop.as_mut_ptr(),
                chains.push_words.as_mut_ptr(),
                chains.push_stride,
                chains.n_push.as_mut_ptr(),
                chains.heads.as_mut_ptr(),
                status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok(status)
    }

    /// `encode_symbols_reverse` with explicit `(left_cumulative, probability)` per symbol.
    pub fn encode_symbols_reverse_with_cp(
        &self,
        chains: &mut ChainBatch,
        left: &DeviceBuffer<u32>,
        prob: &DeviceBuffer<u32>,
        n_per_stream: usize,
        stream: &Stream,
    ) -> Result<DeviceBuffer<i32>> {
        let n_streams = chains.n_streams;
        let count = n_streams.checked_mul(n_per_stream).ok_or(Error::InvalidArgument)?;
        if left.len() < count || prob.len() < count {
            return Err(Error::InvalidArgument);
        }
        let mut status: DeviceBuffer<i32> = DeviceBuffer::new(n_streams)?;
        check(unsafe {
            ffi::cst_chain_encode_cp_batch(
                self.config,
                left.as_ptr(),
                prob.as_ptr(),
                n_streams,
                n_per_stream,
                self.layout.raw(),
                chains.pop_words.as_ptr(),
                core::ptr::null(),
                chains.pop_stride,
                chains.n_pop.as_mut_ptr(),
                chains.push_words.as_mut_ptr(),
                chains.push_stride,
                chains.n_push.as_mut_ptr(),
                chains.heads.as_mut_ptr(),
                status.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok(status)
    }
}

/// Hands the library's scratch pool (per-symbol entry points) back to the device.
pub fn release_scratch() -> Result<()> {
    check(unsafe { ffi::cst_release_scratch() })
}

// ---------------------------------------------------------------------------------------------------------------------
// multi-GPU: one process per GPU, streams sharded in contiguous blocks; the only exchange is the gather of the packed
// words to one rank (and the inverse scatter) over RCCL / xGMI.  The reference has no counterpart (single-threaded coders).
// ---------------------------------------------------------------------------------------------------------------------

/// An RCCL communicator owned by the library (librccl is opened at first use).
pub struct Communicator {
    raw: *mut c_void,
    pub n_ranks: usize,
    pub rank: usize,
}

impl Communicator {
    /// Rank 0 creates the 128-byte id; the caller's launcher hands it to the other ranks.
    pub fn unique_id() -> Result<[u8; 128]> {
        let mut id = [0u8; 128];
        check(unsafe { ffi::cst_rccl_get_unique_id(id.as_mut_ptr() as *mut c_void) })?;
        Ok(id)
    }

    pub fn new(id: &[u8; 128], n_ranks: usize, rank: usize) -> Result<Self> {
        let mut raw = core::ptr::null_mut();
        check(unsafe { ffi::cst_rccl_comm_init(id.as_ptr() as *const c_void, n_ranks as i32, rank as i32, &mut raw) })?;
        Ok(Communicator { raw, n_ranks, rank })
    }

    /// `(n_streams, total_words)` of every rank, in rank order (one small all-gather; synchronises `stream`).
    pub fn gather_sizes(&self, packed: &PackedBatch, stream: &Stream) -> Result<Vec<u64>> {
        let mut sizes: DeviceBuffer<u64> = DeviceBuffer::zeroed(2 * self.n_ranks, stream)?;
        check(unsafe {
            ffi::cst_gather_sizes_rccl(
                self.raw,
                self.n_ranks as i32,
                self.rank as i32,
                packed.offsets.as_ptr(),
                packed.n_streams,
                sizes.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        stream.synchronize()?;
        Ok(sizes.to_vec()?)
    }

    /// Every rank's packed words and offsets to `root`, straight into their final positions (grouped point-to-point
    /// transfers, each peer on its own xGMI link).  Returns `(all_words, all_offsets)` on the root, `None` elsewhere.
    pub fn gather(
        &self,
        packed: &PackedBatch,
        sizes: &[u64],
        root: usize,
        stream: &Stream,
    ) -> Result<Option<(DeviceBuffer<u32>, DeviceBuffer<u64>)>> {
        if sizes.len() != 2 * self.n_ranks || root >= self.n_ranks {
            return Err(Error::InvalidArgument);
        }
        let total_streams: u64 = sizes.iter().step_by(2).sum();
        let total_words: u64 = sizes.iter().skip(1).step_by(2).sum();
        if self.rank != root {
            check(unsafe {
                ffi::cst_gather_rccl(
                    self.raw,
                    self.n_ranks as i32,
                    self.rank as i32,
                    root as i32,
                    packed.words.as_ptr(),
                    packed.offsets.as_ptr(),
                    sizes.as_ptr(),
                    core::ptr::null_mut(),
                    core::ptr::null_mut(),
                    stream.as_raw(),
                )
            })?;
            return Ok(None);
        }
        let mut all_words: DeviceBuffer<u32> = DeviceBuffer::new(total_words as usize)?;
        let mut all_offsets: DeviceBuffer<u64> = DeviceBuffer::new(total_streams as usize + 1)?;
        check(unsafe {
            ffi::cst_gather_rccl(
                self.raw,
                self.n_ranks as i32,
                self.rank as i32,
                root as i32,
                packed.words.as_ptr(),
                packed.offsets.as_ptr(),
                sizes.as_ptr(),
                all_words.as_mut_ptr(),
                all_offsets.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok(Some((all_words, all_offsets)))
    }

    /// The inverse: `root` hands every rank the words of its own streams and their offsets rebased to 0 -- what
    /// [`BatchedAnsCoder::decode_iid_symbols_packed`] takes.  `all` is `Some` on the root only; `n_words` are the
    /// rank's own per-stream word counts (differences of the offsets work as well).
    pub fn scatter(
        &self,
        all: Option<(&DeviceBuffer<u32>, &DeviceBuffer<u64>)>,
        sizes: &[u64],
        root: usize,
        n_words: DeviceBuffer<u32>,
        config: CoderConfig,
        stream: &Stream,
    ) -> Result<PackedBatch> {
        if sizes.len() != 2 * self.n_ranks || root >= self.n_ranks || (self.rank == root) != all.is_some() {
            return Err(Error::InvalidArgument);
        }
        let my_streams = sizes[2 * self.rank] as usize;
        let my_words = sizes[2 * self.rank + 1] as usize;
        let mut words: DeviceBuffer<u32> = DeviceBuffer::new(my_words)?;
        let mut offsets: DeviceBuffer<u64> = DeviceBuffer::new(my_streams + 1)?;
        let (all_words, all_offsets) = match all {
            Some((w, o)) => (w.as_ptr(), o.as_ptr()),
            None => (core::ptr::null(), core::ptr::null()),
        };
        check(unsafe {
            ffi::cst_scatter_rccl(
                self.raw,
                self.n_ranks as i32,
                self.rank as i32,
                root as i32,
                all_words,
                all_offsets,
                sizes.as_ptr(),
                words.as_mut_ptr(),
                offsets.as_mut_ptr(),
                stream.as_raw(),
            )
        })?;
        Ok(PackedBatch { words, offsets, n_words, n_streams: my_streams, config })
    }
}

impl Drop for Communicator {
    fn drop(&mut self) {
        unsafe { ffi::cst_rccl_comm_destroy(self.raw) };
    }
}
