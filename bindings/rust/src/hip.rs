//! The handful of HIP runtime calls the safe wrappers need: device buffers, streams, copies.  (The coder library itself
//! takes plain device pointers and a `hipStream_t` as `void *`; any other HIP binding's pointers work as well.)
use core::ffi::c_void;
use core::marker::PhantomData;

#[link(name = "amdhip64")]
extern "C" {
    fn hipMalloc(ptr: *mut *mut c_void, size: usize) -> i32;
    fn hipFree(ptr: *mut c_void) -> i32;
    fn hipMemcpy(dst: *mut c_void, src: *const c_void, size: usize, kind: i32) -> i32;
    fn hipMemsetAsync(dst: *mut c_void, value: i32, size: usize, stream: *mut c_void) -> i32;
    fn hipStreamCreate(stream: *mut *mut c_void) -> i32;
    fn hipStreamDestroy(stream: *mut c_void) -> i32;
    fn hipStreamSynchronize(stream: *mut c_void) -> i32;
}

const HIP_MEMCPY_HOST_TO_DEVICE: i32 = 1;
const HIP_MEMCPY_DEVICE_TO_HOST: i32 = 2;

/// A HIP runtime error code (`hipError_t`).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct HipError(pub i32);

fn hip(code: i32) -> Result<(), HipError> {
    if code == 0 {
        Ok(())
    } else {
        Err(HipError(code))
    }
}

/// A HIP stream; every batched call of the library is asynchronous on the stream it is given.
pub struct Stream(*mut c_void);

impl Stream {
    pub fn new() -> Result<Self, HipError> {
        let mut raw = core::ptr::null_mut();
        hip(unsafe { hipStreamCreate(&mut raw) })?;
        Ok(Stream(raw))
    }

    /// The default (NULL) stream.
    pub fn default_stream() -> Self {
        Stream(core::ptr::null_mut())
    }

    pub fn synchronize(&self) -> Result<(), HipError> {
        hip(unsafe { hipStreamSynchronize(self.0) })
    }

    pub fn as_raw(&self) -> *mut c_void {
        self.0
    }
}

impl Drop for Stream {
    fn drop(&mut self) {
        if !self.0.is_null() {
            unsafe { hipStreamDestroy(self.0) };
        }
    }
}

/// `len` elements of `T` in HBM (hipMalloc); freed on drop.
pub struct DeviceBuffer<T: Copy> {
    ptr: *mut T,
    len: usize,
    _marker: PhantomData<T>,
}

impl<T: Copy> DeviceBuffer<T> {
    /// Uninitialised device memory (at least one element is allocated so that the pointer is never NULL).
    pub fn new(len: usize) -> Result<Self, HipError> {
        let mut raw = core::ptr::null_mut();
        let bytes = core::cmp::max(len, 1) * core::mem::size_of::<T>();
        hip(unsafe { hipMalloc(&mut raw, bytes) })?;
        Ok(DeviceBuffer { ptr: raw as *mut T, len, _marker: PhantomData })
    }

    pub fn zeroed(len: usize, stream: &Stream) -> Result<Self, HipError> {
        let buf = Self::new(len)?;
        hip(unsafe { hipMemsetAsync(buf.ptr as *mut c_void, 0, len * core::mem::size_of::<T>(), stream.as_raw()) })?;
        Ok(buf)
    }

    /// Copies a host slice to the device (synchronous, like the reference's copy of the numpy input,
    /// src/pybindings/mod.rs:240-243).
    pub fn from_slice(host: &[T]) -> Result<Self, HipError> {
        let buf = Self::new(host.len())?;
        if !host.is_empty() {
            let bytes = host.len() * core::mem::size_of::<T>();
            hip(unsafe { hipMemcpy(buf.ptr as *mut c_void, host.as_ptr() as *const c_void, bytes, HIP_MEMCPY_HOST_TO_DEVICE) })?;
        }
        Ok(buf)
    }

    /// Copies the first `count` elements back to the host (synchronises the device).
    pub fn to_vec_prefix(&self, count: usize) -> Result<Vec<T>, HipError> {
        assert!(count <= self.len);
        let mut host: Vec<T> = Vec::with_capacity(count);
        if count > 0 {
            let bytes = count * core::mem::size_of::<T>();
            hip(unsafe { hipMemcpy(host.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, bytes, HIP_MEMCPY_DEVICE_TO_HOST) })?;
        }
        unsafe { host.set_len(count) };
        Ok(host)
    }

    pub fn to_vec(&self) -> Result<Vec<T>, HipError> {
        self.to_vec_prefix(self.len)
    }

    pub fn len(&self) -> usize {
        self.len
    }

    pub fn is_empty(&self) -> bool {
        self.len == 0
    }

    pub fn as_ptr(&self) -> *const T {
        self.ptr
    }

    pub fn as_mut_ptr(&mut self) -> *mut T {
        self.ptr
    }
}

impl<T: Copy> Drop for DeviceBuffer<T> {
    fn drop(&mut self) {
        unsafe { hipFree(self.ptr as *mut c_void) };
    }
}
