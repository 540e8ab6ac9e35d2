use std::rc::Rc;

use super::ffi;

/// Owner of the raw handle: destroyed when the last `Device` clone goes away.
#[derive(Debug)]
struct Raw(*mut ffi::nk_device);

impl Drop for Raw {
    fn drop(&mut self) {
        unsafe { ffi::nk_device_destroy(self.0) };
    }
}

thread_local! {
    static CURRENT: std::cell::RefCell<Option<Device>> = std::cell::RefCell::new(None);
}

/// Handle to one MI355X: a compute stream (tape-ordered kernels), a high-priority communication stream (RCCL) and a
/// copy stream, created by `nk_device_create`.  Cloning shares the handle, like the reference's `Rc`-held contexts
/// (`cuda/device.rs:11-16`).  `!Send`: one host thread drives one device, as the `Rc<RefCell>` tape requires.
#[derive(Clone, Debug)]
pub struct Device {
    raw: Rc<Raw>,
}

impl Device {
    /// Creates a handle to GPU `device`.
    ///
    /// # Panics
    ///
    /// If the requested device is not found (reference behaviour: `cuda/device.rs:36-45` unwraps).
    pub fn new(device: u32) -> Self {
        let mut raw = std::ptr::null_mut();
        ffi::check(unsafe { ffi::nk_device_create(device as i32, &mut raw) });
        let this = Self { raw: Rc::new(Raw(raw)) };
        CURRENT.with(|c| *c.borrow_mut() = Some(this.clone()));
        this
    }

    /// The device handle most recently created on this thread (a process drives one GPU in the one-process-per-GPU model;
    /// `Gradient::with_grad` re-allocates on it).
    pub fn current() -> Self {
        CURRENT.with(|c| c.borrow().clone().expect("no hip::Device created on this thread"))
    }

    /// Number of visible GPUs.
    pub fn count() -> usize {
        let mut n = 0;
        ffi::check(unsafe { ffi::nk_device_count(&mut n) });
        n as usize
    }

    /// Waits for everything enqueued on the device's streams (`item()`-style host reads do this implicitly).
    pub fn sync(&self) {
        ffi::check(unsafe { ffi::nk_device_sync(self.as_raw()) });
    }

    pub(crate) fn as_raw(&self) -> *mut ffi::nk_device {
        self.raw.0
    }
}
