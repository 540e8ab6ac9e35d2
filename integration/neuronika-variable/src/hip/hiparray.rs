use ndarray::{Array, Dimension};

use super::{device::Device, ffi};

/// A dense, C-contiguous `f32` array in HBM: the device twin of `ndarray::Array<f32, D>` (`CuArray<f32, D>` in the
/// reference's template, `cuda/cuarray.rs:10-19`).  Owns its buffer and frees it on drop; the C side never retains the
/// pointer past a call.
pub struct HipArray<D>
where
    D: Dimension,
{
    ptr: *mut f32,
    dim: D,
    device: Device,
}

impl<D> HipArray<D>
where
    D: Dimension,
{
    /// New array of zeros (`CuArray::zeroed`, `cuda/cuarray.rs:35-42`): node outputs and gradients are allocated
    /// zeroed at graph-build time (`var.rs:224,1041`, `gradient.rs:47-54`).
    pub(crate) fn zeroed(dim: D, device: Device) -> Self {
        let mut ptr = std::ptr::null_mut();
        ffi::check(unsafe { ffi::nk_alloc_zeroed(device.as_raw(), dim.size(), &mut ptr) });
        Self { ptr, dim, device }
    }

    /// Upload of a host slice (`CuArray::from_slice`, `cuda/cuarray.rs:62-72`).
    pub(crate) fn from_slice(slice: &[f32], dim: D, device: Device) -> Self {
        assert_eq!(slice.len(), dim.size());
        let this = Self::zeroed(dim, device);
        ffi::check(unsafe { ffi::nk_upload(this.device.as_raw(), this.ptr, slice.as_ptr(), slice.len()) });
        this
    }

    pub(crate) fn dimension(&self) -> D {
        self.dim.clone()
    }

    /// Shape as the C ABI takes it (`const int*`, rank).
    pub(crate) fn shape_c(&self) -> Vec<i32> {
        self.dim.slice().iter().map(|&s| s as i32).collect()
    }

    pub(crate) fn len(&self) -> usize {
        self.dim.size()
    }

    pub(crate) fn device(&self) -> &Device {
        &self.device
    }

    pub(crate) fn as_ptr(&self) -> *const f32 {
        self.ptr
    }

    pub(crate) fn as_mut_ptr(&mut self) -> *mut f32 {
        self.ptr
    }

    /// `fill` (root-gradient seeding `vardiff.rs:133`, `zero_grad` `vardiff.rs:100-102`).
    pub(crate) fn fill(&mut self, value: f32) {
        ffi::check(unsafe { ffi::nk_fill(self.device.as_raw(), self.ptr, self.len(), value) });
    }

    /// Download into a host array (`CuArray::as_ndarray`, `cuda/cuarray.rs:101-105`); synchronises.
    pub fn as_ndarray(&self) -> Array<f32, D> {
        let mut host = vec![0f32; self.len()];
        ffi::check(unsafe { ffi::nk_download(self.device.as_raw(), host.as_mut_ptr(), self.ptr, host.len()) });
        Array::from_shape_vec(self.dim.clone(), host).unwrap()
    }

    /// Upload of a host array (`CuArray::from_ndarray`, `cuda/cuarray.rs:113-117`).
    pub fn from_ndarray(array: &Array<f32, D>, device: Device) -> Self {
        Self::from_slice(array.as_standard_layout().as_slice().unwrap(), array.raw_dim(), device)
    }
}

impl<D> Drop for HipArray<D>
where
    D: Dimension,
{
    fn drop(&mut self) {
        unsafe { ffi::nk_free(self.device.as_raw(), self.ptr) };
    }
}
