// To be appended to `neuronika-variable/src/gradient.rs` under `#[cfg(feature = "hip")]` (the fields of `Gradient` are
// private to that module): the device twins of `ndarray_zeros` / `no_grad` / `with_grad` (`gradient.rs:47-79`).
#[cfg(feature = "hip")]
impl<D> Gradient<crate::hip::HipArray<D>, D>
where
    D: Dimension,
{
    /// Zeroed gradient buffer in HBM (`ndarray_zeros`, `gradient.rs:47-54`).
    pub(crate) fn hip_zeros(dim: D, device: crate::hip::Device) -> Self {
        Self {
            shape: dim.clone(),
            array: RefCell::new(Some(crate::hip::HipArray::zeroed(dim, device))),
        }
    }
}

#[cfg(feature = "hip")]
impl<D> NoGrad for Gradient<crate::hip::HipArray<D>, D>
where
    D: Dimension,
{
    /// De-allocates the device buffer (`gradient.rs:64-66`).
    fn no_grad(&self) {
        *self.array.borrow_mut() = None;
    }

    /// Re-allocates it zeroed (`gradient.rs:68-78`); the device is the one the gradient was created on.
    fn with_grad(&self) {
        let mut option = self.array.borrow_mut();
        if option.is_none() {
            let device = crate::hip::Device::current();
            *option = Some(crate::hip::HipArray::zeroed(self.shape.clone(), device))
        }
    }
}
