//! Device-resident optimizer steps: what `SGDParam::optimize` (`neuronika-optim/src/sgd/mod.rs:186-236`)
//! and `AdamParam::optimize` (`adam/mod.rs:126-174`) do per parameter, on the parameter's own HBM buffers - called from the
//! optimizer's `step()` (`optimizer.rs:81-94`) on the far side of the gradient exchange.
use ndarray::Dimension;

use crate::hip::{ffi, hiparray::HipArray};

/// `w -= lr * (grad + penalty)` with optional momentum / dampening / Nesterov (`velocity`: `None` = plain SGD).
#[allow(clippy::too_many_arguments)]
pub(crate) fn sgd_step<D: Dimension>(w: &mut HipArray<D>, grad: &mut HipArray<D>, velocity: Option<&mut HipArray<D>>, lr: f32, momentum: f32,
                                     dampening: f32, nesterov: bool, l1: f32, l2: f32) {
    let v = velocity.map_or(std::ptr::null_mut(), |v| v.as_mut_ptr());
    ffi::check(unsafe {
        ffi::nk_sgd_step(w.device().as_raw(), w.as_mut_ptr(), grad.as_mut_ptr(), v, w.len(), lr, momentum, dampening, nesterov as i32, l1, l2)
    });
}

/// Adam / AMSGrad (`max_exp_avg_sq`: `Some` selects AMSGrad); `step` is the 1-based step count of the bias corrections.
#[allow(clippy::too_many_arguments)]
pub(crate) fn adam_step<D: Dimension>(w: &mut HipArray<D>, grad: &mut HipArray<D>, exp_avg: &mut HipArray<D>, exp_avg_sq: &mut HipArray<D>,
                                      max_exp_avg_sq: Option<&mut HipArray<D>>, lr: f32, beta1: f32, beta2: f32, eps: f32, step: i32,
                                      l1: f32, l2: f32) {
    let m = max_exp_avg_sq.map_or(std::ptr::null_mut(), |m| m.as_mut_ptr());
    ffi::check(unsafe {
        ffi::nk_adam_step(w.device().as_raw(), w.as_mut_ptr(), grad.as_mut_ptr(), exp_avg.as_mut_ptr(), exp_avg_sq.as_mut_ptr(), m, w.len(),
                          lr, beta1, beta2, eps, step, l1, l2)
    });
}
