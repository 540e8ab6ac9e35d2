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

/// `Optimizer::step` (`neuronika-optim/src/optimizer.rs:81-86`) for SGD over ALL registered parameters as ONE launch
/// (`nk_sgd_step_multi`): the per-parameter updates are independent, element for element the arithmetic of `sgd_step`.
/// `params`: (weights, gradient, velocity) raw buffers and element counts, collected by the optimizer from its
/// `SGDParam`s (`sgd/mod.rs:186-236`); a parameter registered twice is updated once (the launch's updates run concurrently).
#[allow(clippy::too_many_arguments)]
pub(crate) fn sgd_step_multi(device: &crate::hip::device::Device, params: &[(*mut f32, *mut f32, *mut f32, usize)], lr: f32, momentum: f32,
                             dampening: f32, nesterov: bool, l1: f32, l2: f32) {
    let mut seen = std::collections::HashSet::new();
    let unique: Vec<&(*mut f32, *mut f32, *mut f32, usize)> = params.iter().filter(|p| seen.insert(p.0 as usize)).collect();
    let w: Vec<*mut f32> = unique.iter().map(|p| p.0).collect();
    let g: Vec<*mut f32> = unique.iter().map(|p| p.1).collect();
    let v: Vec<*mut f32> = unique.iter().map(|p| p.2).collect();
    let n: Vec<usize> = unique.iter().map(|p| p.3).collect();
    let velocity = if momentum != 0. { v.as_ptr() } else { std::ptr::null() };
    ffi::check(unsafe {
        ffi::nk_sgd_step_multi(device.as_raw(), w.len() as i32, w.as_ptr(), g.as_ptr(), velocity, n.as_ptr(), lr, momentum, dampening,
                               nesterov as i32, l1, l2)
    });
}
