use super::grad_id;
use std::{cell::Cell, rc::Rc};

use ndarray::Dimension;

use crate::{
    autograd::{Backward, Forward},
    gradient::Gradient,
    hip::{ffi, hiparray::HipArray},
    utils::Shared,
};

/// `ReLU::forward` (`node/relu/mod.rs:29-38`).
pub(crate) struct ReLU<D: Dimension> {
    operand_data: Shared<HipArray<D>>,
    data: Shared<HipArray<D>>,
}

impl<D: Dimension> ReLU<D> {
    pub(crate) fn new(operand_data: Shared<HipArray<D>>, data: Shared<HipArray<D>>) -> Self {
        Self { operand_data, data }
    }
}

impl<D: Dimension> Forward for ReLU<D> {
    fn forward(&self) {
        let x = self.operand_data.borrow();
        let mut y = self.data.borrow_mut();
        ffi::check(unsafe { ffi::nk_relu_fwd(x.device().as_raw(), x.as_ptr(), y.as_mut_ptr(), x.len()) });
    }
}

/// `ReLUBackward::backward` (`:67-79`): `dx += (x > 0) * g` on the node's INPUT, strict inequality.
pub(crate) struct ReLUBackward<D: Dimension> {
    operand_data: Shared<HipArray<D>>,
    operand_gradient: Rc<Gradient<HipArray<D>, D>>,
    gradient: Rc<Gradient<HipArray<D>, D>>,
}

impl<D: Dimension> ReLUBackward<D> {
    pub(crate) fn new(operand_gradient: Rc<Gradient<HipArray<D>, D>>, operand_data: Shared<HipArray<D>>, gradient: Rc<Gradient<HipArray<D>, D>>) -> Self {
        Self { operand_gradient, operand_data, gradient }
    }
}

impl<D: Dimension> Backward for ReLUBackward<D> {
    fn backward(&self) {
        let (g, x) = (self.gradient.borrow(), self.operand_data.borrow());
        let mut dx = self.operand_gradient.borrow_mut();
        ffi::check(unsafe { ffi::nk_relu_bwd(g.device().as_raw(), dx.as_mut_ptr(), g.as_ptr(), x.as_ptr(), x.len()) });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.operand_gradient)]
    }
}

/// `Softmax::forward` (`node/softmax/mod.rs:37-53`) along `axis`.
pub(crate) struct Softmax<D: Dimension> {
    operand_data: Shared<HipArray<D>>,
    data: Shared<HipArray<D>>,
    axis: usize,
}

impl<D: Dimension> Softmax<D> {
    pub(crate) fn new(operand_data: Shared<HipArray<D>>, data: Shared<HipArray<D>>, axis: usize) -> Self {
        Self { operand_data, data, axis }
    }
}

impl<D: Dimension> Forward for Softmax<D> {
    fn forward(&self) {
        let x = self.operand_data.borrow();
        let mut y = self.data.borrow_mut();
        let s = x.shape_c();
        ffi::check(unsafe { ffi::nk_softmax_fwd(x.device().as_raw(), x.as_ptr(), y.as_mut_ptr(), s.as_ptr(), s.len() as i32, self.axis as i32) });
    }
}

/// `SoftmaxBackward::backward` (`:84-104`): `dx += y * (g - sum(g * y))` per lane.
pub(crate) struct SoftmaxBackward<D: Dimension> {
    data: Shared<HipArray<D>>,
    operand_gradient: Rc<Gradient<HipArray<D>, D>>,
    gradient: Rc<Gradient<HipArray<D>, D>>,
    axis: usize,
}

impl<D: Dimension> SoftmaxBackward<D> {
    pub(crate) fn new(operand_gradient: Rc<Gradient<HipArray<D>, D>>, data: Shared<HipArray<D>>, gradient: Rc<Gradient<HipArray<D>, D>>, axis: usize) -> Self {
        Self { operand_gradient, data, gradient, axis }
    }
}

impl<D: Dimension> Backward for SoftmaxBackward<D> {
    fn backward(&self) {
        let (g, y) = (self.gradient.borrow(), self.data.borrow());
        let mut dx = self.operand_gradient.borrow_mut();
        let s = y.shape_c();
        ffi::check(unsafe {
            ffi::nk_softmax_bwd(g.device().as_raw(), dx.as_mut_ptr(), g.as_ptr(), y.as_ptr(), s.as_ptr(), s.len() as i32, self.axis as i32)
        });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.operand_gradient)]
    }
}

/// `Dropout::forward` (`node/dropout/mod.rs:53-79`).  The reference draws Bernoulli(1 - p) noise from `thread_rng` on
/// every forward; here the mask is Philox4x32-10 keyed by `seed`, with the counter offset advanced per call, and is
/// stored (0/1, f32) for the backward node exactly as the reference's noise buffer is.
pub(crate) struct Dropout<D: Dimension> {
    operand_data: Shared<HipArray<D>>,
    data: Shared<HipArray<D>>,
    noise: Shared<HipArray<D>>,
    p: f64,
    status: Rc<Cell<bool>>,
    seed: u64,
    calls: Cell<u64>,
}

impl<D: Dimension> Dropout<D> {
    /// Panics on `p` outside `[0, 1]` like the reference (`node/dropout/mod.rs:38-40`).
    pub(crate) fn new(operand_data: Shared<HipArray<D>>, data: Shared<HipArray<D>>, p: f64, noise: Shared<HipArray<D>>, status: Rc<Cell<bool>>,
                      seed: u64) -> Self {
        if !(0. ..=1.).contains(&p) {
            panic!("Wrong probability received: {}.", p);
        }
        Self { operand_data, data, noise, p, status, seed, calls: Cell::new(0) }
    }
}

impl<D: Dimension> Forward for Dropout<D> {
    fn forward(&self) {
        let x = self.operand_data.borrow();
        let (mut y, mut noise) = (self.data.borrow_mut(), self.noise.borrow_mut());
        let offset = self.calls.get() * ((x.len() as u64 + 7) / 8); // one Philox call serves 8 draws (include/neuronika_hip.h, dropout)
        ffi::check(unsafe {
            ffi::nk_dropout_fwd(x.device().as_raw(), x.as_ptr(), y.as_mut_ptr(), noise.as_mut_ptr(), x.len(), self.p,
                                self.status.get() as i32, self.seed, offset)
        });
        self.calls.set(self.calls.get() + 1); // only a forward that was issued consumes its Philox range (`check` panics on a refusal)
    }
}

/// `DropoutBackward::backward` (`:113-128`): `dx += g * noise` - NOT divided by `1 - p` (reference behaviour, kept).
pub(crate) struct DropoutBackward<D: Dimension> {
    operand_gradient: Rc<Gradient<HipArray<D>, D>>,
    gradient: Rc<Gradient<HipArray<D>, D>>,
    noise: Shared<HipArray<D>>,
    p: f64,
    status: Rc<Cell<bool>>,
}

impl<D: Dimension> DropoutBackward<D> {
    pub(crate) fn new(operand_gradient: Rc<Gradient<HipArray<D>, D>>, gradient: Rc<Gradient<HipArray<D>, D>>, p: f64, noise: Shared<HipArray<D>>, status: Rc<Cell<bool>>) -> Self {
        Self { operand_gradient, gradient, p, noise, status }
    }
}

impl<D: Dimension> Backward for DropoutBackward<D> {
    fn backward(&self) {
        let (g, noise) = (self.gradient.borrow(), self.noise.borrow());
        let mut dx = self.operand_gradient.borrow_mut();
        ffi::check(unsafe {
            ffi::nk_dropout_bwd(g.device().as_raw(), dx.as_mut_ptr(), g.as_ptr(), noise.as_ptr(), g.len(), self.p, self.status.get() as i32)
        });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.operand_gradient)]
    }
}

/// `LogSoftmax::forward` (`node/logsoftmax/mod.rs:37-53`): `y = x - ln(sum exp(x - m)) - m` along `axis`.
pub(crate) struct LogSoftmax<D: Dimension> {
    operand_data: Shared<HipArray<D>>,
    data: Shared<HipArray<D>>,
    axis: usize,
}

impl<D: Dimension> LogSoftmax<D> {
    pub(crate) fn new(operand_data: Shared<HipArray<D>>, data: Shared<HipArray<D>>, axis: usize) -> Self {
        Self { operand_data, data, axis }
    }
}

impl<D: Dimension> Forward for LogSoftmax<D> {
    fn forward(&self) {
        let x = self.operand_data.borrow();
        let mut y = self.data.borrow_mut();
        let s = x.shape_c();
        ffi::check(unsafe { ffi::nk_log_softmax_fwd(x.device().as_raw(), x.as_ptr(), y.as_mut_ptr(), s.as_ptr(), s.len() as i32, self.axis as i32) });
    }
}

/// `LogSoftmaxBackward::backward` (`:84-102`): `dx += g - exp(y) * sum(g)` per lane.
pub(crate) struct LogSoftmaxBackward<D: Dimension> {
    operand_gradient: Rc<Gradient<HipArray<D>, D>>,
    data: Shared<HipArray<D>>,
    gradient: Rc<Gradient<HipArray<D>, D>>,
    axis: usize,
}

impl<D: Dimension> LogSoftmaxBackward<D> {
    pub(crate) fn new(operand_gradient: Rc<Gradient<HipArray<D>, D>>, data: Shared<HipArray<D>>, gradient: Rc<Gradient<HipArray<D>, D>>,
                      axis: usize) -> Self {
        Self { operand_gradient, data, gradient, axis }
    }
}

impl<D: Dimension> Backward for LogSoftmaxBackward<D> {
    fn backward(&self) {
        let (g, y) = (self.gradient.borrow(), self.data.borrow());
        let mut dx = self.operand_gradient.borrow_mut();
        let s = y.shape_c();
        ffi::check(unsafe {
            ffi::nk_log_softmax_bwd(g.device().as_raw(), dx.as_mut_ptr(), g.as_ptr(), y.as_ptr(), s.as_ptr(), s.len() as i32, self.axis as i32)
        });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.operand_gradient)]
    }
}
