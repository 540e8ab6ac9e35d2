use super::grad_id;
use std::rc::Rc;

use ndarray::{Dimension, Ix0};

use crate::{
    autograd::{Backward, Forward},
    gradient::Gradient,
    hip::{ffi, hiparray::HipArray},
    utils::Shared,
    Reduction,
};

/// `Sum::forward` (`node/sum/mod.rs:28-35`): full reduction to a scalar (two-pass, fixed order: deterministic).
pub(crate) struct Sum<D: Dimension> {
    operand_data: Shared<HipArray<D>>,
    data: Shared<HipArray<Ix0>>,
}

impl<D: Dimension> Sum<D> {
    pub(crate) fn new(operand_data: Shared<HipArray<D>>, data: Shared<HipArray<Ix0>>) -> Self {
        Self { operand_data, data }
    }
}

impl<D: Dimension> Forward for Sum<D> {
    fn forward(&self) {
        let x = self.operand_data.borrow();
        let mut out = self.data.borrow_mut();
        ffi::check(unsafe { ffi::nk_sum_fwd(x.device().as_raw(), x.as_ptr(), x.len(), out.as_mut_ptr()) });
    }
}

/// `SumBackward::backward` (`:60-67`): `dx += g`.
pub(crate) struct SumBackward<D: Dimension> {
    operand_gradient: Rc<Gradient<HipArray<D>, D>>,
    gradient: Rc<Gradient<HipArray<Ix0>, Ix0>>,
}

impl<D: Dimension> SumBackward<D> {
    pub(crate) fn new(operand_gradient: Rc<Gradient<HipArray<D>, D>>, gradient: Rc<Gradient<HipArray<Ix0>, Ix0>>) -> Self {
        Self { operand_gradient, gradient }
    }
}

impl<D: Dimension> Backward for SumBackward<D> {
    fn backward(&self) {
        let g = self.gradient.borrow();
        let mut dx = self.operand_gradient.borrow_mut();
        let n = dx.len();
        ffi::check(unsafe { ffi::nk_sum_bwd(g.device().as_raw(), dx.as_mut_ptr(), n, g.as_ptr()) });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.operand_gradient)]
    }
}

/// `SquaredError::forward` (`node/squared_error/mod.rs:42-59`): `sum((x - t)^2)` (/ n for `Reduction::Mean`).
pub(crate) struct SquaredError<D: Dimension> {
    input_data: Shared<HipArray<D>>,
    target_data: Shared<HipArray<D>>,
    data: Shared<HipArray<Ix0>>,
    reduction: Reduction,
}

impl<D: Dimension> SquaredError<D> {
    pub(crate) fn new(input_data: Shared<HipArray<D>>, target_data: Shared<HipArray<D>>, data: Shared<HipArray<Ix0>>, reduction: Reduction) -> Self {
        Self { input_data, target_data, data, reduction }
    }
}

impl<D: Dimension> Forward for SquaredError<D> {
    fn forward(&self) {
        let (x, t) = (self.input_data.borrow(), self.target_data.borrow());
        let mut out = self.data.borrow_mut();
        let red = matches!(self.reduction, Reduction::Mean) as i32;
        ffi::check(unsafe { ffi::nk_mse_fwd(x.device().as_raw(), x.as_ptr(), t.as_ptr(), x.len(), red, out.as_mut_ptr()) });
    }
}

/// `SquaredErrorBackward::backward` (`:94-123`): `dx += 2 (x - t) g (/ n)`.
pub(crate) struct SquaredErrorBackward<D: Dimension> {
    input_data: Shared<HipArray<D>>,
    target_data: Shared<HipArray<D>>,
    input_gradient: Rc<Gradient<HipArray<D>, D>>,
    gradient: Rc<Gradient<HipArray<Ix0>, Ix0>>,
    reduction: Reduction,
}

impl<D: Dimension> SquaredErrorBackward<D> {
    pub(crate) fn new(input_data: Shared<HipArray<D>>, target_data: Shared<HipArray<D>>, input_gradient: Rc<Gradient<HipArray<D>, D>>, gradient: Rc<Gradient<HipArray<Ix0>, Ix0>>, reduction: Reduction) -> Self {
        Self { input_data, target_data, input_gradient, gradient, reduction }
    }
}

impl<D: Dimension> Backward for SquaredErrorBackward<D> {
    fn backward(&self) {
        let (g, x, t) = (self.gradient.borrow(), self.input_data.borrow(), self.target_data.borrow());
        let mut dx = self.input_gradient.borrow_mut();
        let red = matches!(self.reduction, Reduction::Mean) as i32;
        ffi::check(unsafe { ffi::nk_mse_bwd(g.device().as_raw(), dx.as_mut_ptr(), g.as_ptr(), x.as_ptr(), t.as_ptr(), x.len(), red) });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.input_gradient)]
    }
}

/// `Mean::forward` (`node/mean/mod.rs:28-35`): `sum / len`.
pub(crate) struct Mean<D: Dimension> {
    operand_data: Shared<HipArray<D>>,
    data: Shared<HipArray<Ix0>>,
}

impl<D: Dimension> Mean<D> {
    pub(crate) fn new(operand_data: Shared<HipArray<D>>, data: Shared<HipArray<Ix0>>) -> Self {
        Self { operand_data, data }
    }
}

impl<D: Dimension> Forward for Mean<D> {
    fn forward(&self) {
        let x = self.operand_data.borrow();
        let mut out = self.data.borrow_mut();
        ffi::check(unsafe { ffi::nk_mean_fwd(x.device().as_raw(), x.as_ptr(), x.len(), out.as_mut_ptr()) });
    }
}

/// `MeanBackward::backward` (`:60-72`): `dx += g / len`.
pub(crate) struct MeanBackward<D: Dimension> {
    operand_gradient: Rc<Gradient<HipArray<D>, D>>,
    gradient: Rc<Gradient<HipArray<Ix0>, Ix0>>,
}

impl<D: Dimension> MeanBackward<D> {
    pub(crate) fn new(operand_gradient: Rc<Gradient<HipArray<D>, D>>, gradient: Rc<Gradient<HipArray<Ix0>, Ix0>>) -> Self {
        Self { operand_gradient, gradient }
    }
}

impl<D: Dimension> Backward for MeanBackward<D> {
    fn backward(&self) {
        let g = self.gradient.borrow();
        let mut dx = self.operand_gradient.borrow_mut();
        let n = dx.len();
        ffi::check(unsafe { ffi::nk_mean_bwd(g.device().as_raw(), dx.as_mut_ptr(), n, g.as_ptr()) });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.operand_gradient)]
    }
}
