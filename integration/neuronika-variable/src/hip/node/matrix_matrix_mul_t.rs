use super::grad_id;
use std::rc::Rc;

use ndarray::Ix2;

use crate::{
    autograd::{Backward, Forward},
    gradient::Gradient,
    hip::{ffi, hiparray::HipArray},
    utils::Shared,
};

/// `MatrixMatrixMulT::forward` (`node/matrix_matrix_mul_t/mod.rs:31-41`): `general_mat_mul(1, A, B^T, 0, C)` becomes
/// `nk_mm_t_fwd` (f32 MFMA GEMM, NT layout).
pub(crate) struct MatrixMatrixMulT {
    left_data: Shared<HipArray<Ix2>>,
    right_data: Shared<HipArray<Ix2>>,
    data: Shared<HipArray<Ix2>>,
}

impl MatrixMatrixMulT {
    pub(crate) fn new(left_data: Shared<HipArray<Ix2>>, right_data: Shared<HipArray<Ix2>>, data: Shared<HipArray<Ix2>>) -> Self {
        Self { left_data, right_data, data }
    }
}

impl Forward for MatrixMatrixMulT {
    fn forward(&self) {
        let (a, b) = (self.left_data.borrow(), self.right_data.borrow());
        let mut c = self.data.borrow_mut();
        let (n, m, o) = (a.dimension()[0] as i32, a.dimension()[1] as i32, b.dimension()[0] as i32);
        ffi::check(unsafe { ffi::nk_mm_t_fwd(a.device().as_raw(), a.as_ptr(), b.as_ptr(), c.as_mut_ptr(), n, m, o) });
    }
}

/// `MatrixMatrixMulTBackwardLeft::backward` (`:63-73`): `dA += G . B` (NN, beta = 1).
pub(crate) struct MatrixMatrixMulTBackwardLeft {
    right_data: Shared<HipArray<Ix2>>,
    left_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
    gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
}

impl MatrixMatrixMulTBackwardLeft {
    pub(crate) fn new(right_data: Shared<HipArray<Ix2>>, left_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
                      gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>) -> Self {
        Self { right_data, left_gradient, gradient }
    }
}

impl Backward for MatrixMatrixMulTBackwardLeft {
    fn backward(&self) {
        let (g, b) = (self.gradient.borrow(), self.right_data.borrow());
        let mut da = self.left_gradient.borrow_mut();
        let (n, o, m) = (g.dimension()[0] as i32, g.dimension()[1] as i32, b.dimension()[1] as i32);
        ffi::check(unsafe { ffi::nk_mm_t_bwd_left(g.device().as_raw(), da.as_mut_ptr(), g.as_ptr(), b.as_ptr(), n, m, o) });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.left_gradient)]
    }
}

/// `MatrixMatrixMulTBackwardRight::backward` (`:95-105`): `dB += G^T . A` (TN, beta = 1).
pub(crate) struct MatrixMatrixMulTBackwardRight {
    left_data: Shared<HipArray<Ix2>>,
    right_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
    gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
}

impl MatrixMatrixMulTBackwardRight {
    pub(crate) fn new(left_data: Shared<HipArray<Ix2>>, right_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
                      gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>) -> Self {
        Self { left_data, right_gradient, gradient }
    }
}

impl Backward for MatrixMatrixMulTBackwardRight {
    fn backward(&self) {
        let (g, a) = (self.gradient.borrow(), self.left_data.borrow());
        let mut db = self.right_gradient.borrow_mut();
        let (n, o, m) = (g.dimension()[0] as i32, g.dimension()[1] as i32, a.dimension()[1] as i32);
        ffi::check(unsafe { ffi::nk_mm_t_bwd_right(g.device().as_raw(), db.as_mut_ptr(), g.as_ptr(), a.as_ptr(), n, m, o) });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.right_gradient)]
    }
}

/// `MatrixMatrixMulTBackward` (`:107-126`): both halves as ONE call of `nk_mm_t_bwd` (see `MatrixMatrixMulBackward`).
pub(crate) struct MatrixMatrixMulTBackward {
    left: MatrixMatrixMulTBackwardLeft,
    right: MatrixMatrixMulTBackwardRight,
}

impl MatrixMatrixMulTBackward {
    pub(crate) fn new(left: MatrixMatrixMulTBackwardLeft, right: MatrixMatrixMulTBackwardRight) -> Self {
        Self { left, right }
    }
}

impl Backward for MatrixMatrixMulTBackward {
    fn backward(&self) {
        if Rc::ptr_eq(&self.left.left_gradient, &self.right.right_gradient) {
            self.left.backward();
            self.right.backward();
            return;
        }
        let g = self.left.gradient.borrow();
        let (a, b) = (self.right.left_data.borrow(), self.left.right_data.borrow());
        let (mut da, mut db) = (self.left.left_gradient.borrow_mut(), self.right.right_gradient.borrow_mut());
        let (n, m, o) = (a.dimension()[0] as i32, a.dimension()[1] as i32, b.dimension()[0] as i32);
        ffi::check(unsafe {
            ffi::nk_mm_t_bwd(g.device().as_raw(), da.as_mut_ptr(), db.as_mut_ptr(), g.as_ptr(), a.as_ptr(), b.as_ptr(), n, m, o, 0, 0)
        });
    }

    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.left.left_gradient), grad_id(&self.right.right_gradient)]
    }
}
