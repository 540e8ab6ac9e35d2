use super::grad_id;
use std::rc::Rc;

use ndarray::Ix2;

use crate::{
    autograd::{Backward, Forward},
    gradient::Gradient,
    hip::{ffi, hiparray::HipArray},
    utils::Shared,
};

/// `MatrixMatrixMul::forward` (`node/matrix_matrix_mul/mod.rs:31-41`): `C(n,o) = A(n,m) . B(m,o)`, beta = 0.
pub(crate) struct MatrixMatrixMul {
    left_data: Shared<HipArray<Ix2>>,
    right_data: Shared<HipArray<Ix2>>,
    data: Shared<HipArray<Ix2>>,
}

impl MatrixMatrixMul {
    pub(crate) fn new(left_data: Shared<HipArray<Ix2>>, right_data: Shared<HipArray<Ix2>>, data: Shared<HipArray<Ix2>>) -> Self {
        Self { left_data, right_data, data }
    }
}

impl Forward for MatrixMatrixMul {
    fn forward(&self) {
        let (a, b) = (self.left_data.borrow(), self.right_data.borrow());
        let mut c = self.data.borrow_mut();
        let (n, m, o) = (a.dimension()[0] as i32, a.dimension()[1] as i32, b.dimension()[1] as i32);
        ffi::check(unsafe { ffi::nk_mm_fwd(a.device().as_raw(), a.as_ptr(), b.as_ptr(), c.as_mut_ptr(), n, m, o) });
    }
}

/// `MatrixMatrixMulBackwardLeft::backward` (`:63-73`): `dA += G . B^T` (NT, beta = 1).
pub(crate) struct MatrixMatrixMulBackwardLeft {
    right_data: Shared<HipArray<Ix2>>,
    left_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
    gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
}

impl MatrixMatrixMulBackwardLeft {
    pub(crate) fn new(right_data: Shared<HipArray<Ix2>>, left_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>, gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>) -> Self {
        Self { right_data, left_gradient, gradient }
    }
}

impl Backward for MatrixMatrixMulBackwardLeft {
    fn backward(&self) {
        let (g, b) = (self.gradient.borrow(), self.right_data.borrow());
        let mut da = self.left_gradient.borrow_mut();
        let (n, o, m) = (g.dimension()[0] as i32, g.dimension()[1] as i32, b.dimension()[0] as i32);
        ffi::check(unsafe { ffi::nk_mm_bwd_left(g.device().as_raw(), da.as_mut_ptr(), g.as_ptr(), b.as_ptr(), n, m, o) });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.left_gradient)]
    }
}

/// `MatrixMatrixMulBackwardRight::backward` (`:95-105`): `dB += A^T . G` (TN, beta = 1).
pub(crate) struct MatrixMatrixMulBackwardRight {
    left_data: Shared<HipArray<Ix2>>,
    right_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
    gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
}

impl MatrixMatrixMulBackwardRight {
    pub(crate) fn new(left_data: Shared<HipArray<Ix2>>, right_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>, gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>) -> Self {
        Self { left_data, right_gradient, gradient }
    }
}

impl Backward for MatrixMatrixMulBackwardRight {
    fn backward(&self) {
        let (g, a) = (self.gradient.borrow(), self.left_data.borrow());
        let mut db = self.right_gradient.borrow_mut();
        let (n, m, o) = (a.dimension()[0] as i32, a.dimension()[1] as i32, g.dimension()[1] as i32);
        ffi::check(unsafe { ffi::nk_mm_bwd_right(g.device().as_raw(), db.as_mut_ptr(), a.as_ptr(), g.as_ptr(), n, m, o) });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.right_gradient)]
    }
}

/// `MatrixMatrixMulBackward` (`:107-126`): the node's two halves as ONE call - `nk_mm_bwd` puts both products into a single
/// launch when neither fills the chip by itself (1024^3: 43.7 -> 39.1 us) and falls back to two launches otherwise, also when
/// both gradients are one buffer (`x.mm(x)`).  Same bits as `left.backward(); right.backward()` without k-pair blocks.
pub(crate) struct MatrixMatrixMulBackward {
    left: MatrixMatrixMulBackwardLeft,
    right: MatrixMatrixMulBackwardRight,
}

impl MatrixMatrixMulBackward {
    pub(crate) fn new(left: MatrixMatrixMulBackwardLeft, right: MatrixMatrixMulBackwardRight) -> Self {
        Self { left, right }
    }
}

impl Backward for MatrixMatrixMulBackward {
    fn backward(&self) {
        if Rc::ptr_eq(&self.left.left_gradient, &self.right.right_gradient) {
            // one gradient, two contributions: the RefCell allows one mutable borrow at a time
            self.left.backward();
            self.right.backward();
            return;
        }
        let g = self.left.gradient.borrow();
        let (a, b) = (self.right.left_data.borrow(), self.left.right_data.borrow());
        let (mut da, mut db) = (self.left.left_gradient.borrow_mut(), self.right.right_gradient.borrow_mut());
        let (n, m, o) = (a.dimension()[0] as i32, a.dimension()[1] as i32, b.dimension()[1] as i32);
        ffi::check(unsafe {
            ffi::nk_mm_bwd(g.device().as_raw(), da.as_mut_ptr(), db.as_mut_ptr(), g.as_ptr(), a.as_ptr(), b.as_ptr(), n, m, o, 0, 0)
        });
    }

    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.left.left_gradient), grad_id(&self.right.right_gradient)]
    }
}
