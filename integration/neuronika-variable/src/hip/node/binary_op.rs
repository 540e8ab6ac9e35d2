use super::grad_id;
use std::rc::Rc;

use ndarray::{DimMax, Dimension};

use crate::{
    autograd::{Backward, Forward},
    gradient::Gradient,
    hip::{ffi, hiparray::HipArray},
    utils::{Broadcast, Shared},
};

/// `op` argument of `nk_binary_*` (`NK_ADD` .. `NK_DIV` in the header).
#[derive(Clone, Copy)]
pub(crate) enum BinaryOp {
    Add = 0,
    Sub = 1,
    Mul = 2,
    Div = 3,
}

/// `Addition / Subtraction / Multiplication / Division::forward` (`node/addition/mod.rs:39-50` and siblings): the
/// `Zip::for_each` body becomes `nk_binary_fwd`.
pub(crate) struct BinaryOperation<D, E>
where
    D: Dimension + DimMax<E>,
    E: Dimension,
{
    op: BinaryOp,
    left_data: Shared<HipArray<D>>,
    right_data: Shared<HipArray<E>>,
    data: Shared<HipArray<Broadcast<D, E>>>,
}

impl<D, E> BinaryOperation<D, E>
where
    D: Dimension + DimMax<E>,
    E: Dimension,
{
    pub(crate) fn new(op: BinaryOp, left_data: Shared<HipArray<D>>, right_data: Shared<HipArray<E>>,
                      data: Shared<HipArray<Broadcast<D, E>>>) -> Self {
        Self { op, left_data, right_data, data }
    }
}

impl<D, E> Forward for BinaryOperation<D, E>
where
    D: Dimension + DimMax<E>,
    E: Dimension,
{
    fn forward(&self) {
        let (l, r) = (self.left_data.borrow(), self.right_data.borrow());
        let mut out = self.data.borrow_mut();
        let (ls, rs, os) = (l.shape_c(), r.shape_c(), out.shape_c());
        ffi::check(unsafe {
            ffi::nk_binary_fwd(l.device().as_raw(), self.op as i32, out.as_mut_ptr(), os.as_ptr(), os.len() as i32, l.as_ptr(),
                               ls.as_ptr(), ls.len() as i32, r.as_ptr(), rs.as_ptr(), rs.len() as i32)
        });
    }
}

/// `*BackwardLeft::backward` (`node/addition/mod.rs:81-106`, `multiplication/mod.rs:85-115`, ...): local gradient and
/// the un-broadcast reduction (`utils.rs:152-192`, intended semantics) in one call, accumulated (`+=`).
pub(crate) struct BinaryOperationBackwardLeft<D, E>
where
    D: Dimension + DimMax<E>,
    E: Dimension,
{
    op: BinaryOp,
    right_data: Shared<HipArray<E>>,
    left_gradient: Rc<Gradient<HipArray<D>, D>>,
    gradient: Rc<Gradient<HipArray<Broadcast<D, E>>, Broadcast<D, E>>>,
}

impl<D, E> BinaryOperationBackwardLeft<D, E>
where
    D: Dimension + DimMax<E>,
    E: Dimension, {
    pub(crate) fn new(op: BinaryOp, right_data: Shared<HipArray<E>>, left_gradient: Rc<Gradient<HipArray<D>, D>>, gradient: Rc<Gradient<HipArray<Broadcast<D, E>>, Broadcast<D, E>>>) -> Self {
        Self { op, right_data, left_gradient, gradient }
    }
}

impl<D, E> Backward for BinaryOperationBackwardLeft<D, E>
where
    D: Dimension + DimMax<E>,
    E: Dimension,
{
    fn backward(&self) {
        let (g, r) = (self.gradient.borrow(), self.right_data.borrow());
        let mut d = self.left_gradient.borrow_mut();
        let (ds, gs, rs) = (d.shape_c(), g.shape_c(), r.shape_c());
        ffi::check(unsafe {
            ffi::nk_binary_bwd_left(g.device().as_raw(), self.op as i32, d.as_mut_ptr(), ds.as_ptr(), ds.len() as i32, g.as_ptr(),
                                    gs.as_ptr(), gs.len() as i32, r.as_ptr(), rs.as_ptr(), rs.len() as i32)
        });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.left_gradient)]
    }
}

/// `*BackwardRight::backward` (`node/subtraction/mod.rs:110-136`, `division/mod.rs:134-149`, ...).
pub(crate) struct BinaryOperationBackwardRight<D, E>
where
    D: Dimension + DimMax<E>,
    E: Dimension,
{
    op: BinaryOp,
    left_data: Shared<HipArray<D>>,
    right_data: Shared<HipArray<E>>,
    right_gradient: Rc<Gradient<HipArray<E>, E>>,
    gradient: Rc<Gradient<HipArray<Broadcast<D, E>>, Broadcast<D, E>>>,
}

impl<D, E> BinaryOperationBackwardRight<D, E>
where
    D: Dimension + DimMax<E>,
    E: Dimension, {
    pub(crate) fn new(op: BinaryOp, left_data: Shared<HipArray<D>>, right_data: Shared<HipArray<E>>, right_gradient: Rc<Gradient<HipArray<E>, E>>, gradient: Rc<Gradient<HipArray<Broadcast<D, E>>, Broadcast<D, E>>>) -> Self {
        Self { op, left_data, right_data, right_gradient, gradient }
    }
}

impl<D, E> Backward for BinaryOperationBackwardRight<D, E>
where
    D: Dimension + DimMax<E>,
    E: Dimension,
{
    fn backward(&self) {
        let (g, l, r) = (self.gradient.borrow(), self.left_data.borrow(), self.right_data.borrow());
        let mut d = self.right_gradient.borrow_mut();
        let (ds, gs, ls) = (d.shape_c(), g.shape_c(), l.shape_c());
        ffi::check(unsafe {
            ffi::nk_binary_bwd_right(g.device().as_raw(), self.op as i32, d.as_mut_ptr(), ds.as_ptr(), ds.len() as i32, g.as_ptr(),
                                     gs.as_ptr(), gs.len() as i32, l.as_ptr(), ls.as_ptr(), ls.len() as i32, r.as_ptr())
        });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.right_gradient)]
    }
}
