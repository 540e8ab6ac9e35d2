use std::{
    cell::{Cell, RefCell},
    rc::Rc,
};

use ndarray::{DimMax, Dimension, Ix2};

use super::{
    hiparray::HipArray,
    node::{BinaryOp, BinaryOperation, BinaryOperationBackwardLeft, BinaryOperationBackwardRight, MatrixMatrixMulT,
           MatrixMatrixMulTBackwardLeft, MatrixMatrixMulTBackwardRight},
};
use crate::{
    autograd::{Backward, Forward},
    gradient::{Gradient, NoGrad},
    history::History,
    utils::{cobroadcast, Broadcast, Shared},
};

/// A non-differentiable variable with data in HBM.  Same fields and tape as `Var<D>` (`var.rs:34-40`) /
/// `CuVar<D>` (`cuda/cuvar.rs:19-46`): only the array type differs.
pub struct HipVar<D>
where
    D: Dimension,
{
    pub(crate) data: Shared<HipArray<D>>,
    pub(crate) history: History<(Rc<dyn Forward>, Cell<bool>)>,
}

impl<D> HipVar<D>
where
    D: Dimension,
{
    pub(crate) fn leaf(array: HipArray<D>) -> Self {
        Self { data: Rc::new(RefCell::new(array)), history: History::default() }
    }

    pub(crate) fn node(data: Shared<HipArray<D>>, op: Rc<dyn Forward>, mut history: History<(Rc<dyn Forward>, Cell<bool>)>) -> Self {
        history.insert(Rc::as_ptr(&op) as *const () as usize, (op, Cell::default()));
        Self { data, history }
    }

    /// `Var::forward` (`var.rs:110-128`), verbatim logic: the ops are enqueued on the device's compute stream in
    /// tape order and return immediately; nothing synchronises until the host reads data back.
    pub fn forward(&self) {
        let mut buffer = self.history.buffer_mut();
        if buffer.is_empty() {
            *buffer = self.history.to_vec()
        } else {
            buffer.iter().for_each(|(_, computed)| computed.set(false));
        }
        buffer.iter().filter(|(_, computed)| !computed.get()).for_each(|(op, computed)| {
            op.forward();
            computed.set(true)
        });
    }

    /// Broadcast binary (`Addition` ... `Division`): shape rule `cobroadcast` (`utils.rs:97-125`) stays on the host.
    pub(crate) fn binary<E>(mut self, op: BinaryOp, rhs: HipVar<E>) -> HipVar<Broadcast<D, E>>
    where
        D: 'static + DimMax<E>,
        E: 'static + Dimension,
    {
        self.history.merge(rhs.history);
        let dim = cobroadcast(self.data.borrow().dimension(), rhs.data.borrow().dimension());
        let device = self.data.borrow().device().clone();
        let data = Rc::new(RefCell::new(HipArray::zeroed(dim, device)));
        let node = Rc::new(BinaryOperation::new(op, self.data, rhs.data, data.clone()));
        HipVar::node(data, node, self.history)
    }
}

/// A differentiable variable with data and gradient in HBM (`VarDiff<D>`, `vardiff.rs:35-42`).
pub struct HipVarDiff<D>
where
    D: Dimension,
{
    pub(crate) var: HipVar<D>,
    pub(crate) grad: Rc<Gradient<HipArray<D>, D>>,
    pub(crate) history: History<(Rc<dyn Backward>, Rc<dyn NoGrad>)>,
}

impl<D> HipVarDiff<D>
where
    D: 'static + Dimension,
{
    pub(crate) fn node(var: HipVar<D>, grad: Rc<Gradient<HipArray<D>, D>>, op: (Rc<dyn Backward>, Rc<dyn NoGrad>),
                       mut history: History<(Rc<dyn Backward>, Rc<dyn NoGrad>)>) -> Self {
        history.insert(Rc::as_ptr(&op.0) as *const () as usize, op);
        Self { var, grad, history }
    }

    pub fn forward(&self) {
        self.var.forward();
    }

    /// `VarDiff::backward` (`vardiff.rs:125-141`): seed the root gradient, run the tape in reverse.  Launches are
    /// asynchronous; the data-parallel hook (`dp::GradientSync`) is handed each leaf gradient as soon as the last node
    /// writing it has been issued.
    pub fn backward(&self, seed: f32) {
        debug_assert_eq!(self.var.history.len(), self.var.history.buffer_len(), "Perhaps you forgot to call .forward()?");
        self.grad.borrow_mut().fill(seed);
        let mut buffer = self.history.buffer_mut();
        if buffer.is_empty() {
            *buffer = self.history.to_vec();
        }
        buffer.iter().rev().for_each(|(op, _)| op.backward());
    }
}

impl HipVarDiff<Ix2> {
    /// `mm_t` (`vardiff.rs:1110-1143`): the node `nn::Linear::forward` is made of (`neuronika-nn/src/lib.rs:443-446`).
    pub fn mm_t(mut self, rhs: HipVarDiff<Ix2>) -> HipVarDiff<Ix2> {
        self.var.history.merge(rhs.var.history);
        self.history.merge(rhs.history);
        let (n, o) = (self.var.data.borrow().dimension()[0], rhs.var.data.borrow().dimension()[0]);
        let device = self.var.data.borrow().device().clone();
        let data = Rc::new(RefCell::new(HipArray::zeroed(ndarray::Dim([n, o]), device.clone())));
        let fwd = Rc::new(MatrixMatrixMulT::new(self.var.data.clone(), rhs.var.data.clone(), data.clone()));
        let var = HipVar::node(data, fwd, self.var.history);
        let grad = Rc::new(Gradient::hip_zeros(ndarray::Dim([n, o]), device));
        let left = MatrixMatrixMulTBackwardLeft::new(rhs.var.data.clone(), self.grad.clone(), grad.clone());
        let right = MatrixMatrixMulTBackwardRight::new(self.var.data.clone(), rhs.grad.clone(), grad.clone());
        let op: Rc<dyn Backward> = Rc::new(super::node::Pair(left, right));
        HipVarDiff::node(var, grad.clone(), (op, grad), self.history)
    }
}

impl<D, E> std::ops::Add<HipVar<E>> for HipVar<D>
where
    D: 'static + DimMax<E>,
    E: 'static + Dimension,
{
    type Output = HipVar<Broadcast<D, E>>;

    fn add(self, rhs: HipVar<E>) -> Self::Output {
        self.binary(BinaryOp::Add, rhs)
    }
}
