use super::grad_id;
use std::rc::Rc;

use ndarray::Dimension;

use crate::{
    autograd::{Backward, Forward},
    gradient::Gradient,
    hip::{ffi, hiparray::HipArray},
    utils::Shared,
};

/// `Transpose::forward` (`node/transpose/mod.rs:28-37`): reversed axes, materialised (the reference assigns `view.t()`).
pub(crate) struct Transpose<D: Dimension> {
    operand_data: Shared<HipArray<D>>,
    data: Shared<HipArray<D>>,
}

impl<D: Dimension> Transpose<D> {
    pub(crate) fn new(operand_data: Shared<HipArray<D>>, data: Shared<HipArray<D>>) -> Self {
        Self { operand_data, data }
    }
}

impl<D: Dimension> Forward for Transpose<D> {
    fn forward(&self) {
        let x = self.operand_data.borrow();
        let mut y = self.data.borrow_mut();
        let s = x.shape_c();
        ffi::check(unsafe { ffi::nk_transpose_fwd(x.device().as_raw(), x.as_ptr(), y.as_mut_ptr(), s.as_ptr(), s.len() as i32) });
    }
}

/// `TransposeBackward::backward` (`:62-69`): `dx += g^T`.
pub(crate) struct TransposeBackward<D: Dimension> {
    operand_gradient: Rc<Gradient<HipArray<D>, D>>,
    gradient: Rc<Gradient<HipArray<D>, D>>,
}

impl<D: Dimension> TransposeBackward<D> {
    pub(crate) fn new(operand_gradient: Rc<Gradient<HipArray<D>, D>>, gradient: Rc<Gradient<HipArray<D>, D>>) -> Self {
        Self { operand_gradient, gradient }
    }
}

impl<D: Dimension> Backward for TransposeBackward<D> {
    fn backward(&self) {
        let g = self.gradient.borrow();
        let mut dx = self.operand_gradient.borrow_mut();
        let s = dx.shape_c();
        ffi::check(unsafe { ffi::nk_transpose_bwd(g.device().as_raw(), dx.as_mut_ptr(), g.as_ptr(), s.as_ptr(), s.len() as i32) });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.operand_gradient)]
    }
}

/// `Chunk::forward` (`node/chunk/mod.rs:48-64`): chunk number `chunk_no` of `exact_chunks(chunk_shape)`, row-major chunk order.
pub(crate) struct Chunk<D: Dimension> {
    operand_data: Shared<HipArray<D>>,
    data: Shared<HipArray<D>>,
    chunk_no: usize,
}

impl<D: Dimension> Chunk<D> {
    pub(crate) fn new(operand_data: Shared<HipArray<D>>, data: Shared<HipArray<D>>, chunk_no: usize) -> Self {
        Self { operand_data, data, chunk_no }
    }
}

impl<D: Dimension> Forward for Chunk<D> {
    fn forward(&self) {
        let x = self.operand_data.borrow();
        let mut y = self.data.borrow_mut();
        let (xs, cs) = (x.shape_c(), y.shape_c());
        ffi::check(unsafe {
            ffi::nk_chunk_fwd(x.device().as_raw(), x.as_ptr(), xs.as_ptr(), y.as_mut_ptr(), cs.as_ptr(), xs.len() as i32, self.chunk_no as i32)
        });
    }
}

/// `ChunkBackward::backward` (`:99-113`): `dx[chunk] += g`.
pub(crate) struct ChunkBackward<D: Dimension> {
    operand_gradient: Rc<Gradient<HipArray<D>, D>>,
    gradient: Rc<Gradient<HipArray<D>, D>>,
    chunk_no: usize,
}

impl<D: Dimension> ChunkBackward<D> {
    pub(crate) fn new(operand_gradient: Rc<Gradient<HipArray<D>, D>>, gradient: Rc<Gradient<HipArray<D>, D>>, chunk_no: usize) -> Self {
        Self { operand_gradient, gradient, chunk_no }
    }
}

impl<D: Dimension> Backward for ChunkBackward<D> {
    fn backward(&self) {
        let g = self.gradient.borrow();
        let mut dx = self.operand_gradient.borrow_mut();
        let (xs, cs) = (dx.shape_c(), g.shape_c());
        ffi::check(unsafe {
            ffi::nk_chunk_bwd(g.device().as_raw(), dx.as_mut_ptr(), xs.as_ptr(), g.as_ptr(), cs.as_ptr(), xs.len() as i32, self.chunk_no as i32)
        });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.operand_gradient)]
    }
}

/// `MultiConcatenate::forward` (`node/multi_concatenate/mod.rs:37-50`): one strided copy per operand into its slice of `axis`.
pub(crate) struct MultiConcatenate<D: Dimension> {
    operands_data: Vec<Shared<HipArray<D>>>,
    data: Shared<HipArray<D>>,
    axis: usize,
}

impl<D: Dimension> MultiConcatenate<D> {
    pub(crate) fn new(operands_data: Vec<Shared<HipArray<D>>>, data: Shared<HipArray<D>>, axis: usize) -> Self {
        Self { operands_data, data, axis }
    }
}

impl<D: Dimension> Forward for MultiConcatenate<D> {
    fn forward(&self) {
        let mut out = self.data.borrow_mut();
        let os = out.shape_c();
        let mut offset = 0i32;
        for operand in &self.operands_data {
            let x = operand.borrow();
            let len = x.shape_c()[self.axis];
            ffi::check(unsafe {
                ffi::nk_concat_fwd_part(x.device().as_raw(), x.as_ptr(), out.as_mut_ptr(), os.as_ptr(), os.len() as i32, self.axis as i32, offset, len)
            });
            offset += len;
        }
    }
}

/// `MultiConcatenateBackward::backward` (`:81-97`): every operand gradient `+=` its slice of `g`.
pub(crate) struct MultiConcatenateBackward<D: Dimension> {
    operands_gradients: Vec<Rc<Gradient<HipArray<D>, D>>>,
    gradient: Rc<Gradient<HipArray<D>, D>>,
    axis: usize,
}

impl<D: Dimension> MultiConcatenateBackward<D> {
    pub(crate) fn new(operands_gradients: Vec<Rc<Gradient<HipArray<D>, D>>>, gradient: Rc<Gradient<HipArray<D>, D>>, axis: usize) -> Self {
        Self { operands_gradients, gradient, axis }
    }
}

impl<D: Dimension> Backward for MultiConcatenateBackward<D> {
    fn backward(&self) {
        let g = self.gradient.borrow();
        let gs = g.shape_c();
        let mut offset = 0i32;
        for operand in &self.operands_gradients {
            let mut d = operand.borrow_mut();
            let len = d.shape_c()[self.axis];
            ffi::check(unsafe {
                ffi::nk_concat_bwd_part(g.device().as_raw(), d.as_mut_ptr(), g.as_ptr(), gs.as_ptr(), gs.len() as i32, self.axis as i32, offset, len)
            });
            offset += len;
        }
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        self.operands_gradients.iter().map(grad_id).collect()
    }
}

/// The four `PaddingMode`s of `node/pad/` (`constant/mod.rs:14-39`, `zero` = `Constant(0.)`, `reflective/mod.rs:9-136`,
/// `replicative/mod.rs:9-134`).
#[derive(Clone, Copy)]
pub(crate) enum PadMode {
    Constant(f32),
    Reflective,
    Replicative,
}

/// `Pad::forward` (`node/pad/mod.rs:97-129`): `padding[i]` elements on both sides of spatial axis i.
pub(crate) struct Pad<D: Dimension> {
    operand_data: Shared<HipArray<D>>,
    data: Shared<HipArray<D>>,
    mode: PadMode,
    padding: Vec<i32>,
}

impl<D: Dimension> Pad<D> {
    pub(crate) fn new(operand_data: Shared<HipArray<D>>, data: Shared<HipArray<D>>, mode: PadMode, padding: Vec<i32>) -> Self {
        Self { operand_data, data, mode, padding }
    }
}

impl<D: Dimension> Forward for Pad<D> {
    fn forward(&self) {
        let x = self.operand_data.borrow();
        let mut y = self.data.borrow_mut();
        let xs = x.shape_c();
        let (dev, nd) = (x.device().as_raw(), xs.len() as i32 - 2);
        ffi::check(unsafe {
            match self.mode {
                PadMode::Constant(value) => ffi::nk_pad_const_fwd(dev, nd, x.as_ptr(), xs.as_ptr(), y.as_mut_ptr(), self.padding.as_ptr(), value),
                PadMode::Reflective => ffi::nk_pad_reflective_fwd(dev, nd, x.as_ptr(), xs.as_ptr(), y.as_mut_ptr(), self.padding.as_ptr()),
                PadMode::Replicative => ffi::nk_pad_replicative_fwd(dev, nd, x.as_ptr(), xs.as_ptr(), y.as_mut_ptr(), self.padding.as_ptr()),
            }
        });
    }
}

/// `PadBackward::backward` (`node/pad/mod.rs:157-181`): `dx += centre(g)` for every mode.
pub(crate) struct PadBackward<D: Dimension> {
    operand_gradient: Rc<Gradient<HipArray<D>, D>>,
    gradient: Rc<Gradient<HipArray<D>, D>>,
    padding: Vec<i32>,
}

impl<D: Dimension> PadBackward<D> {
    pub(crate) fn new(operand_gradient: Rc<Gradient<HipArray<D>, D>>, gradient: Rc<Gradient<HipArray<D>, D>>, padding: Vec<i32>) -> Self {
        Self { operand_gradient, gradient, padding }
    }
}

impl<D: Dimension> Backward for PadBackward<D> {
    fn backward(&self) {
        let g = self.gradient.borrow();
        let mut dx = self.operand_gradient.borrow_mut();
        let xs = dx.shape_c();
        ffi::check(unsafe { ffi::nk_pad_bwd(g.device().as_raw(), xs.len() as i32 - 2, dx.as_mut_ptr(), xs.as_ptr(), g.as_ptr(), self.padding.as_ptr()) });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.operand_gradient)]
    }
}
