use super::grad_id;
use std::cell::{Cell, RefCell};
use std::rc::Rc;

use ndarray::{Ix1, Ix2};

use crate::{
    autograd::{Backward, Forward},
    gradient::Gradient,
    hip::{ffi, hiparray::HipArray},
    utils::Shared,
};

/// Mask state of a fused Linear+ReLU node's output gradient: `source` is the node's output `y` (the mask is `y > 0`, the same
/// bits as `x > 0` on the pre-activation, `node/relu/mod.rs:76`); `premasked` is decided per backward pass
/// (`HipVarDiff::backward`): true when every writer of that gradient on the tape applied the mask while storing.
pub struct ReluMask {
    pub(crate) source: Shared<HipArray<Ix2>>,
    pub(crate) premasked: Cell<bool>,
}

impl ReluMask {
    pub(crate) fn new(source: Shared<HipArray<Ix2>>) -> Self {
        Self { source, premasked: Cell::new(false) }
    }
}

/// `nn::Linear::forward` as ONE node: `MatrixMatrixMulT::forward` (`node/matrix_matrix_mul_t/mod.rs:31-41`) with the broadcast
/// `Addition` of the bias (`node/addition/mod.rs:31-50`) in the GEMM epilogue - `nk_linear_fwd` - and, with `relu`, the
/// `ReLU::forward` that follows it in the reference's words `lin.forward(x).relu()` (`node/relu/mod.rs:29-38`) in the same
/// epilogue - `nk_linear_relu_fwd`.  Bit-identical to the three nodes.
pub(crate) struct Linear {
    input_data: Shared<HipArray<Ix2>>,
    weight_data: Shared<HipArray<Ix2>>,
    bias_data: Shared<HipArray<Ix1>>,
    data: Shared<HipArray<Ix2>>,
    relu: bool,
}

impl Linear {
    pub(crate) fn new(input_data: Shared<HipArray<Ix2>>, weight_data: Shared<HipArray<Ix2>>, bias_data: Shared<HipArray<Ix1>>,
                      data: Shared<HipArray<Ix2>>, relu: bool) -> Self {
        Self { input_data, weight_data, bias_data, data, relu }
    }
}

impl Forward for Linear {
    fn forward(&self) {
        let (x, w, b) = (self.input_data.borrow(), self.weight_data.borrow(), self.bias_data.borrow());
        let mut y = self.data.borrow_mut();
        let (n, m, o) = (x.dimension()[0] as i32, x.dimension()[1] as i32, w.dimension()[0] as i32);
        if self.relu {
            ffi::check(unsafe { ffi::nk_linear_relu_fwd(x.device().as_raw(), x.as_ptr(), w.as_ptr(), b.as_ptr(), y.as_mut_ptr(), n, m, o) });
        } else {
            ffi::check(unsafe { ffi::nk_linear_fwd(x.device().as_raw(), x.as_ptr(), w.as_ptr(), b.as_ptr(), y.as_mut_ptr(), n, m, o) });
        }
    }
}

/// The backward entry of the fused node: `MatrixMatrixMulTBackwardLeft` (`:63-73`), `AdditionBackwardRight`
/// (`node/addition/mod.rs:113-135`: the bias gradient, an un-broadcast column sum) and `MatrixMatrixMulTBackwardRight`
/// (`:95-105`), in that order - the small gradient first, so that the data-parallel exchange can send the biases of the whole
/// model as one group in front of the last weight-gradient GEMMs.
/// * `output_mask` (Linear+ReLU): the node's gradient holds dL/dy; dL/dz = (y > 0) * dL/dy is what the three products need.
///   When the pass decided `premasked`, the writers already stored it; otherwise it is formed once into a scratch copy
///   (`nk_relu_bwd_assign`), the gradient buffer itself keeps dL/dy.
/// * `input_mask`: the input IS the output of a Linear+ReLU node; in a `premasked` pass the input gradient is written through
///   `nk_linear_bwd_input_relu` (the mask in the GEMM epilogue, no ReLU kernel at all).
pub(crate) struct LinearBackward {
    input_data: Shared<HipArray<Ix2>>,
    weight_data: Shared<HipArray<Ix2>>,
    output_mask: Option<Rc<ReluMask>>,
    input_mask: Option<Rc<ReluMask>>,
    input_gradient: Option<Rc<Gradient<HipArray<Ix2>, Ix2>>>,
    weight_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
    bias_gradient: Rc<Gradient<HipArray<Ix1>, Ix1>>,
    gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
    /// (y > 0) * dL/dy of a pass whose writers stored plain values.  Owned by the node: allocated the first time a pass needs it
    /// and kept, so that no later pass allocates (a captured hipGraph keeps a pointer that stays this node's).
    masked_scratch: RefCell<Option<HipArray<Ix2>>>,
}

impl LinearBackward {
    #[allow(clippy::too_many_arguments)]
    pub(crate) fn new(input_data: Shared<HipArray<Ix2>>, weight_data: Shared<HipArray<Ix2>>, output_mask: Option<Rc<ReluMask>>,
                      input_mask: Option<Rc<ReluMask>>, input_gradient: Option<Rc<Gradient<HipArray<Ix2>, Ix2>>>,
                      weight_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>, bias_gradient: Rc<Gradient<HipArray<Ix1>, Ix1>>,
                      gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>) -> Self {
        Self { input_data, weight_data, output_mask, input_mask, input_gradient, weight_gradient, bias_gradient, gradient,
               masked_scratch: RefCell::new(None) }
    }
}

impl Backward for LinearBackward {
    fn backward(&self) {
        let (x, w) = (self.input_data.borrow(), self.weight_data.borrow());
        let g = self.gradient.borrow();
        let dev = g.device().as_raw();
        let (n, m, o) = (x.dimension()[0] as i32, x.dimension()[1] as i32, w.dimension()[0] as i32);
        // dL/dz of a Linear+ReLU node whose writers stored plain values: one masking pass into a scratch copy
        let mut scratch = self.masked_scratch.borrow_mut();
        let mut gz: *const f32 = g.as_ptr();
        if let Some(mask) = &self.output_mask {
            if !mask.premasked.get() {
                let buf = scratch.get_or_insert_with(|| HipArray::zeroed(g.dimension(), g.device().clone()));
                let y = mask.source.borrow();
                ffi::check(unsafe { ffi::nk_relu_bwd_assign(dev, buf.as_mut_ptr(), g.as_ptr(), y.as_ptr(), g.len()) });
                gz = buf.as_ptr();
            }
        }
        if let Some(input_gradient) = &self.input_gradient {
            let mut dx = input_gradient.borrow_mut();
            let through_mask = self.input_mask.as_ref().map_or(false, |mask| mask.premasked.get());
            if through_mask {
                // dX += (x > 0) * (G . W): x is the Linear+ReLU output that owns this gradient
                ffi::check(unsafe { ffi::nk_linear_bwd_input_relu(dev, dx.as_mut_ptr(), gz, w.as_ptr(), x.as_ptr(), n, m, o, 0) });
            } else {
                ffi::check(unsafe { ffi::nk_mm_t_bwd_left(dev, dx.as_mut_ptr(), gz, w.as_ptr(), n, m, o) });
            }
        }
        {
            let mut db = self.bias_gradient.borrow_mut();
            let (db_shape, g_shape) = ([o], [n, o]);
            ffi::check(unsafe { ffi::nk_unbroadcast_add(dev, db.as_mut_ptr(), db_shape.as_ptr(), 1, gz, g_shape.as_ptr(), 2) });
        }
        let mut dw = self.weight_gradient.borrow_mut();
        ffi::check(unsafe { ffi::nk_mm_t_bwd_right(dev, dw.as_mut_ptr(), gz, x.as_ptr(), n, m, o) });
    }

    fn targets(&self) -> Vec<usize> {
        let mut t = vec![grad_id(&self.weight_gradient), grad_id(&self.bias_gradient)];
        if let Some(input_gradient) = &self.input_gradient {
            t.push(grad_id(input_gradient));
        }
        t
    }

    fn premask_targets(&self) -> Vec<usize> {
        match (&self.input_mask, &self.input_gradient) {
            (Some(_), Some(input_gradient)) => vec![grad_id(input_gradient)],
            _ => Vec::new(),
        }
    }

    fn masked_gradient(&self) -> Option<(usize, Rc<ReluMask>)> {
        self.output_mask.as_ref().map(|mask| (grad_id(&self.gradient), mask.clone()))
    }
}
