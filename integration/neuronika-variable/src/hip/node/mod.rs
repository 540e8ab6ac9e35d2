//! Node bodies: each `forward()` / `backward()` is one call of the C ABI (`include/neuronika_hip.h` cites, per entry
//! point, the reference method it replaces).  The structs hold the same `Shared` operand / output handles as the
//! reference's ndarray nodes (`node/*/mod.rs`), so graph construction code is unchanged.  Written here: the nodes of
//! the BASELINE configurations (MatMul / MatMulT, Convolution, broadcast binaries, ReLU, Softmax, Dropout, Sum,
//! SquaredError, the fused attention core of the composed MHA) and their glue (LogSoftmax, Mean, Pad in all four modes, Chunk,
//! MultiConcatenate, Transpose, the SGD / Adam steps).  Every node is constructed by a `HipVar` / `HipVarDiff` method (`hipvar.rs`).
mod attention;
mod binary_op;
mod convolution;
mod layout;
mod linear;
mod matrix_matrix_mul;
mod matrix_matrix_mul_t;
mod optim;
mod pointwise;
mod reduction;

pub(crate) use attention::*;
pub(crate) use binary_op::*;
pub(crate) use convolution::*;
pub(crate) use layout::*;
pub(crate) use linear::*;
pub(crate) use matrix_matrix_mul::*;
pub(crate) use matrix_matrix_mul_t::*;
pub(crate) use optim::*;
pub(crate) use pointwise::*;
pub(crate) use reduction::*;

use std::rc::Rc;

use crate::autograd::Backward;

/// Identity of a gradient buffer on the tape: the address of its `Rc` allocation.  `Backward::targets` names the gradients
/// a node accumulates into with it, `HipVarDiff::backward_sync` matches them against the registered parameters.
pub(crate) fn grad_id<T>(gradient: &Rc<T>) -> usize {
    Rc::as_ptr(gradient) as *const () as usize
}

/// Two backward halves registered as one tape entry (`MatrixMatrixMulTBackward`, `node/matrix_matrix_mul_t/mod.rs:107-140`).
pub(crate) struct Pair<L: Backward, R: Backward>(pub(crate) L, pub(crate) R);

impl<L: Backward, R: Backward> Backward for Pair<L, R> {
    fn backward(&self) {
        self.0.backward();
        self.1.backward();
    }

    fn targets(&self) -> Vec<usize> {
        let mut t = self.0.targets();
        t.extend(self.1.targets());
        t
    }

    fn premask_targets(&self) -> Vec<usize> {
        let mut t = self.0.premask_targets();
        t.extend(self.1.premask_targets());
        t
    }
}
