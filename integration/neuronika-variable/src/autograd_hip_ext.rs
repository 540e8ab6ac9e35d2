// To be merged into `neuronika-variable/src/autograd.rs` (`autograd.rs:17-25`): ONE defaulted method on the crate-private
// `Backward` trait.  Every existing ndarray node keeps compiling unchanged (the default names no gradient); the device
// nodes (`hip/node/*.rs`) override it.  It is what lets `HipVarDiff::backward_sync` (`hip/hipvar.rs`) tell, while the tape
// is being issued, which node is the LAST one that accumulates into a registered parameter gradient - the moment that
// gradient's all-reduce may start on the side stream (`hip/dp.rs`), overlapped with the rest of the backward pass.
//
// The tested twin of this rule is `Backward::targets` / `VarDiff::run_backward` in this repository's C++ tape
// (`host/neuronika.hpp`, `host/neuronika.cpp`).
pub(crate) trait Backward {
    /// Propagates the computations backwards.
    fn backward(&self);

    /// Identities (`hip::node::grad_id`: the address of the `Rc<Gradient<..>>` allocation) of the gradient buffers this node
    /// accumulates into.  Default: none (nodes of the CPU backend are never asked).
    fn targets(&self) -> Vec<usize> {
        Vec::new()
    }

    /// The subset of `targets` this node can write PRE-MASKED when the gradient belongs to a fused Linear+ReLU node (a
    /// following Linear's input gradient: `nk_linear_bwd_input_relu` applies `(y > 0) *` in the GEMM epilogue).  Default: none.
    fn premask_targets(&self) -> Vec<usize> {
        Vec::new()
    }

    /// For the backward node of a fused Linear+ReLU (`hip::node::LinearBackward` with a mask): the identity of its OWN output
    /// gradient and the state `HipVarDiff::backward` sets per pass - pre-masked iff every node that writes that gradient on
    /// the tape can mask while storing.  Default: not such a node.
    fn masked_gradient(&self) -> Option<(usize, std::rc::Rc<crate::hip::ReluMask>)> {
        None
    }
}
