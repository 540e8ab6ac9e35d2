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
}
