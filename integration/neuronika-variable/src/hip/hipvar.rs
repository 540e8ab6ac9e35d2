//! `HipVar` / `HipVarDiff`: the device twins of `Var` / `VarDiff` (`var.rs:34-40`, `vardiff.rs:35-42`), shaped like the
//! reference's own accelerator template (`CuVar`, `cuda/cuvar.rs:19-100`).  Every method below builds the same tape entry as
//! its reference counterpart - same operand handles, same `History` merge, outputs and gradients allocated zeroed at
//! graph-build time - with a node whose body is one call of the C ABI (`node/*.rs`).  Shape rules stay on the host
//! (`cobroadcast`, `conv_out_shape`, `check_conv_args`: `utils.rs:46-55,97-125,207-237,427-497`).
use std::{
    cell::{Cell, RefCell},
    rc::Rc,
};

use ndarray::{DimMax, Dimension, IntoDimension, Ix0, Ix1, Ix2, Ix3, RemoveAxis};

use super::{
    device::Device,
    dp::GradientSync,
    hiparray::HipArray,
    node::{
        AttentionState, BinaryOp, BinaryOperation, BinaryOperationBackwardLeft, BinaryOperationBackwardRight, Chunk, ChunkBackward,
        Convolution, ConvolutionBackwardInput, ConvolutionBackwardKernel, ConvolutionBackwardKernelBias, ConvolutionBackwardPadded, ConvolutionBias,
        ConvolutionBiasPadded, Dropout,
        DropoutBackward, Heads, HeadsAttention, HeadsAttentionBackward, Linear, LinearBackward, LogSoftmax, LogSoftmaxBackward, MatrixMatrixMul, MatrixMatrixMulBackwardLeft,
        MatrixMatrixMulBackwardRight, MatrixMatrixMulT, MatrixMatrixMulTBackwardLeft, MatrixMatrixMulTBackwardRight, Mean, MeanBackward,
        MultiConcatenate, MultiConcatenateBackward, PackedHeadsAttention, PackedHeadsAttentionBackward, Pad, PadBackward, PadMode, Pair, ReLU,
        ReLUBackward, ReluMask, Softmax, SoftmaxBackward,
        SquaredError, SquaredErrorBackward, Sum, SumBackward, Transpose, TransposeBackward,
    },
};
use crate::{
    autograd::{Backward, Forward},
    gradient::{Gradient, NoGrad},
    history::History,
    utils::{check_conv_args, check_groups_args, cobroadcast, conv_out_shape, Broadcast, Shared},
    Reduction,
};

type Fwd = History<(Rc<dyn Forward>, Cell<bool>)>;
type Bwd = History<(Rc<dyn Backward>, Rc<dyn NoGrad>)>;

fn shared<D: Dimension>(dim: D, device: &Device) -> Shared<HipArray<D>> {
    Rc::new(RefCell::new(HipArray::zeroed(dim, device.clone())))
}

/// Seed of the next random node (dropout): the Philox key.  `thread_rng` in the reference (`node/dropout/mod.rs:70`) is
/// non-reproducible by design; a counter-based generator keyed per node keeps masks reproducible and per-rank distinct.
thread_local! {
    static NEXT_SEED: Cell<u64> = Cell::new(0x9E37_79B9_7F4A_7C15);
}

/// Fixes the key of the next random node created on this thread (each node advances it).
pub fn manual_seed(seed: u64) {
    NEXT_SEED.with(|s| s.set(seed));
}

fn next_seed() -> u64 {
    NEXT_SEED.with(|s| {
        let v = s.get();
        s.set(v.wrapping_add(0x9E37_79B9_7F4A_7C15));
        v
    })
}

/// The reference's padding modes (`node/pad/{zero,constant,reflective,replicative}/mod.rs`) as one value type.
#[derive(Clone, Copy, Debug, PartialEq)]
pub enum PaddingMode {
    Zero,
    Constant(f32),
    Reflective,
    Replicative,
}

impl From<PaddingMode> for PadMode {
    fn from(mode: PaddingMode) -> Self {
        match mode {
            PaddingMode::Zero => PadMode::Constant(0.),
            PaddingMode::Constant(value) => PadMode::Constant(value),
            PaddingMode::Reflective => PadMode::Reflective,
            PaddingMode::Replicative => PadMode::Replicative,
        }
    }
}

/// A non-differentiable variable with data in HBM.  Same fields and tape as `Var<D>` (`var.rs:34-40`) /
/// `CuVar<D>` (`cuda/cuvar.rs:19-46`): only the array type differs.
pub struct HipVar<D>
where
    D: Dimension,
{
    pub(crate) data: Shared<HipArray<D>>,
    pub(crate) history: Fwd,
}

impl<D: Dimension> Clone for HipVar<D> {
    fn clone(&self) -> Self {
        Self { data: self.data.clone(), history: self.history.clone() }
    }
}

impl<D> HipVar<D>
where
    D: 'static + Dimension,
{
    pub(crate) fn leaf(array: HipArray<D>) -> Self {
        Self { data: Rc::new(RefCell::new(array)), history: History::default() }
    }

    /// Uploads a host array (`CuVar::from_ndarray`-style entry; `neuronika::from_ndarray`, `lib.rs`).
    pub fn from_ndarray(array: &ndarray::Array<f32, D>, device: Device) -> Self {
        Self::leaf(HipArray::from_ndarray(array, device))
    }

    pub(crate) fn node(data: Shared<HipArray<D>>, op: Rc<dyn Forward>, mut history: Fwd) -> Self {
        history.insert(Rc::as_ptr(&op) as *const () as usize, (op, Cell::default()));
        Self { data, history }
    }

    fn device(&self) -> Device {
        self.data.borrow().device().clone()
    }

    /// Promotes to a differentiable leaf (`Var::requires_grad`, `var.rs:138-148`).
    pub fn requires_grad(self) -> HipVarDiff<D> {
        let (dim, device) = (self.data.borrow().dimension(), self.device());
        HipVarDiff { var: self, grad: Rc::new(Gradient::hip_zeros(dim, device)), history: History::default(), relu_mask: None }
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

    /// Extents of the data, without touching the device.
    pub fn shape(&self) -> Vec<usize> {
        self.data.borrow().dimension().slice().to_vec()
    }

    /// Host copy of the data (synchronises), `Var::data` (`var.rs:131-136`).
    pub fn data(&self) -> ndarray::Array<f32, D> {
        self.data.borrow().as_ndarray()
    }

    /// Broadcast binary (`Addition` ... `Division`): shape rule `cobroadcast` (`utils.rs:97-125`) stays on the host.
    pub(crate) fn binary<E>(mut self, op: BinaryOp, rhs: HipVar<E>) -> HipVar<Broadcast<D, E>>
    where
        D: DimMax<E>,
        E: 'static + Dimension,
    {
        self.history.merge(rhs.history);
        let dim = cobroadcast(self.data.borrow().dimension(), rhs.data.borrow().dimension());
        let data = shared(dim, &self.device());
        let node = Rc::new(BinaryOperation::new(op, self.data, rhs.data, data.clone()));
        HipVar::node(data, node, self.history)
    }

    /// `Var::sum` (`var.rs:201-207`).
    pub fn sum(self) -> HipVar<Ix0> {
        let data = shared(ndarray::Dim(()), &self.device());
        let op = Sum::new(self.data, data.clone());
        HipVar::node(data, Rc::new(op), self.history)
    }

    /// `Var::mean` (`var.rs:209-214`).
    pub fn mean(self) -> HipVar<Ix0> {
        let data = shared(ndarray::Dim(()), &self.device());
        let op = Mean::new(self.data, data.clone());
        HipVar::node(data, Rc::new(op), self.history)
    }

    /// `Var::relu` (`var.rs:243-249`).
    pub fn relu(self) -> HipVar<D> {
        let data = shared(self.data.borrow().dimension(), &self.device());
        let op = ReLU::new(self.data, data.clone());
        HipVar::node(data, Rc::new(op), self.history)
    }

    /// `Var::softmax` (`var.rs:318-330`).
    pub fn softmax(self, axis: usize) -> HipVar<D> {
        let data = shared(self.data.borrow().dimension(), &self.device());
        let op = Softmax::new(self.data, data.clone(), axis);
        HipVar::node(data, Rc::new(op), self.history)
    }

    /// `Var::log_softmax` (`var.rs:332-344`).
    pub fn log_softmax(self, axis: usize) -> HipVar<D> {
        let data = shared(self.data.borrow().dimension(), &self.device());
        let op = LogSoftmax::new(self.data, data.clone(), axis);
        HipVar::node(data, Rc::new(op), self.history)
    }

    /// `Var::t` (`var.rs:346-352`): dimensions reversed.
    pub fn t(self) -> HipVar<D> {
        let mut dim = self.data.borrow().dimension();
        dim.slice_mut().reverse();
        let data = shared(dim, &self.device());
        let op = Transpose::new(self.data, data.clone());
        HipVar::node(data, Rc::new(op), self.history)
    }

    /// `Var::dropout` (`var.rs:375-393`): the noise buffer is shared with the backward node when there is one.
    pub fn dropout(self, p: f64, status: Rc<Cell<bool>>) -> HipVar<D> {
        let noise = shared(self.data.borrow().dimension(), &self.device());
        self.dropout_with_noise(p, noise, status)
    }

    pub(crate) fn dropout_with_noise(self, p: f64, noise: Shared<HipArray<D>>, status: Rc<Cell<bool>>) -> HipVar<D> {
        let data = shared(self.data.borrow().dimension(), &self.device());
        let op = Dropout::new(self.data, data.clone(), p, noise, status, next_seed());
        HipVar::node(data, Rc::new(op), self.history)
    }

    /// `Var::chunks` (`var.rs:401-417`): `exact_chunks(chunk_size)` in row-major chunk order, remainders skipped.
    pub fn chunks<E>(self, chunk_size: E) -> Vec<HipVar<D>>
    where
        E: IntoDimension<Dim = D>,
    {
        let chunk = chunk_size.into_dimension();
        let count: usize = self.data.borrow().dimension().slice().iter().zip(chunk.slice()).map(|(n, c)| n / c).product();
        (0..count)
            .map(|i| {
                let data = shared(chunk.clone(), &self.device());
                let op = Chunk::new(self.data.clone(), data.clone(), i);
                HipVar::node(data, Rc::new(op), self.history.clone())
            })
            .collect()
    }

    /// `Var::cat` (`var.rs:564-584`): `self` followed by `variables` along `axis`.
    pub fn cat(mut self, variables: &[Self], axis: usize) -> HipVar<D> {
        let mut dim = self.data.borrow().dimension();
        let mut operands_data = vec![self.data.clone()];
        variables.iter().cloned().for_each(|variable| {
            dim.slice_mut()[axis] += variable.data.borrow().dimension().slice()[axis];
            self.history.merge(variable.history);
            operands_data.push(variable.data);
        });
        let data = shared(dim, &self.device());
        let op = MultiConcatenate::new(operands_data, data.clone(), axis);
        HipVar::node(data, Rc::new(op), self.history)
    }

    /// `Var::mse` (`var.rs:454-459`).
    pub fn mse(mut self, target: HipVar<D>, reduction: Reduction) -> HipVar<Ix0> {
        self.history.merge(target.history);
        let data = shared(ndarray::Dim(()), &self.device());
        let op = SquaredError::new(self.data, target.data, data.clone(), reduction);
        HipVar::node(data, Rc::new(op), self.history)
    }

    /// `Var::pad` (`var.rs:726-744`) for the four modes of `node/pad/`: `padding[i]` on both sides of spatial axis `i`.
    pub(crate) fn pad_with(self, padding: &[usize], mode: PadMode) -> HipVar<D> {
        let mut dim = self.data.borrow().dimension();
        dim.slice_mut().iter_mut().skip(2).zip(padding).for_each(|(n, p)| *n += 2 * p);
        let data = shared(dim, &self.device());
        let op = Pad::new(self.data, data.clone(), mode, padding.iter().map(|&p| p as i32).collect());
        HipVar::node(data, Rc::new(op), self.history)
    }

    /// Zero padding (`PaddingMode` `Zero`, `node/pad/zero/mod.rs`).
    pub fn pad_zero(self, padding: &[usize]) -> HipVar<D> {
        self.pad_with(padding, PadMode::Constant(0.))
    }
}

impl<D> HipVar<D>
where
    D: 'static + Dimension + RemoveAxis,
{
    /// `Convolution::convolution` for `Var` kernels (`var.rs:1296-1371`): `self` is the KERNEL, as in the reference; checks
    /// and the output shape come from the host-side helpers unchanged.
    pub fn convolution(mut self, input: HipVar<D>, stride: &[usize], dilation: &[usize], groups: usize) -> HipVar<D> {
        self.history.merge(input.history);
        let shape: D = {
            let (x, w) = (input.data.borrow(), self.data.borrow());
            let (xs, ws): (Vec<usize>, Vec<usize>) = (x.dimension().slice().to_vec(), w.dimension().slice().to_vec());
            check_conv_args(&xs, &ws, stride, dilation);
            check_groups_args(&xs, &ws, groups);
            conv_out_shape(&xs, &ws, stride, dilation)
        };
        let data = shared(shape, &self.device());
        let to_i32 = |v: &[usize]| v.iter().map(|&s| s as i32).collect::<Vec<_>>();
        let op = Convolution::new(input.data, self.data, data.clone(), to_i32(stride), to_i32(dilation), groups as i32);
        HipVar::node(data, Rc::new(op), self.history)
    }
}

impl<D> HipVar<D>
where
    D: 'static + Dimension + RemoveAxis,
{
    /// `convolution` followed by the broadcast `+ bias` of the `nn::Conv*` layers as ONE forward node (`nk_conv_bias_fwd`: the
    /// bias in the epilogue that writes the output).  `bias` has the layer's shape `(out_channels, 1, ..)`.
    pub fn convolution_bias<B>(mut self, input: HipVar<D>, bias: HipVar<B>, stride: &[usize], dilation: &[usize], groups: usize) -> HipVar<D>
    where
        B: 'static + Dimension,
    {
        self.history.merge(input.history);
        self.history.merge(bias.history);
        let shape: D = {
            let (x, w) = (input.data.borrow(), self.data.borrow());
            let (xs, ws): (Vec<usize>, Vec<usize>) = (x.dimension().slice().to_vec(), w.dimension().slice().to_vec());
            check_conv_args(&xs, &ws, stride, dilation);
            check_groups_args(&xs, &ws, groups);
            conv_out_shape(&xs, &ws, stride, dilation)
        };
        let data = shared(shape, &self.device());
        let to_i32 = |v: &[usize]| v.iter().map(|&s| s as i32).collect::<Vec<_>>();
        let op = ConvolutionBias::new(input.data, self.data, bias.data, data.clone(), to_i32(stride), to_i32(dilation), groups as i32);
        HipVar::node(data, Rc::new(op), self.history)
    }
}

impl HipVar<Ix2> {
    /// `Var::mm` (`var.rs:1034-1061`).
    pub fn mm(mut self, rhs: HipVar<Ix2>) -> HipVar<Ix2> {
        self.history.merge(rhs.history);
        let (n, o) = (self.data.borrow().dimension()[0], rhs.data.borrow().dimension()[1]);
        let data = shared(ndarray::Dim([n, o]), &self.device());
        let op = MatrixMatrixMul::new(self.data, rhs.data, data.clone());
        HipVar::node(data, Rc::new(op), self.history)
    }

    /// `Var::mm_t` (`var.rs:1065-1094`).
    pub fn mm_t(mut self, rhs: HipVar<Ix2>) -> HipVar<Ix2> {
        self.history.merge(rhs.history);
        let (n, o) = (self.data.borrow().dimension()[0], rhs.data.borrow().dimension()[0]);
        let data = shared(ndarray::Dim([n, o]), &self.device());
        let op = MatrixMatrixMulT::new(self.data, rhs.data, data.clone());
        HipVar::node(data, Rc::new(op), self.history)
    }
}

impl HipVar<Ix2> {
    /// `Var::mm_t(VarDiff)` (`var.rs:1081-1094`): only the right operand is differentiable, so only
    /// `MatrixMatrixMulTBackwardRight` goes on the tape - the input layer of C4, whose input-gradient GEMM the reference never
    /// runs either.
    pub fn mm_t_diff(self, rhs: HipVarDiff<Ix2>) -> HipVarDiff<Ix2> {
        let left_data = self.data.clone();
        let var = self.mm_t(rhs.var);
        let grad = Rc::new(Gradient::hip_zeros(var.data.borrow().dimension(), var.device()));
        let op: Rc<dyn Backward> = Rc::new(MatrixMatrixMulTBackwardRight::new(left_data, rhs.grad.clone(), grad.clone()));
        HipVarDiff::node(var, grad.clone(), (op, grad), rhs.history)
    }
}

impl HipVar<Ix2> {
    /// `nn::Linear::forward` (`neuronika-nn/src/lib.rs:441-447`: `input.mm_t(weight) + bias`) as ONE forward node -
    /// `nk_linear_fwd`, the bias in the GEMM epilogue - and, with `relu`, the `.relu()` that follows it in the reference's
    /// words (`vardiff.rs:282-288`) in the same epilogue (`nk_linear_relu_fwd`).  Bit-identical to the separate nodes.
    pub fn linear(mut self, weight: HipVar<Ix2>, bias: HipVar<Ix1>, relu: bool) -> HipVar<Ix2> {
        self.history.merge(weight.history);
        self.history.merge(bias.history);
        let (n, o) = (self.data.borrow().dimension()[0], weight.data.borrow().dimension()[0]);
        let data = shared(ndarray::Dim([n, o]), &self.device());
        let op = Linear::new(self.data, weight.data, bias.data, data.clone(), relu);
        HipVar::node(data, Rc::new(op), self.history)
    }

    /// The same layer over an input WITHOUT gradient (the first layer of C4: its input-gradient GEMM is never issued, as in
    /// `Var::mm_t(VarDiff)`, `var.rs:1081-1094`).
    pub fn linear_diff(self, weight: HipVarDiff<Ix2>, bias: HipVarDiff<Ix1>, relu: bool) -> HipVarDiff<Ix2> {
        let (input_data, weight_data) = (self.data.clone(), weight.var.data.clone());
        let mut history = weight.history;
        history.merge(bias.history);
        let var = self.linear(weight.var, bias.var, relu);
        let grad = Rc::new(Gradient::hip_zeros(var.data.borrow().dimension(), var.device()));
        let mask = if relu { Some(Rc::new(ReluMask::new(var.data.clone()))) } else { None };
        let op: Rc<dyn Backward> = Rc::new(LinearBackward::new(input_data, weight_data, mask.clone(), None, None, weight.grad.clone(),
                                                               bias.grad.clone(), grad.clone()));
        let mut out = HipVarDiff::node(var, grad.clone(), (op, grad), history);
        out.relu_mask = mask;
        out
    }
}

/// A differentiable variable with data and gradient in HBM (`VarDiff<D>`, `vardiff.rs:35-42`).
pub struct HipVarDiff<D>
where
    D: Dimension,
{
    pub(crate) var: HipVar<D>,
    pub(crate) grad: Rc<Gradient<HipArray<D>, D>>,
    pub(crate) history: Bwd,
    /// Set on the output of a fused Linear+ReLU node (`linear(.., relu = true)`): the mask record a following `linear` hands to
    /// its backward node, so that the input gradient can be written pre-masked (`nk_linear_bwd_input_relu`).
    pub(crate) relu_mask: Option<Rc<ReluMask>>,
}

impl<D: Dimension> Clone for HipVarDiff<D> {
    fn clone(&self) -> Self {
        Self { var: self.var.clone(), grad: self.grad.clone(), history: self.history.clone(), relu_mask: self.relu_mask.clone() }
    }
}

impl<D> HipVarDiff<D>
where
    D: 'static + Dimension,
{
    pub(crate) fn node(var: HipVar<D>, grad: Rc<Gradient<HipArray<D>, D>>, op: (Rc<dyn Backward>, Rc<dyn NoGrad>), mut history: Bwd) -> Self {
        history.insert(Rc::as_ptr(&op.0) as *const () as usize, op);
        Self { var, grad, history, relu_mask: None }
    }

    fn new_grad<E: Dimension>(&self, dim: E) -> Rc<Gradient<HipArray<E>, E>> {
        Rc::new(Gradient::hip_zeros(dim, self.var.device()))
    }

    pub fn forward(&self) {
        self.var.forward();
    }

    /// `VarDiff::backward` (`vardiff.rs:125-141`): seed the root gradient, run the tape in reverse.  Launches are
    /// asynchronous.  The data-parallel step uses `backward_sync` below (overlapped exchange); calling
    /// `dp::GradientSync::all_reduce` after this plain `backward` is the serialised fallback.
    pub fn backward(&self, seed: f32) {
        debug_assert_eq!(self.var.history.len(), self.var.history.buffer_len(), "Perhaps you forgot to call .forward()?");
        self.grad.borrow_mut().fill(seed);
        let mut buffer = self.history.buffer_mut();
        if buffer.is_empty() {
            *buffer = self.history.to_vec();
        }
        Self::decide_premasking(&buffer);
        buffer.iter().rev().for_each(|(op, _)| op.backward());
    }

    /// Per pass, for every fused Linear+ReLU node on the tape: its output gradient may be stored PRE-MASKED (`(y > 0) *` applied
    /// in the writer's GEMM epilogue, no ReLU-backward kernel) iff every node that accumulates into that gradient can mask
    /// while storing (`Backward::premask_targets`); one plain writer (a second consumer of the activation) and the node masks
    /// a scratch copy instead.  The rule `VarDiff::run_backward` applies in this repository's C++ tape (`host/neuronika.cpp`).
    fn decide_premasking(buffer: &[(Rc<dyn Backward>, Rc<dyn NoGrad>)]) {
        for (op, _) in buffer.iter() {
            if let Some((id, mask)) = op.masked_gradient() {
                let writers = buffer.iter().filter(|(w, _)| w.targets().contains(&id));
                let mut all_mask = true;
                let mut any = false;
                for (w, _) in writers {
                    any = true;
                    all_mask &= w.premask_targets().contains(&id);
                }
                mask.premasked.set(any && all_mask);
            }
        }
    }

    /// The data-parallel form of `backward` (`vardiff.rs:125-141` with the exchange of `hip/dp.rs` inserted): seeds the root
    /// with `seed` (`1 / world` for mean semantics over the global batch), issues the tape in reverse and hands every
    /// registered parameter gradient to `sync` right after the LAST node that accumulates into it has been issued - its
    /// all-reduce then runs on the device's side stream underneath the remaining backward kernels.  Which node is the last
    /// writer is read off `Backward::targets` (`autograd_hip_ext.rs`), exactly as `VarDiff::run_backward` does in the C++
    /// tape of this repository (`host/neuronika.cpp`).  Call `sync.join()` before `Optimizer::step`.
    pub fn backward_sync(&self, seed: f32, sync: &mut GradientSync) {
        debug_assert_eq!(self.var.history.len(), self.var.history.buffer_len(), "Perhaps you forgot to call .forward()?");
        self.grad.borrow_mut().fill(seed);
        let mut buffer = self.history.buffer_mut();
        if buffer.is_empty() {
            *buffer = self.history.to_vec();
        }
        Self::decide_premasking(&buffer);
        // execution order is the reverse of the tape: the LAST writer of a gradient is the entry with the smallest index
        let mut last_writer: std::collections::HashMap<usize, usize> = std::collections::HashMap::new();
        for (index, (op, _)) in buffer.iter().enumerate() {
            for id in op.targets() {
                last_writer.entry(id).or_insert(index);
            }
        }
        for (index, (op, _)) in buffer.iter().enumerate().rev() {
            op.backward();
            let mut finished = op.targets();
            finished.sort_unstable();
            finished.dedup(); // `x * x` names the same gradient twice: one hand-over
            for id in finished {
                if last_writer[&id] == index {
                    if let Some(bucket) = sync.bucket_of(id) {
                        sync.grad_ready(bucket);
                    }
                }
            }
        }
    }

    /// `VarDiff::zero_grad` (`vardiff.rs:100-102`).
    /// Extents of the data, without touching the device.
    pub fn shape(&self) -> Vec<usize> {
        self.var.shape()
    }

    /// A differentiable leaf holding `array` (parameter construction: `zeros(..).requires_grad()` + `init::uniform` in the
    /// reference, `neuronika-nn/src/lib.rs:425-433`; here the values are drawn on the host and uploaded once).
    pub fn parameter(array: &ndarray::Array<f32, D>, device: Device) -> Self {
        HipVar::from_ndarray(array, device).requires_grad()
    }

    /// Registration record of this parameter for `dp::GradientSync::new`.
    pub fn sync_entry(&self) -> super::dp::SyncEntry {
        let mut grad = self.grad.borrow_mut();
        super::dp::SyncEntry { id: super::node::grad_id(&self.grad), ptr: grad.as_mut_ptr(), len: grad.len() }
    }

    pub fn zero_grad(&self) {
        self.grad.borrow_mut().fill(0.);
    }

    /// Host copy of the gradient (synchronises), `VarDiff::grad` (`vardiff.rs:144-150`).
    pub fn grad(&self) -> ndarray::Array<f32, D> {
        self.grad.borrow().as_ndarray()
    }

    /// `VarDiff::sum` (`vardiff.rs:238-245`).
    pub fn sum(self) -> HipVarDiff<Ix0> {
        let grad = self.new_grad(ndarray::Dim(()));
        let op = SumBackward::new(self.grad.clone(), grad.clone());
        let var = self.var.sum();
        HipVarDiff::node(var, grad.clone(), (Rc::new(op), grad), self.history)
    }

    /// `VarDiff::mean` (`vardiff.rs:247-253`).
    pub fn mean(self) -> HipVarDiff<Ix0> {
        let grad = self.new_grad(ndarray::Dim(()));
        let op = MeanBackward::new(self.grad.clone(), grad.clone());
        let var = self.var.mean();
        HipVarDiff::node(var, grad.clone(), (Rc::new(op), grad), self.history)
    }

    /// `VarDiff::relu` (`vardiff.rs:282-288`).
    pub fn relu(self) -> HipVarDiff<D> {
        let grad = self.new_grad(self.grad.shape());
        let op = ReLUBackward::new(self.grad.clone(), self.var.data.clone(), grad.clone());
        let var = self.var.relu();
        HipVarDiff::node(var, grad.clone(), (Rc::new(op), grad), self.history)
    }

    /// `VarDiff::softmax` (`vardiff.rs:359-365`).
    pub fn softmax(self, axis: usize) -> HipVarDiff<D> {
        let grad = self.new_grad(self.grad.shape());
        let var = self.var.softmax(axis);
        let op = SoftmaxBackward::new(self.grad.clone(), var.data.clone(), grad.clone(), axis);
        HipVarDiff::node(var, grad.clone(), (Rc::new(op), grad), self.history)
    }

    /// `VarDiff::log_softmax` (`vardiff.rs:381-387`).
    pub fn log_softmax(self, axis: usize) -> HipVarDiff<D> {
        let grad = self.new_grad(self.grad.shape());
        let var = self.var.log_softmax(axis);
        let op = LogSoftmaxBackward::new(self.grad.clone(), var.data.clone(), grad.clone(), axis);
        HipVarDiff::node(var, grad.clone(), (Rc::new(op), grad), self.history)
    }

    /// `VarDiff::t` (`vardiff.rs:390-396`).
    pub fn t(self) -> HipVarDiff<D> {
        let var = self.var.t();
        let grad = Rc::new(Gradient::hip_zeros(var.data.borrow().dimension(), var.device()));
        let op = TransposeBackward::new(self.grad.clone(), grad.clone());
        HipVarDiff::node(var, grad.clone(), (Rc::new(op), grad), self.history)
    }

    /// `VarDiff::dropout` (`vardiff.rs:418-427`): forward and backward node share the noise buffer; the backward node does
    /// NOT divide by `1 - p` (`node/dropout/mod.rs:123-126`, kept).
    pub fn dropout(self, p: f64, status: Rc<Cell<bool>>) -> HipVarDiff<D> {
        let grad = self.new_grad(self.grad.shape());
        let noise = shared(self.grad.shape(), &self.var.device());
        let var = self.var.dropout_with_noise(p, noise.clone(), status.clone());
        let op = DropoutBackward::new(self.grad.clone(), grad.clone(), p, noise, status);
        HipVarDiff::node(var, grad.clone(), (Rc::new(op), grad), self.history)
    }

    /// `VarDiff::chunks` (`vardiff.rs:435-452`).
    pub fn chunks<E>(self, chunk_size: E) -> Vec<HipVarDiff<D>>
    where
        E: IntoDimension<Dim = D>,
    {
        self.var
            .chunks(chunk_size)
            .into_iter()
            .enumerate()
            .map(|(i, var)| {
                let grad = Rc::new(Gradient::hip_zeros(var.data.borrow().dimension(), var.device()));
                let op = ChunkBackward::new(self.grad.clone(), grad.clone(), i);
                HipVarDiff::node(var, grad.clone(), (Rc::new(op), grad), self.history.clone())
            })
            .collect()
    }

    /// `VarDiff::cat` (`vardiff.rs:627-650`).
    pub fn cat(mut self, vars: &[Self], axis: usize) -> HipVarDiff<D> {
        let mut operands_gradients = vec![self.grad.clone()];
        let mut operands = Vec::with_capacity(vars.len());
        vars.iter().cloned().for_each(|v| {
            self.history.merge(v.history);
            operands_gradients.push(v.grad);
            operands.push(v.var);
        });
        let var = self.var.cat(&operands, axis);
        let grad = Rc::new(Gradient::hip_zeros(var.data.borrow().dimension(), var.device()));
        let op = MultiConcatenateBackward::new(operands_gradients, grad.clone(), axis);
        HipVarDiff::node(var, grad.clone(), (Rc::new(op), grad), self.history)
    }

    /// `VarDiff::mse` (`vardiff.rs:495-506`).
    pub fn mse(self, target: HipVar<D>, reduction: Reduction) -> HipVarDiff<Ix0> {
        let grad = self.new_grad(ndarray::Dim(()));
        let op = SquaredErrorBackward::new(self.var.data.clone(), target.data.clone(), self.grad.clone(), grad.clone(), reduction.clone());
        let var = self.var.mse(target, reduction);
        HipVarDiff::node(var, grad.clone(), (Rc::new(op), grad), self.history)
    }

    /// `VarDiff::pad` (`vardiff.rs:746-766`), zero mode; `pad_with` takes the other three modes.
    pub fn pad_zero(self, padding: &[usize]) -> HipVarDiff<D> {
        self.pad_with(padding, PadMode::Constant(0.))
    }

    /// `VarDiff::pad(padding, mode)` with the mode as a value (`PaddingMode`: the reference's four marker types `Zero`,
    /// `Constant`, `Reflective`, `Replicative`, `node/pad/`), which is how the `nn` convolution layers store it.
    pub fn pad(self, padding: &[usize], mode: PaddingMode) -> HipVarDiff<D> {
        self.pad_with(padding, mode.into())
    }

    pub(crate) fn pad_with(self, padding: &[usize], mode: PadMode) -> HipVarDiff<D> {
        let var = self.var.pad_with(padding, mode);
        let grad = Rc::new(Gradient::hip_zeros(var.data.borrow().dimension(), var.device()));
        let op = PadBackward::new(self.grad.clone(), grad.clone(), padding.iter().map(|&p| p as i32).collect());
        HipVarDiff::node(var, grad.clone(), (Rc::new(op), grad), self.history)
    }

    /// Broadcast binary with two differentiable operands (`vardiff.rs:766-862` and the operator impls): forward node +
    /// the two fused un-broadcast backward nodes as one tape entry.
    pub(crate) fn binary<E>(mut self, op: BinaryOp, rhs: HipVarDiff<E>) -> HipVarDiff<Broadcast<D, E>>
    where
        D: DimMax<E>,
        E: 'static + Dimension,
    {
        self.history.merge(rhs.history);
        let (left_data, right_data) = (self.var.data.clone(), rhs.var.data.clone());
        let var = self.var.binary(op, rhs.var);
        let grad = Rc::new(Gradient::hip_zeros(var.data.borrow().dimension(), var.device()));
        let left = BinaryOperationBackwardLeft::new(op, right_data.clone(), self.grad.clone(), grad.clone());
        let right = BinaryOperationBackwardRight::new(op, left_data, right_data, rhs.grad.clone(), grad.clone());
        let node: Rc<dyn Backward> = Rc::new(Pair(left, right));
        HipVarDiff::node(var, grad.clone(), (node, grad), self.history)
    }
}

impl<D> HipVarDiff<D>
where
    D: 'static + Dimension + RemoveAxis,
{
    /// `Convolution::convolution` for a differentiable kernel and input (`vardiff.rs:1357-1431`): `self` is the KERNEL.
    /// One forward node; `ConvolutionBackwardInput` + `ConvolutionBackwardKernel` as one tape entry (`ConvolutionBackward`,
    /// `node/convolution/mod.rs:357-388`).
    pub fn convolution(mut self, input: HipVarDiff<D>, stride: &[usize], dilation: &[usize], groups: usize) -> HipVarDiff<D> {
        self.history.merge(input.history);
        let (input_data, kernel_data) = (input.var.data.clone(), self.var.data.clone());
        let var = self.var.convolution(input.var, stride, dilation, groups);
        let grad = Rc::new(Gradient::hip_zeros(var.data.borrow().dimension(), var.device()));
        let to_i32 = |v: &[usize]| v.iter().map(|&s| s as i32).collect::<Vec<_>>();
        let bwd_input = ConvolutionBackwardInput::new(kernel_data, input.grad.clone(), grad.clone(), to_i32(stride), to_i32(dilation), groups as i32);
        let bwd_kernel = ConvolutionBackwardKernel::new(input_data, self.grad.clone(), grad.clone(), to_i32(stride), to_i32(dilation), groups as i32);
        let node: Rc<dyn Backward> = Rc::new(Pair(bwd_input, bwd_kernel));
        HipVarDiff::node(var, grad.clone(), (node, grad), self.history)
    }
}

impl<D> HipVarDiff<D>
where
    D: 'static + Dimension + RemoveAxis,
{
    /// `convolution(..) + bias` of the `nn::Conv*` layers with kernel, input and bias differentiable: one forward node
    /// (`HipVar::convolution_bias`), and as one tape entry `ConvolutionBackwardInput` + `ConvolutionBackwardKernelBias` (the bias
    /// gradient summed on the way through the kernel-gradient pass - no second read of the output gradient).
    pub fn convolution_bias<B>(mut self, input: HipVarDiff<D>, bias: HipVarDiff<B>, stride: &[usize], dilation: &[usize],
                               groups: usize) -> HipVarDiff<D>
    where
        B: 'static + Dimension,
    {
        self.history.merge(input.history);
        self.history.merge(bias.history);
        let (input_data, kernel_data) = (input.var.data.clone(), self.var.data.clone());
        let var = self.var.convolution_bias(input.var, bias.var, stride, dilation, groups);
        let grad = Rc::new(Gradient::hip_zeros(var.data.borrow().dimension(), var.device()));
        let to_i32 = |v: &[usize]| v.iter().map(|&s| s as i32).collect::<Vec<_>>();
        let bwd_input = ConvolutionBackwardInput::new(kernel_data, input.grad.clone(), grad.clone(), to_i32(stride), to_i32(dilation), groups as i32);
        let bwd_kernel = ConvolutionBackwardKernelBias::new(input_data, self.grad.clone(), bias.grad.clone(), grad.clone(), to_i32(stride),
                                                            to_i32(dilation), groups as i32);
        let node: Rc<dyn Backward> = Rc::new(Pair(bwd_input, bwd_kernel));
        HipVarDiff::node(var, grad.clone(), (node, grad), self.history)
    }
}

impl<D> HipVarDiff<D>
where
    D: 'static + Dimension + RemoveAxis,
{
    /// Does the library run this module geometry with the Zero padding folded in (`nk_conv_padding_folds`: 3 x 3, stride 1, one group,
    /// padding 0 / 1, 64 | channel counts, even output extents, and sizes the rules give to the Winograd kernels)?  `self` is the KERNEL.
    pub fn padding_folds(&self, input: &HipVarDiff<D>, padding: &[usize], stride: &[usize], dilation: &[usize], groups: usize) -> bool {
        let to_i32 = |v: &[usize]| v.iter().map(|&s| s as i32).collect::<Vec<_>>();
        let (xs, ws) = (input.var.data.borrow().shape_c(), self.var.data.borrow().shape_c());
        let mut folds = 0i32;
        super::ffi::check(unsafe {
            super::ffi::nk_conv_padding_folds(self.var.device().as_raw(), xs.len() as i32 - 2, xs.as_ptr(), to_i32(padding).as_ptr(), ws.as_ptr(),
                                              to_i32(stride).as_ptr(), to_i32(dilation).as_ptr(), groups as i32, &mut folds)
        });
        folds != 0
    }

    /// `pad(padding, Zero) -> convolution -> + bias` of the `nn::Conv*` layers as ONE node pair WITHOUT the Pad node: `input` is the
    /// unpadded variable, forward on `nk_conv_bias_fwd_padded`, backward `nk_conv_bwd_input_padded` + `nk_conv_bwd_kernel_bias_padded`.
    /// Call only after `padding_folds` said yes (the entry points refuse other geometries).
    pub fn convolution_bias_padded<B>(mut self, input: HipVarDiff<D>, bias: HipVarDiff<B>, padding: &[usize], stride: &[usize],
                                      dilation: &[usize], groups: usize) -> HipVarDiff<D>
    where
        B: 'static + Dimension,
    {
        self.history.merge(input.history);
        self.history.merge(bias.history);
        let mut fwd_history = self.var.history;
        fwd_history.merge(input.var.history);
        fwd_history.merge(bias.var.history);
        let to_i32 = |v: &[usize]| v.iter().map(|&s| s as i32).collect::<Vec<_>>();
        let device = self.var.data.borrow().device().clone();
        let shape: D = {
            let (x, w) = (input.var.data.borrow(), self.var.data.borrow());
            let mut xs: Vec<usize> = x.dimension().slice().to_vec();
            for (i, p) in padding.iter().enumerate() {
                xs[2 + i] += 2 * p;
            }
            let ws: Vec<usize> = w.dimension().slice().to_vec();
            check_conv_args(&xs, &ws, stride, dilation);
            check_groups_args(&xs, &ws, groups);
            conv_out_shape(&xs, &ws, stride, dilation)
        };
        let data = shared(shape.clone(), &device);
        let fwd = ConvolutionBiasPadded::new(input.var.data.clone(), self.var.data.clone(), bias.var.data.clone(), data.clone(), to_i32(padding),
                                             to_i32(stride), to_i32(dilation), groups as i32);
        let var = HipVar::node(data, Rc::new(fwd), fwd_history);
        let grad = Rc::new(Gradient::hip_zeros(shape, device));
        let bwd = ConvolutionBackwardPadded::new(input.var.data, self.var.data, input.grad, self.grad, bias.grad, grad.clone(), to_i32(padding),
                                                 to_i32(stride), to_i32(dilation), groups as i32);
        let op: Rc<dyn Backward> = Rc::new(bwd);
        HipVarDiff::node(var, grad.clone(), (op, grad), self.history)
    }
}

impl HipVarDiff<Ix2> {
    /// `mm` (`vardiff.rs:1073-1106`).
    pub fn mm(mut self, rhs: HipVarDiff<Ix2>) -> HipVarDiff<Ix2> {
        self.history.merge(rhs.history);
        let (left_data, right_data) = (self.var.data.clone(), rhs.var.data.clone());
        let var = self.var.mm(rhs.var);
        let grad = Rc::new(Gradient::hip_zeros(var.data.borrow().dimension(), var.device()));
        let left = MatrixMatrixMulBackwardLeft::new(right_data, self.grad.clone(), grad.clone());
        let right = MatrixMatrixMulBackwardRight::new(left_data, rhs.grad.clone(), grad.clone());
        let op: Rc<dyn Backward> = Rc::new(MatrixMatrixMulBackward::new(left, right));
        HipVarDiff::node(var, grad.clone(), (op, grad), self.history)
    }

    /// `mm_t` (`vardiff.rs:1110-1143`): the node `nn::Linear::forward` is made of (`neuronika-nn/src/lib.rs:443-446`).
    pub fn mm_t(mut self, rhs: HipVarDiff<Ix2>) -> HipVarDiff<Ix2> {
        self.history.merge(rhs.history);
        let (left_data, right_data) = (self.var.data.clone(), rhs.var.data.clone());
        let var = self.var.mm_t(rhs.var);
        let grad = Rc::new(Gradient::hip_zeros(var.data.borrow().dimension(), var.device()));
        let left = MatrixMatrixMulTBackwardLeft::new(right_data, self.grad.clone(), grad.clone());
        let right = MatrixMatrixMulTBackwardRight::new(left_data, rhs.grad.clone(), grad.clone());
        let op: Rc<dyn Backward> = Rc::new(MatrixMatrixMulTBackward::new(left, right));
        HipVarDiff::node(var, grad.clone(), (op, grad), self.history)
    }

    /// `nn::Linear::forward` over a differentiable input: ONE forward node (`HipVar::linear`) and ONE backward entry
    /// (`LinearBackward`: input gradient, bias gradient, weight gradient).  When `self` is itself the output of a
    /// Linear+ReLU node, its mask record travels into the backward node: the input gradient is then written through
    /// `nk_linear_bwd_input_relu` whenever the pass allows (`decide_premasking`) - the C4 graph runs without a single ReLU kernel.
    pub fn linear(mut self, weight: HipVarDiff<Ix2>, bias: HipVarDiff<Ix1>, relu: bool) -> HipVarDiff<Ix2> {
        self.history.merge(weight.history);
        self.history.merge(bias.history);
        let (input_data, weight_data) = (self.var.data.clone(), weight.var.data.clone());
        let var = self.var.linear(weight.var, bias.var, relu);
        let grad = Rc::new(Gradient::hip_zeros(var.data.borrow().dimension(), var.device()));
        let mask = if relu { Some(Rc::new(ReluMask::new(var.data.clone()))) } else { None };
        let op: Rc<dyn Backward> = Rc::new(LinearBackward::new(input_data, weight_data, mask.clone(), self.relu_mask.clone(),
                                                               Some(self.grad.clone()), weight.grad.clone(), bias.grad.clone(), grad.clone()));
        let mut out = HipVarDiff::node(var, grad.clone(), (op, grad), self.history);
        out.relu_mask = mask;
        out
    }

    /// `heads_attention` for PACKED projections: `self` is the `(batch*seq, 3*heads*dh)` output of one `Linear` over the
    /// row-stacked weights `[Wq; Wk; Wv]`; queries, keys and values are read in place as its three column blocks
    /// (`nk_attention_qkv_fwd` / `nk_attention_qkv_bwd`).  Output `(batch*seq, heads*dh)`.
    #[allow(clippy::too_many_arguments)]
    pub fn packed_heads_attention(self, batch: usize, seq: usize, heads: usize, dh: usize, scale: f32, p: f64,
                                  status: Rc<Cell<bool>>) -> HipVarDiff<Ix2> {
        let device = self.var.data.borrow().device().clone();
        let geometry = Heads { batch: batch as i32, seq: seq as i32, heads: heads as i32, dh: dh as i32 };
        let sp = (seq + 31) / 32 * 32;
        let big = |last: usize| shared(ndarray::Dim([batch * heads, sp, last]), &device);
        let state = Rc::new(AttentionState { scores: big(sp), stats: big(2), mask_bits: big(sp / 32), calls: Cell::new(0) });
        let dim = ndarray::Dim([batch * seq, heads * dh]);
        let data = shared(dim, &device);
        let fwd = PackedHeadsAttention::new(geometry, self.var.data.clone(), state.clone(), data.clone(), scale, p, status.clone(), next_seed());
        let var = HipVar::node(data.clone(), Rc::new(fwd), self.var.history);
        let grad = Rc::new(Gradient::hip_zeros(dim, device));
        let bwd = PackedHeadsAttentionBackward::new(geometry, self.var.data, data, state, big(sp), big(sp), self.grad, grad.clone(), scale, p,
                                                    status);
        let op: Rc<dyn Backward> = Rc::new(bwd);
        HipVarDiff::node(var, grad.clone(), (op, grad), self.history)
    }

    /// The composed multi-head attention's per-(sample, head) chain `mm_t -> * scale -> softmax(1) -> dropout -> mm` as ONE
    /// node on the fused kernels (SURVEY.md 8a note; `self` = queries in the `(batch*seq, heads*dh)` projection layout).
    /// Callers check `ffi::nk_attention_supported` first and compose the five reference nodes on `chunks` otherwise.
    #[allow(clippy::too_many_arguments)]
    pub fn heads_attention(mut self, keys: HipVarDiff<Ix2>, values: HipVarDiff<Ix2>, batch: usize, seq: usize, heads: usize, dh: usize,
                           scale: f32, p: f64, status: Rc<Cell<bool>>) -> HipVarDiff<Ix2> {
        self.history.merge(keys.history);
        self.history.merge(values.history);
        let mut fwd_history = self.var.history;
        fwd_history.merge(keys.var.history);
        fwd_history.merge(values.var.history);
        let device = self.var.data.borrow().device().clone();
        let geometry = Heads { batch: batch as i32, seq: seq as i32, heads: heads as i32, dh: dh as i32 };
        // the scratch tensors are padded to whole 32 x 32 tiles: row count and row stride `sp` (include/neuronika_hip.h)
        let sp = (seq + 31) / 32 * 32;
        let big = |last: usize| shared(ndarray::Dim([batch * heads, sp, last]), &device);
        let state = Rc::new(AttentionState { scores: big(sp), stats: big(2), mask_bits: big(sp / 32), calls: Cell::new(0) });
        let dim = self.var.data.borrow().dimension();
        let data = shared(dim, &device);
        let fwd = HeadsAttention::new(geometry, self.var.data.clone(), keys.var.data.clone(), values.var.data.clone(), state.clone(),
                                      data.clone(), scale, p, status.clone(), next_seed());
        let var = HipVar::node(data.clone(), Rc::new(fwd), fwd_history);
        let grad = Rc::new(Gradient::hip_zeros(dim, device));
        let bwd = HeadsAttentionBackward::new(geometry, self.var.data, keys.var.data, values.var.data, data, state, big(sp), big(sp),
                                              self.grad, keys.grad, values.grad, grad.clone(), scale, p, status);
        let op: Rc<dyn Backward> = Rc::new(bwd);
        HipVarDiff::node(var, grad.clone(), (op, grad), self.history)
    }
}

// Operators: `+ - * /` between device variables (`var.rs:746-838`, `vardiff.rs:766-862` and their `std::ops` impls).
macro_rules! impl_binary {
    ($trait:ident, $fun:ident, $op:expr) => {
        impl<D, E> std::ops::$trait<HipVar<E>> for HipVar<D>
        where
            D: 'static + DimMax<E>,
            E: 'static + Dimension,
        {
            type Output = HipVar<Broadcast<D, E>>;

            fn $fun(self, rhs: HipVar<E>) -> Self::Output {
                self.binary($op, rhs)
            }
        }

        impl<D, E> std::ops::$trait<HipVarDiff<E>> for HipVarDiff<D>
        where
            D: 'static + DimMax<E>,
            E: 'static + Dimension,
        {
            type Output = HipVarDiff<Broadcast<D, E>>;

            fn $fun(self, rhs: HipVarDiff<E>) -> Self::Output {
                self.binary($op, rhs)
            }
        }
    };
}

impl_binary!(Add, add, BinaryOp::Add);
impl_binary!(Sub, sub, BinaryOp::Sub);
impl_binary!(Mul, mul, BinaryOp::Mul);
impl_binary!(Div, div, BinaryOp::Div);

/// The unused-import guard of `Ix3` (attention buffers are `Ix3`, built through `shared`).
#[allow(dead_code)]
type AttentionBuffer = HipArray<Ix3>;
