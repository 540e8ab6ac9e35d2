use super::grad_id;
use std::rc::Rc;

use ndarray::Dimension;

use crate::{
    autograd::{Backward, Forward},
    gradient::Gradient,
    hip::{ffi, hiparray::HipArray},
    utils::Shared,
};

/// `Convolution::forward` (`node/convolution/mod.rs:296-355`, kernels `:85-144`): cross-correlation, no internal padding,
/// `groups` splits input and output channels.  The im2col + sgemm of the reference becomes one implicit-GEMM launch.
pub(crate) struct Convolution<D>
where
    D: Dimension,
{
    input_data: Shared<HipArray<D>>,
    kernel_data: Shared<HipArray<D>>,
    data: Shared<HipArray<D>>,
    stride: Vec<i32>,
    dilation: Vec<i32>,
    groups: i32,
}

impl<D> Convolution<D>
where
    D: Dimension, {
    pub(crate) fn new(input_data: Shared<HipArray<D>>, kernel_data: Shared<HipArray<D>>, data: Shared<HipArray<D>>, stride: Vec<i32>, dilation: Vec<i32>, groups: i32) -> Self {
        Self { input_data, kernel_data, data, stride, dilation, groups }
    }
}

impl<D> Forward for Convolution<D>
where
    D: Dimension,
{
    fn forward(&self) {
        let (x, w) = (self.input_data.borrow(), self.kernel_data.borrow());
        let mut y = self.data.borrow_mut();
        let (xs, ws) = (x.shape_c(), w.shape_c());
        ffi::check(unsafe {
            ffi::nk_conv_fwd(x.device().as_raw(), xs.len() as i32 - 2, x.as_ptr(), xs.as_ptr(), w.as_ptr(), ws.as_ptr(), y.as_mut_ptr(),
                             self.stride.as_ptr(), self.dilation.as_ptr(), self.groups)
        });
    }
}

/// `ConvolutionBackwardInput::backward` (`:390-449`, kernel `:146-189`): `dX +=` gather form of col2im (deterministic).
pub(crate) struct ConvolutionBackwardInput<D>
where
    D: Dimension,
{
    kernel_data: Shared<HipArray<D>>,
    input_gradient: Rc<Gradient<HipArray<D>, D>>,
    gradient: Rc<Gradient<HipArray<D>, D>>,
    stride: Vec<i32>,
    dilation: Vec<i32>,
    groups: i32,
}

impl<D> ConvolutionBackwardInput<D>
where
    D: Dimension, {
    pub(crate) fn new(kernel_data: Shared<HipArray<D>>, input_gradient: Rc<Gradient<HipArray<D>, D>>, gradient: Rc<Gradient<HipArray<D>, D>>, stride: Vec<i32>, dilation: Vec<i32>, groups: i32) -> Self {
        Self { kernel_data, input_gradient, gradient, stride, dilation, groups }
    }
}

impl<D> Backward for ConvolutionBackwardInput<D>
where
    D: Dimension,
{
    fn backward(&self) {
        let (g, w) = (self.gradient.borrow(), self.kernel_data.borrow());
        let mut dx = self.input_gradient.borrow_mut();
        let (xs, ws) = (dx.shape_c(), w.shape_c());
        ffi::check(unsafe {
            ffi::nk_conv_bwd_input(g.device().as_raw(), xs.len() as i32 - 2, dx.as_mut_ptr(), xs.as_ptr(), g.as_ptr(), w.as_ptr(), ws.as_ptr(),
                                   self.stride.as_ptr(), self.dilation.as_ptr(), self.groups)
        });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.input_gradient)]
    }
}

/// `ConvolutionBackwardKernel::backward` (`:451-510`, kernel `:191-226`): `dW +=`, a reduction over (sample, position).
pub(crate) struct ConvolutionBackwardKernel<D>
where
    D: Dimension,
{
    input_data: Shared<HipArray<D>>,
    kernel_gradient: Rc<Gradient<HipArray<D>, D>>,
    gradient: Rc<Gradient<HipArray<D>, D>>,
    stride: Vec<i32>,
    dilation: Vec<i32>,
    groups: i32,
}

impl<D> ConvolutionBackwardKernel<D>
where
    D: Dimension, {
    pub(crate) fn new(input_data: Shared<HipArray<D>>, kernel_gradient: Rc<Gradient<HipArray<D>, D>>, gradient: Rc<Gradient<HipArray<D>, D>>, stride: Vec<i32>, dilation: Vec<i32>, groups: i32) -> Self {
        Self { input_data, kernel_gradient, gradient, stride, dilation, groups }
    }
}

impl<D> Backward for ConvolutionBackwardKernel<D>
where
    D: Dimension,
{
    fn backward(&self) {
        let (g, x) = (self.gradient.borrow(), self.input_data.borrow());
        let mut dw = self.kernel_gradient.borrow_mut();
        let (ws, xs) = (dw.shape_c(), x.shape_c());
        ffi::check(unsafe {
            ffi::nk_conv_bwd_kernel(g.device().as_raw(), xs.len() as i32 - 2, dw.as_mut_ptr(), ws.as_ptr(), g.as_ptr(), x.as_ptr(), xs.as_ptr(),
                                    self.stride.as_ptr(), self.dilation.as_ptr(), self.groups)
        });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.kernel_gradient)]
    }
}

/// `nn::Conv{1,2,3}d::forward`'s convolution and bias addition as ONE node: `Convolution::forward` (`:296-355`) with the
/// broadcast `Addition` of the `(Cout, 1, ..)` bias (`node/addition/mod.rs:31-50`) in the kernel's epilogue -
/// `nk_conv_bias_fwd`.  Bit-identical to the two nodes (one f32 add per output on top of the same sums).
pub(crate) struct ConvolutionBias<D, B>
where
    D: Dimension,
    B: Dimension,
{
    input_data: Shared<HipArray<D>>,
    kernel_data: Shared<HipArray<D>>,
    bias_data: Shared<HipArray<B>>,
    data: Shared<HipArray<D>>,
    stride: Vec<i32>,
    dilation: Vec<i32>,
    groups: i32,
}

impl<D, B> ConvolutionBias<D, B>
where
    D: Dimension,
    B: Dimension, {
    #[allow(clippy::too_many_arguments)]
    pub(crate) fn new(input_data: Shared<HipArray<D>>, kernel_data: Shared<HipArray<D>>, bias_data: Shared<HipArray<B>>, data: Shared<HipArray<D>>, stride: Vec<i32>, dilation: Vec<i32>, groups: i32) -> Self {
        Self { input_data, kernel_data, bias_data, data, stride, dilation, groups }
    }
}

impl<D, B> Forward for ConvolutionBias<D, B>
where
    D: Dimension,
    B: Dimension,
{
    fn forward(&self) {
        let (x, w, b) = (self.input_data.borrow(), self.kernel_data.borrow(), self.bias_data.borrow());
        let mut y = self.data.borrow_mut();
        let (xs, ws) = (x.shape_c(), w.shape_c());
        ffi::check(unsafe {
            ffi::nk_conv_bias_fwd(x.device().as_raw(), xs.len() as i32 - 2, x.as_ptr(), xs.as_ptr(), w.as_ptr(), ws.as_ptr(), b.as_ptr(), y.as_mut_ptr(),
                                  self.stride.as_ptr(), self.dilation.as_ptr(), self.groups)
        });
    }
}

/// `ConvolutionBackwardKernel::backward` (`:451-510`) and the `AdditionBackwardRight` of the module's bias
/// (`node/addition/mod.rs:113-135`: the un-broadcast sum of the output gradient over samples and positions) as ONE call:
/// `nk_conv_bwd_kernel_bias` sums the bias gradient on the way through the kernel-gradient pass (same reduction, no second
/// read of the 206 MB gradient at C3).
pub(crate) struct ConvolutionBackwardKernelBias<D, B>
where
    D: Dimension,
    B: Dimension,
{
    input_data: Shared<HipArray<D>>,
    kernel_gradient: Rc<Gradient<HipArray<D>, D>>,
    bias_gradient: Rc<Gradient<HipArray<B>, B>>,
    gradient: Rc<Gradient<HipArray<D>, D>>,
    stride: Vec<i32>,
    dilation: Vec<i32>,
    groups: i32,
}

impl<D, B> ConvolutionBackwardKernelBias<D, B>
where
    D: Dimension,
    B: Dimension, {
    #[allow(clippy::too_many_arguments)]
    pub(crate) fn new(input_data: Shared<HipArray<D>>, kernel_gradient: Rc<Gradient<HipArray<D>, D>>, bias_gradient: Rc<Gradient<HipArray<B>, B>>, gradient: Rc<Gradient<HipArray<D>, D>>, stride: Vec<i32>, dilation: Vec<i32>, groups: i32) -> Self {
        Self { input_data, kernel_gradient, bias_gradient, gradient, stride, dilation, groups }
    }
}

impl<D, B> Backward for ConvolutionBackwardKernelBias<D, B>
where
    D: Dimension,
    B: Dimension,
{
    fn backward(&self) {
        let (g, x) = (self.gradient.borrow(), self.input_data.borrow());
        let (mut dw, mut db) = (self.kernel_gradient.borrow_mut(), self.bias_gradient.borrow_mut());
        let (ws, xs) = (dw.shape_c(), x.shape_c());
        ffi::check(unsafe {
            ffi::nk_conv_bwd_kernel_bias(g.device().as_raw(), xs.len() as i32 - 2, dw.as_mut_ptr(), db.as_mut_ptr(), ws.as_ptr(), g.as_ptr(), x.as_ptr(),
                                         xs.as_ptr(), self.stride.as_ptr(), self.dilation.as_ptr(), self.groups, 0, 0)
        });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.kernel_gradient), grad_id(&self.bias_gradient)]
    }
}

/// The `Conv2d` module with its ZERO padding folded in (`nn::Conv*::forward` = pad -> convolution -> + bias, `neuronika-nn/src/lib.rs:
/// 724-812`): `input_data` is the UNPADDED input, no `Pad` node and no padded copy exist (`Pad::forward`, `node/pad/zero/mod.rs:5-31`:
/// 110 MB and a 40 us kernel per step at C3) - `nk_conv_bias_fwd_padded`.  Built by `HipVarDiff::convolution_bias_padded` only after
/// `nk_conv_padding_folds` said that the Winograd kernels take this geometry (they read out-of-range elements as zeros).
pub(crate) struct ConvolutionBiasPadded<D, B>
where
    D: Dimension,
    B: Dimension,
{
    input_data: Shared<HipArray<D>>,
    kernel_data: Shared<HipArray<D>>,
    bias_data: Shared<HipArray<B>>,
    data: Shared<HipArray<D>>,
    padding: Vec<i32>,
    stride: Vec<i32>,
    dilation: Vec<i32>,
    groups: i32,
}

impl<D, B> ConvolutionBiasPadded<D, B>
where
    D: Dimension,
    B: Dimension, {
    #[allow(clippy::too_many_arguments)]
    pub(crate) fn new(input_data: Shared<HipArray<D>>, kernel_data: Shared<HipArray<D>>, bias_data: Shared<HipArray<B>>, data: Shared<HipArray<D>>, padding: Vec<i32>, stride: Vec<i32>, dilation: Vec<i32>, groups: i32) -> Self {
        Self { input_data, kernel_data, bias_data, data, padding, stride, dilation, groups }
    }
}

impl<D, B> Forward for ConvolutionBiasPadded<D, B>
where
    D: Dimension,
    B: Dimension,
{
    fn forward(&self) {
        let (x, w, b) = (self.input_data.borrow(), self.kernel_data.borrow(), self.bias_data.borrow());
        let mut y = self.data.borrow_mut();
        let (xs, ws) = (x.shape_c(), w.shape_c());
        ffi::check(unsafe {
            ffi::nk_conv_bias_fwd_padded(x.device().as_raw(), xs.len() as i32 - 2, x.as_ptr(), xs.as_ptr(), self.padding.as_ptr(), w.as_ptr(), ws.as_ptr(),
                                         b.as_ptr(), y.as_mut_ptr(), self.stride.as_ptr(), self.dilation.as_ptr(), self.groups)
        });
    }
}

/// Its backward entry: `ConvolutionBackwardInput` + `PadBackward` as one kernel (`nk_conv_bwd_input_padded`: the gradient lands in the
/// UNPADDED input's buffer), then kernel and bias gradient against the unpadded input (`nk_conv_bwd_kernel_bias_padded`).
pub(crate) struct ConvolutionBackwardPadded<D, B>
where
    D: Dimension,
    B: Dimension,
{
    input_data: Shared<HipArray<D>>,
    kernel_data: Shared<HipArray<D>>,
    input_gradient: Rc<Gradient<HipArray<D>, D>>,
    kernel_gradient: Rc<Gradient<HipArray<D>, D>>,
    bias_gradient: Rc<Gradient<HipArray<B>, B>>,
    gradient: Rc<Gradient<HipArray<D>, D>>,
    padding: Vec<i32>,
    stride: Vec<i32>,
    dilation: Vec<i32>,
    groups: i32,
}

impl<D, B> ConvolutionBackwardPadded<D, B>
where
    D: Dimension,
    B: Dimension, {
    #[allow(clippy::too_many_arguments)]
    pub(crate) fn new(input_data: Shared<HipArray<D>>, kernel_data: Shared<HipArray<D>>, input_gradient: Rc<Gradient<HipArray<D>, D>>, kernel_gradient: Rc<Gradient<HipArray<D>, D>>, bias_gradient: Rc<Gradient<HipArray<B>, B>>, gradient: Rc<Gradient<HipArray<D>, D>>, padding: Vec<i32>, stride: Vec<i32>, dilation: Vec<i32>, groups: i32) -> Self {
        Self { input_data, kernel_data, input_gradient, kernel_gradient, bias_gradient, gradient, padding, stride, dilation, groups }
    }
}

impl<D, B> Backward for ConvolutionBackwardPadded<D, B>
where
    D: Dimension,
    B: Dimension,
{
    fn backward(&self) {
        let (g, x, w) = (self.gradient.borrow(), self.input_data.borrow(), self.kernel_data.borrow());
        let (xs, ws) = (x.shape_c(), w.shape_c());
        let dev = g.device().as_raw();
        {
            let mut dx = self.input_gradient.borrow_mut();
            ffi::check(unsafe {
                ffi::nk_conv_bwd_input_padded(dev, xs.len() as i32 - 2, dx.as_mut_ptr(), xs.as_ptr(), self.padding.as_ptr(), g.as_ptr(), w.as_ptr(),
                                              ws.as_ptr(), self.stride.as_ptr(), self.dilation.as_ptr(), self.groups)
            });
        }
        let (mut dw, mut db) = (self.kernel_gradient.borrow_mut(), self.bias_gradient.borrow_mut());
        ffi::check(unsafe {
            ffi::nk_conv_bwd_kernel_bias_padded(dev, xs.len() as i32 - 2, dw.as_mut_ptr(), db.as_mut_ptr(), ws.as_ptr(), g.as_ptr(), x.as_ptr(),
                                                xs.as_ptr(), self.padding.as_ptr(), self.stride.as_ptr(), self.dilation.as_ptr(), self.groups, 0, 0)
        });
    }

    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.input_gradient), grad_id(&self.kernel_gradient), grad_id(&self.bias_gradient)]
    }
}
