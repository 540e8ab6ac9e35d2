//! Device-resident SGD over `HipVarDiff` parameters: the far side of the gradient exchange.  Shaped like
//! `neuronika-optim`'s `SGD` (`neuronika-optim/src/sgd/mod.rs:20-236`) and its `Optimizer` trait (`optimizer.rs:60-94`):
//! `register` the model's parameters, `step()` after `backward` (after `GradientSync::join` in a data-parallel step),
//! `zero_grad()`.  `step` is ONE kernel launch for all parameters (`nk_sgd_step_multi`) - at C4 six updates of 3 x (64 MB + 16 KB)
//! in 97 us instead of six launches.  The tested twin is `optim::SGD` in `host/neuronika.{hpp,cpp}` of this repository.
use ndarray::{Dimension, IxDyn};

use super::{device::Device, hiparray::HipArray, hipvar::HipVarDiff, node::sgd_step_multi};

/// One registered parameter: raw views of its data and gradient buffers (kept alive by the `HipVarDiff` clone next to them) and
/// its momentum buffer (`SGDParam::buffer`, `sgd/mod.rs:150-184`).
struct Param {
    data: *mut f32,
    grad: *mut f32,
    len: usize,
    velocity: Option<HipArray<IxDyn>>,
    zero: Box<dyn Fn()>,
}

/// `SGD` (`sgd/mod.rs:20-60`): learning rate, optional momentum / dampening / Nesterov (`with_momentum`, `:62-110`) and the
/// L1 / L2 / ElasticNet penalty weights (`penalty.rs:63-79`).
pub struct SGD {
    device: Device,
    params: Vec<Param>,
    pub lr: f32,
    pub momentum: f32,
    pub dampening: f32,
    pub nesterov: bool,
    pub l1: f32,
    pub l2: f32,
}

impl SGD {
    pub fn new(device: &Device, lr: f32, l1: f32, l2: f32) -> Self {
        Self { device: device.clone(), params: Vec::new(), lr, momentum: 0., dampening: 0., nesterov: false, l1, l2 }
    }

    /// `SGD::with_momentum` (`sgd/mod.rs:62-110`).
    pub fn with_momentum(mut self, momentum: f32, dampening: f32, nesterov: bool) -> Self {
        self.momentum = momentum;
        self.dampening = dampening;
        self.nesterov = nesterov;
        self
    }

    /// `Optimizer::register` (`optimizer.rs:64-70`).  A parameter registered twice is kept twice and updated twice per step, one
    /// update after the other - what the reference's loop over its list does (`optimizer.rs:81-86`).
    pub fn register<D: 'static + Dimension>(&mut self, param: &HipVarDiff<D>) {
        let data = param.var.data.borrow_mut().as_mut_ptr();
        let (grad, len) = {
            let mut g = param.grad.borrow_mut();
            (g.as_mut_ptr(), g.len())
        };
        let keep = param.clone();
        self.params.push(Param { data, grad, len, velocity: None, zero: Box::new(move || keep.zero_grad()) });
    }

    /// `Optimizer::step` (`optimizer.rs:81-86`): every parameter's `SGDParam::optimize` (`sgd/mod.rs:186-236`), all distinct
    /// parameters in one launch; a second registration of a parameter waits for the next launch instead of racing in this one.
    /// Momentum buffers are created here, the first time a step runs with `momentum != 0` (the field is public and may be set
    /// after `register`; `sgd/mod.rs:201-207` starts the buffer from zero as well).
    pub fn step(&mut self) {
        if self.momentum != 0. {
            for p in self.params.iter_mut().filter(|p| p.velocity.is_none()) {
                p.velocity = Some(HipArray::zeroed(IxDyn(&[p.len]), self.device.clone()));
            }
        }
        let mut done = vec![false; self.params.len()];
        while done.iter().any(|d| !*d) {
            let mut list: Vec<(*mut f32, *mut f32, *mut f32, usize)> = Vec::new();
            for (k, p) in self.params.iter_mut().enumerate() {
                if done[k] || list.iter().any(|q| q.0 == p.data) {
                    continue;
                }
                done[k] = true;
                list.push((p.data, p.grad, p.velocity.as_mut().map_or(std::ptr::null_mut(), |v| v.as_mut_ptr()), p.len));
            }
            sgd_step_multi(&self.device, &list, self.lr, self.momentum, self.dampening, self.nesterov, self.l1, self.l2);
        }
    }

    /// `Optimizer::zero_grad` (`optimizer.rs:88-94`).
    pub fn zero_grad(&self) {
        self.params.iter().for_each(|p| (p.zero)());
    }
}
