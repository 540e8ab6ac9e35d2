use super::grad_id;
use std::cell::Cell;
use std::rc::Rc;

use ndarray::{Ix2, Ix3};

use crate::{
    autograd::{Backward, Forward},
    gradient::Gradient,
    hip::{ffi, hiparray::HipArray},
    utils::Shared,
};

/// Geometry of the composed multi-head attention (SURVEY.md 8a note): `batch` samples of `seq` rows, `heads` heads of
/// `dh` columns inside the `(batch*seq, heads*dh)` projection layout.
#[derive(Clone, Copy)]
pub(crate) struct Heads {
    pub batch: i32,
    pub seq: i32,
    pub heads: i32,
    pub dh: i32,
}

/// Buffers the forward and backward node share, like `Dropout`'s noise buffer (`var.rs:375-393`).
pub(crate) struct AttentionState {
    // sp = seq rounded up to a multiple of 32: the kernels work on whole 32 x 32 tiles of these tensors
    pub scores: Shared<HipArray<Ix3>>,    // (batch*heads, sp, sp) raw scores
    pub stats: Shared<HipArray<Ix3>>,     // (batch*heads, sp, 2)
    pub mask_bits: Shared<HipArray<Ix3>>, // batch*heads*sp*sp/32 u32 words in an f32 buffer (layout: include/neuronika_hip.h, opaque here)
    pub calls: Cell<u64>,                 // forwards so far: each one draws a fresh Philox range
}

/// One node for `mm_t` -> scalar `multiplication` -> `softmax` -> `dropout` -> `mm` per (sample, head):
/// `nk_attention_fwd`.  Use `ffi::nk_attention_supported(seq, dh, p, 1) != 0` to decide between this node and the five
/// reference nodes on `chunk` tiles.
pub(crate) struct HeadsAttention {
    geometry: Heads,
    queries: Shared<HipArray<Ix2>>,
    keys: Shared<HipArray<Ix2>>,
    values: Shared<HipArray<Ix2>>,
    state: Rc<AttentionState>,
    data: Shared<HipArray<Ix2>>,
    scale: f32,
    p: f64,
    status: Rc<Cell<bool>>,
    seed: u64,
}

impl HeadsAttention {
    #[allow(clippy::too_many_arguments)]
    pub(crate) fn new(geometry: Heads, queries: Shared<HipArray<Ix2>>, keys: Shared<HipArray<Ix2>>, values: Shared<HipArray<Ix2>>,
                      state: Rc<AttentionState>, data: Shared<HipArray<Ix2>>, scale: f32, p: f64, status: Rc<Cell<bool>>,
                      seed: u64) -> Self {
        if !(0. ..=1.).contains(&p) {
            panic!("Wrong probability received: {}.", p);
        }
        Self { geometry, queries, keys, values, state, data, scale, p, status, seed }
    }
}

impl Forward for HeadsAttention {
    fn forward(&self) {
        let (q, k, v) = (self.queries.borrow(), self.keys.borrow(), self.values.borrow());
        let (mut scores, mut stats, mut bits) = (self.state.scores.borrow_mut(), self.state.stats.borrow_mut(), self.state.mask_bits.borrow_mut());
        let mut out = self.data.borrow_mut();
        let h = self.geometry;
        let sp = ((h.seq as u64) + 31) / 32 * 32; // draws are indexed in the padded (batch*heads, sp, sp) tensor
        let elems = (h.batch as u64) * (h.heads as u64) * sp * sp;
        let offset = self.state.calls.get() * ((elems + 7) / 8); // 8 draws per Philox call
        ffi::check(unsafe {
            ffi::nk_attention_fwd(q.device().as_raw(), q.as_ptr(), k.as_ptr(), v.as_ptr(), scores.as_mut_ptr(), stats.as_mut_ptr(),
                                  bits.as_mut_ptr() as *mut u32, out.as_mut_ptr(), h.batch, h.seq, h.heads, h.dh, self.scale, self.p,
                                  self.status.get() as i32, self.seed, offset)
        });
        self.state.calls.set(self.state.calls.get() + 1); // only a forward that was issued consumes its Philox range
    }
}

/// The matching backward entry: ONE call, `nk_attention_bwd` - the fused kernel (dS, Pd, dQ) and, inside the library, the two
/// products that reduce over the queries (`dK_bh += dS_bh^T . Q_bh`, `dV_bh += Pd_bh^T . dO_bh`).  `d_scores` / `dropped` are
/// scratch; nothing on this side reads them.
pub(crate) struct HeadsAttentionBackward {
    geometry: Heads,
    queries: Shared<HipArray<Ix2>>,
    keys: Shared<HipArray<Ix2>>,
    values: Shared<HipArray<Ix2>>,
    output: Shared<HipArray<Ix2>>,
    state: Rc<AttentionState>,
    d_scores: Shared<HipArray<Ix3>>, // scratch written here
    dropped: Shared<HipArray<Ix3>>,  // scratch written here
    queries_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
    keys_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
    values_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
    gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
    scale: f32,
    p: f64,
    status: Rc<Cell<bool>>,
}

impl HeadsAttentionBackward {
    /// Panics if two of the three operand gradients are the same buffer (self-attention built as `x.heads_attention(x, x)`):
    /// `backward` holds the three `borrow_mut`s at once.  Project first (`q = x.mm_t(wq)` ...) as `nn::MultiheadAttention` does.
    #[allow(clippy::too_many_arguments)]
    pub(crate) fn new(geometry: Heads, queries: Shared<HipArray<Ix2>>, keys: Shared<HipArray<Ix2>>, values: Shared<HipArray<Ix2>>,
                      output: Shared<HipArray<Ix2>>, state: Rc<AttentionState>, d_scores: Shared<HipArray<Ix3>>, dropped: Shared<HipArray<Ix3>>,
                      queries_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>, keys_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
                      values_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>, gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>, scale: f32, p: f64,
                      status: Rc<Cell<bool>>) -> Self {
        assert!(!Rc::ptr_eq(&queries_gradient, &keys_gradient) && !Rc::ptr_eq(&queries_gradient, &values_gradient)
                    && !Rc::ptr_eq(&keys_gradient, &values_gradient),
                "heads_attention: queries, keys and values must be three different differentiable variables");
        Self { geometry, queries, keys, values, output, state, d_scores, dropped, queries_gradient, keys_gradient, values_gradient, gradient,
               scale, p, status }
    }
}

impl Backward for HeadsAttentionBackward {
    fn backward(&self) {
        let g = self.gradient.borrow();
        let (q, k, v, o) = (self.queries.borrow(), self.keys.borrow(), self.values.borrow(), self.output.borrow());
        let (scores, stats, bits) = (self.state.scores.borrow(), self.state.stats.borrow(), self.state.mask_bits.borrow());
        let (mut ds, mut pd) = (self.d_scores.borrow_mut(), self.dropped.borrow_mut());
        let h = self.geometry;
        let dev = g.device().as_raw();
        let (mut dq, mut dk, mut dv) = (self.queries_gradient.borrow_mut(), self.keys_gradient.borrow_mut(), self.values_gradient.borrow_mut());
        ffi::check(unsafe {
            ffi::nk_attention_bwd(dev, dq.as_mut_ptr(), dk.as_mut_ptr(), dv.as_mut_ptr(), ds.as_mut_ptr(), pd.as_mut_ptr(), g.as_ptr(),
                                  o.as_ptr(), scores.as_ptr(), stats.as_ptr(), bits.as_ptr() as *const u32, q.as_ptr(), k.as_ptr(),
                                  v.as_ptr(), h.batch, h.seq, h.heads, h.dh, self.scale, self.p, self.status.get() as i32, 0, 0, 0)
        });
    }

    /// The gradients this node accumulates into (`autograd.rs` extension: the last-writer rule of `backward_sync`).
    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.queries_gradient), grad_id(&self.keys_gradient), grad_id(&self.values_gradient)]
    }
}

/// The same node for `nn::MultiheadAttention`'s PACKED projections: queries, keys and values are the three column blocks of
/// ONE `(batch*seq, 3*heads*dh)` matrix - the output of a single `Linear` over the row-stacked weights `[Wq; Wk; Wv]` (one
/// GEMM with N = 3 d forward, one with K = 3 d for the input gradient, instead of three each): `nk_attention_qkv_fwd`.
pub(crate) struct PackedHeadsAttention {
    geometry: Heads,
    packed: Shared<HipArray<Ix2>>,
    state: Rc<AttentionState>,
    data: Shared<HipArray<Ix2>>,
    scale: f32,
    p: f64,
    status: Rc<Cell<bool>>,
    seed: u64,
}

impl PackedHeadsAttention {
    #[allow(clippy::too_many_arguments)]
    pub(crate) fn new(geometry: Heads, packed: Shared<HipArray<Ix2>>, state: Rc<AttentionState>, data: Shared<HipArray<Ix2>>, scale: f32,
                      p: f64, status: Rc<Cell<bool>>, seed: u64) -> Self {
        if !(0. ..=1.).contains(&p) {
            panic!("Wrong probability received: {}.", p);
        }
        Self { geometry, packed, state, data, scale, p, status, seed }
    }
}

impl Forward for PackedHeadsAttention {
    fn forward(&self) {
        let qkv = self.packed.borrow();
        let (mut scores, mut stats, mut bits) = (self.state.scores.borrow_mut(), self.state.stats.borrow_mut(), self.state.mask_bits.borrow_mut());
        let mut out = self.data.borrow_mut();
        let h = self.geometry;
        let sp = ((h.seq as u64) + 31) / 32 * 32;
        let elems = (h.batch as u64) * (h.heads as u64) * sp * sp;
        let offset = self.state.calls.get() * ((elems + 7) / 8);
        ffi::check(unsafe {
            ffi::nk_attention_qkv_fwd(qkv.device().as_raw(), qkv.as_ptr(), scores.as_mut_ptr(), stats.as_mut_ptr(), bits.as_mut_ptr() as *mut u32,
                                      out.as_mut_ptr(), h.batch, h.seq, h.heads, h.dh, self.scale, self.p, self.status.get() as i32, self.seed,
                                      offset)
        });
        self.state.calls.set(self.state.calls.get() + 1);
    }
}

/// Backward of the packed node: ONE call, `nk_attention_qkv_bwd`, accumulating into the three column blocks of the packed
/// projection's gradient.
pub(crate) struct PackedHeadsAttentionBackward {
    geometry: Heads,
    packed: Shared<HipArray<Ix2>>,
    output: Shared<HipArray<Ix2>>,
    state: Rc<AttentionState>,
    d_scores: Shared<HipArray<Ix3>>, // scratch written here
    dropped: Shared<HipArray<Ix3>>,  // scratch written here
    packed_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
    gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
    scale: f32,
    p: f64,
    status: Rc<Cell<bool>>,
}

impl PackedHeadsAttentionBackward {
    #[allow(clippy::too_many_arguments)]
    pub(crate) fn new(geometry: Heads, packed: Shared<HipArray<Ix2>>, output: Shared<HipArray<Ix2>>, state: Rc<AttentionState>,
                      d_scores: Shared<HipArray<Ix3>>, dropped: Shared<HipArray<Ix3>>, packed_gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>,
                      gradient: Rc<Gradient<HipArray<Ix2>, Ix2>>, scale: f32, p: f64, status: Rc<Cell<bool>>) -> Self {
        Self { geometry, packed, output, state, d_scores, dropped, packed_gradient, gradient, scale, p, status }
    }
}

impl Backward for PackedHeadsAttentionBackward {
    fn backward(&self) {
        let g = self.gradient.borrow();
        let (qkv, o) = (self.packed.borrow(), self.output.borrow());
        let (scores, stats, bits) = (self.state.scores.borrow(), self.state.stats.borrow(), self.state.mask_bits.borrow());
        let (mut ds, mut pd) = (self.d_scores.borrow_mut(), self.dropped.borrow_mut());
        let h = self.geometry;
        let mut dqkv = self.packed_gradient.borrow_mut();
        ffi::check(unsafe {
            ffi::nk_attention_qkv_bwd(g.device().as_raw(), dqkv.as_mut_ptr(), ds.as_mut_ptr(), pd.as_mut_ptr(), g.as_ptr(), o.as_ptr(),
                                      scores.as_ptr(), stats.as_ptr(), bits.as_ptr() as *const u32, qkv.as_ptr(), h.batch, h.seq, h.heads, h.dh,
                                      self.scale, self.p, self.status.get() as i32, 0)
        });
    }

    fn targets(&self) -> Vec<usize> {
        vec![grad_id(&self.packed_gradient)]
    }
}
