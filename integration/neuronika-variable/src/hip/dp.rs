//! Data-parallel gradient exchange (net-new: the reference has no communication backend).  One process per GPU; after
//! `backward(1 / world)` every registered parameter gradient is summed over the ranks by RCCL on the device's side
//! stream, overlapped with the rest of the backward pass, and joined before `Optimizer::step`
//! (`neuronika-optim/src/optimizer.rs:81-86`).  Tested form of the same logic: `dp::GradientSync` in
//! `host/neuronika.{hpp,cpp}` of this repository.
use std::rc::Rc;

use super::{device::Device, ffi};

/// One rank of the RCCL communicator.  The 128-byte id is created on rank 0 (`unique_id`) and handed to the other
/// ranks by the launcher's control plane.
pub struct Communicator {
    raw: *mut ffi::nk_comm,
    device: Device,
}

impl Communicator {
    pub fn unique_id() -> [u8; ffi::NK_COMM_ID_BYTES as usize] {
        let mut id = [0u8; ffi::NK_COMM_ID_BYTES as usize];
        ffi::check(unsafe { ffi::nk_comm_unique_id(id.as_mut_ptr() as *mut _) });
        id
    }

    pub fn new(device: Device, nranks: usize, rank: usize, id: &[u8; ffi::NK_COMM_ID_BYTES as usize]) -> Rc<Self> {
        let mut raw = std::ptr::null_mut();
        ffi::check(unsafe { ffi::nk_comm_init_rank(device.as_raw(), nranks as i32, rank as i32, id.as_ptr() as *const _, &mut raw) });
        Rc::new(Self { raw, device })
    }

    pub fn size(&self) -> usize {
        unsafe { ffi::nk_comm_size(self.raw) as usize }
    }
}

impl Drop for Communicator {
    fn drop(&mut self) {
        unsafe { ffi::nk_comm_destroy(self.raw) };
    }
}

/// What `GradientSync` needs to know about one parameter: the identity of its gradient on the tape (`node::grad_id`, what
/// `Backward::targets` reports), the device buffer and its length.  Built by `HipVarDiff::sync_entry`.
#[derive(Clone, Copy)]
pub struct SyncEntry {
    pub(crate) id: usize,
    pub(crate) ptr: *mut f32,
    pub(crate) len: usize,
}

/// A registered gradient buffer, kept alive by the parameter that owns it.
struct Bucket {
    id: usize,
    ptr: *mut f32,
    len: usize,
    event: *mut ffi::nk_event,
}

/// Exchange of the registered parameter gradients.  Two ways to drive it:
/// * `HipVarDiff::backward_sync(seed, &mut sync)` - the overlapped form: `grad_ready(i)` is called right after the LAST tape
///   node that accumulates into parameter `i` has been issued (reverse layer order), so the all-reduce of that gradient runs
///   on the side stream underneath the remaining backward kernels.  "Last writer" is decided from `Backward::targets`
///   (`autograd_hip_ext.rs`), the rule `VarDiff::run_backward` applies in this repository's C++ tape (`host/neuronika.cpp`);
/// * `all_reduce()` after a plain `HipVarDiff::backward` has ISSUED the whole tape: every gradient is final in stream order;
///   correct, but the exchange is exposed behind the last backward kernel.
/// Small gradients (biases) travel as one RCCL group.  `join()` makes the compute stream wait for the side stream - no host
/// synchronisation.
pub struct GradientSync {
    comm: Rc<Communicator>,
    buckets: Vec<Bucket>,
    small_pending: Vec<usize>,
    small_elems: usize,
}

impl GradientSync {
    pub fn new(comm: Rc<Communicator>, parameters: &[SyncEntry]) -> Self {
        let buckets = parameters
            .iter()
            .map(|entry| {
                let mut event = std::ptr::null_mut();
                ffi::check(unsafe { ffi::nk_event_create(comm.device.as_raw(), &mut event) });
                Bucket { id: entry.id, ptr: entry.ptr, len: entry.len, event }
            })
            .collect();
        Self { comm, buckets, small_pending: Vec::new(), small_elems: 65536 }
    }

    /// Index of the registered parameter whose gradient has this tape identity, if any.
    pub(crate) fn bucket_of(&self, id: usize) -> Option<usize> {
        self.buckets.iter().position(|b| b.id == id)
    }

    pub fn grad_ready(&mut self, i: usize) {
        if self.comm.size() == 1 {
            return;
        }
        if self.buckets[i].len < self.small_elems {
            self.small_pending.push(i);
            let n_small = self.buckets.iter().filter(|b| b.len < self.small_elems).count();
            if self.small_pending.len() == n_small {
                self.flush_small();
            }
            return;
        }
        let b = &self.buckets[i];
        ffi::check(unsafe { ffi::nk_event_record(b.event, 0) }); // everything up to the node that finalised this gradient
        ffi::check(unsafe { ffi::nk_allreduce_sum_async(self.comm.raw, b.ptr, b.len, b.event) });
    }

    /// Non-overlapped form: every registered gradient, in registration order, after backward has been issued.
    pub fn all_reduce(&mut self) {
        for i in 0..self.buckets.len() {
            self.grad_ready(i);
        }
    }

    fn flush_small(&mut self) {
        if self.small_pending.is_empty() {
            return;
        }
        let ptrs: Vec<*mut f32> = self.small_pending.iter().map(|&i| self.buckets[i].ptr).collect();
        let lens: Vec<usize> = self.small_pending.iter().map(|&i| self.buckets[i].len).collect();
        let event = self.buckets[self.small_pending[0]].event;
        ffi::check(unsafe { ffi::nk_event_record(event, 0) });
        ffi::check(unsafe { ffi::nk_allreduce_sum_group_async(self.comm.raw, ptrs.as_ptr(), lens.as_ptr(), ptrs.len() as i32, event) });
        self.small_pending.clear();
    }

    pub fn join(&mut self) {
        if self.comm.size() > 1 {
            self.flush_small();
            ffi::check(unsafe { ffi::nk_comm_join(self.comm.raw) });
        }
    }
}

impl Drop for GradientSync {
    fn drop(&mut self) {
        for b in &self.buckets {
            unsafe { ffi::nk_event_destroy(b.event) };
        }
    }
}
