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

/// A registered gradient buffer: raw pointer + length, kept alive by the parameter that owns it.
struct Bucket {
    ptr: *mut f32,
    len: usize,
    event: *mut ffi::nk_event,
}

/// Exchange of the registered parameter gradients.  Two ways to drive it:
/// * `all_reduce()` after `HipVarDiff::backward` has ISSUED the tape: every gradient is final in stream order, the side
///   stream waits for an event recorded behind the last backward kernel, small gradients travel as one RCCL group;
/// * `grad_ready(i)` right after issuing the LAST tape node that accumulates into parameter `i` (reverse layer order) -
///   the overlapped form.  Deciding "last writer" needs each backward node to name the gradients it writes: the
///   `targets()` extension of `Backward` that this repository's C++ tape carries (`host/neuronika.cpp`: `run_backward`,
///   `BackwardHook`); the reference's trait (`autograd.rs:17-25`) has no such method, so `HipVarDiff::backward` as written
///   does not call it.
/// `join()` makes the compute stream wait for the side stream - no host synchronisation.
pub struct GradientSync {
    comm: Rc<Communicator>,
    buckets: Vec<Bucket>,
    small_pending: Vec<usize>,
    small_elems: usize,
}

impl GradientSync {
    pub fn new(comm: Rc<Communicator>, grads: &[(*mut f32, usize)]) -> Self {
        let buckets = grads
            .iter()
            .map(|&(ptr, len)| {
                let mut event = std::ptr::null_mut();
                ffi::check(unsafe { ffi::nk_event_create(comm.device.as_raw(), &mut event) });
                Bucket { ptr, len, event }
            })
            .collect();
        Self { comm, buckets, small_pending: Vec::new(), small_elems: 65536 }
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
