// Data-parallel gradient exchange: RCCL sum all-reduce on the device's SIDE stream, ordered
// after a compute-stream event, joined back without a host sync.  Net-new: the reference has no
// communication backend; the insertion point is between `VarDiff::backward`
// (neuronika-variable/src/vardiff.rs:125-141) and `Optimizer::step`
// (neuronika-optim/src/optimizer.rs:81-86); the message is the leaf gradient buffers of the
// parameters registered with the optimizer (optimizer.rs:70-77).
#include <rccl/rccl.h>

#include <cstdint>
#include <cstring>

#include "nk_common.h"

struct nk_comm {
    nk_device* dev;
    ncclComm_t comm;  // null for a replica communicator (nk_comm_init_replicas)
    int rank, size;
    // replica communicator only - the overlap PROJECTION of benchmarks/overlap_projection.py: the stand-in "all-reduce"
    // occupies `channels` workgroups (what RCCL's channels occupy next to the GEMMs) and paces its pass over the buffer to
    // `gbps` of algorithm bandwidth (what the fabric would give).  0 / 0: an unthrottled streaming pass (coverage tests).
    int channels = 0;
    double gbps = 0.0;
};

// The sum over `size` ranks that all hold the same values: what a replica communicator "exchanges".
__global__ void __launch_bounds__(256) replica_sum_kernel(float* __restrict__ x, size_t n, float ranks) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] *= ranks;
}

// The paced stand-in: `gridDim.x` workgroups (one per emulated RCCL channel), each owning a contiguous slice, walk the
// buffer in steps of 16 KB per workgroup (4 x 16 B per lane in flight: one workgroup sustains tens of GB/s, as a RCCL channel
// does) and do not start step i before  t0 + i * ticks_per_step  of the constant-rate wall clock (100 MHz, s_memrealtime): the
// launch lasts bytes / gbps - unless `channels` workgroups cannot move that much, which is then what the projection shows -
// and holds `channels` workgroup slots for that long.
__global__ void __launch_bounds__(256) replica_sum_paced_kernel(float* __restrict__ x, size_t n, float ranks, double ticks_per_step) {
    const size_t n4 = n / 4;                                   // whole float4s (the tail is handled by block 0 below)
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    float4* x4 = reinterpret_cast<float4*>(x);
    const unsigned long long t0 = wall_clock64();
    size_t step = 0;
    for (size_t base = lo; base < hi; base += 1024, ++step) {
        const unsigned long long due = t0 + (unsigned long long)(ticks_per_step * (double)step);
        while (wall_clock64() < due) __builtin_amdgcn_s_sleep(4);
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = base + j * 256 + threadIdx.x;
            if (i < hi) v[j] = x4[i];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = base + j * 256 + threadIdx.x;
            if (i < hi) x4[i] = make_float4(v[j].x * ranks, v[j].y * ranks, v[j].z * ranks, v[j].w * ranks);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) x[n4 * 4 + threadIdx.x] *= ranks;
}

static int fail_rccl(ncclResult_t r, const char* what) {
    nk_set_error("RCCL error %d (%s) in `%s`", (int)r, ncclGetErrorString(r), what);
    return NK_ERR_RCCL;
}
#define NK_RCCL(call)                                        \
    do {                                                     \
        ncclResult_t _r = (call);                            \
        if (_r != ncclSuccess) return fail_rccl(_r, #call);  \
    } while (0)

static_assert(sizeof(ncclUniqueId) == NK_COMM_ID_BYTES, "unique id size");

extern "C" {

int nk_comm_unique_id(char id[NK_COMM_ID_BYTES]) {
    NK_CHECK(id != nullptr, "null id");
    ncclUniqueId u;
    NK_RCCL(ncclGetUniqueId(&u));
    memcpy(id, &u, NK_COMM_ID_BYTES);
    return NK_OK;
}

int nk_comm_init_rank(nk_device* dev, int nranks, int rank, const char id[NK_COMM_ID_BYTES], nk_comm** out) {
    NK_USE(dev);
    NK_CHECK(out && id, "null argument");
    NK_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d / %d", rank, nranks);
    ncclUniqueId u;
    memcpy(&u, id, NK_COMM_ID_BYTES);
    nk_comm* c = new nk_comm{dev, nullptr, rank, nranks};
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) { delete c; return fail_rccl(r, "ncclCommInitRank"); }
    *out = c;
    return NK_OK;
}

int nk_comm_init_all(int ndev, nk_device* const* devs, nk_comm** out) {
    NK_CHECK(ndev >= 1 && ndev <= 64 && devs && out, "bad nk_comm_init_all arguments");
    int ids[64];
    ncclComm_t comms[64];
    for (int i = 0; i < ndev; ++i) {
        NK_CHECK(devs[i] != nullptr, "null device handle %d", i);
        for (int j = 0; j < i; ++j) NK_CHECK(devs[j]->idx != devs[i]->idx, "nk_comm_init_all: GPU %d named twice", devs[i]->idx);
        ids[i] = devs[i]->idx;
    }
    NK_RCCL(ncclCommInitAll(comms, ndev, ids));
    for (int i = 0; i < ndev; ++i) out[i] = new nk_comm{devs[i], comms[i], i, ndev};
    return NK_OK;
}

int nk_comm_init_replicas(nk_device* dev, int nranks, int channels, double gbps, nk_comm** out) {
    NK_USE(dev);
    NK_CHECK(out != nullptr, "null argument");
    NK_CHECK(nranks >= 1, "bad replica count %d", nranks);
    NK_CHECK(channels >= 0 && channels <= 1024 && gbps >= 0.0, "bad replica pacing (channels %d, %g GB/s)", channels, gbps);
    nk_comm* c = new nk_comm{dev, nullptr, 0, nranks};
    c->channels = channels;
    c->gbps = gbps;
    *out = c;
    return NK_OK;
}

int nk_comm_destroy(nk_comm* comm) {
    if (!comm) return NK_OK;
    (void)hipSetDevice(comm->dev->idx);
    (void)hipStreamSynchronize(comm->dev->comm);
    if (comm->comm) (void)ncclCommDestroy(comm->comm);
    delete comm;
    return NK_OK;
}

// side stream waits for `after` (or for everything on the compute stream so far)
static int order_after(nk_device* dev, nk_event* after) {
    if (after) {
        NK_HIP(hipStreamWaitEvent(dev->comm, after->ev, 0));
    } else {
        NK_HIP(hipEventRecord(dev->fork, dev->compute));
        NK_HIP(hipStreamWaitEvent(dev->comm, dev->fork, 0));
    }
    return NK_OK;
}

int nk_allreduce_sum_group_async(nk_comm* comm, float* const* bufs, const size_t* counts, int nbufs, nk_event* after) {
    NK_CHECK(comm != nullptr, "null communicator");
    nk_device* dev = comm->dev;
    NK_USE(dev);
    NK_CHECK(nbufs >= 0 && (nbufs == 0 || (bufs && counts)), "bad buffer list");
    int live = 0;
    for (int i = 0; i < nbufs; ++i) {
        NK_CHECK(counts[i] == 0 || bufs[i] != nullptr, "null buffer %d", i);
        live += counts[i] != 0;
    }
    if (live == 0) return NK_OK;
    if (int rc = order_after(dev, after)) return rc;
    if (!comm->comm) {  // replica communicator: every virtual rank holds this rank's values
        for (int i = 0; i < nbufs; ++i) {
            if (counts[i] == 0) continue;
            if (comm->channels > 0 && (reinterpret_cast<uintptr_t>(bufs[i]) & 15) == 0) {  // (the paced walk uses 16-byte accesses)
                // seconds for this buffer at the emulated algorithm bandwidth, spread over the 16 KB steps of one workgroup
                const size_t per = (counts[i] / 4 + comm->channels - 1) / comm->channels, steps = (per + 1023) / 1024;  // float4s, 16 KB steps
                const double ticks = comm->gbps > 0.0 ? (counts[i] * 4.0 / (comm->gbps * 1e9)) * 1e8 / (double)(steps ? steps : 1) : 0.0;
                replica_sum_paced_kernel<<<comm->channels, 256, 0, dev->comm>>>(bufs[i], counts[i], (float)comm->size, ticks);
            } else {
                replica_sum_kernel<<<nk_stream_grid(counts[i], 256), 256, 0, dev->comm>>>(bufs[i], counts[i], (float)comm->size);
            }
            NK_LAUNCH_CHECK();
        }
        return NK_OK;
    }
    // one RCCL group = one fused launch for the whole list (latency-bound small gradients)
    if (live > 1) NK_RCCL(ncclGroupStart());
    ncclResult_t r = ncclSuccess;
    for (int i = 0; i < nbufs && r == ncclSuccess; ++i)
        if (counts[i]) r = ncclAllReduce(bufs[i], bufs[i], counts[i], ncclFloat32, ncclSum, comm->comm, dev->comm);
    if (live > 1) {
        ncclResult_t e = ncclGroupEnd();
        if (r == ncclSuccess) r = e;
    }
    if (r != ncclSuccess) return fail_rccl(r, "ncclAllReduce (group)");
    return NK_OK;
}

int nk_allreduce_sum_async(nk_comm* comm, float* buf, size_t n, nk_event* after) {
    return nk_allreduce_sum_group_async(comm, &buf, &n, 1, after);
}

int nk_comm_join(nk_comm* comm) {
    NK_CHECK(comm != nullptr, "null communicator");
    nk_device* dev = comm->dev;
    NK_USE(dev);
    NK_HIP(hipEventRecord(dev->join, dev->comm));
    NK_HIP(hipStreamWaitEvent(dev->compute, dev->join, 0));
    return NK_OK;
}

int nk_comm_rank(const nk_comm* comm) { return comm ? comm->rank : -1; }
int nk_comm_size(const nk_comm* comm) { return comm ? comm->size : 0; }

}  // extern "C"
