// Device / memory / event entry points of the C ABI (include/neuronika_hip.h).
// Mirrors the role of the reference's accelerator template: `Device::new`
// (neuronika-variable/src/cuda/device.rs:34-58) and `CuArray::{zeroed,from_slice,as_ndarray}`
// (cuda/cuarray.rs:35-117) — re-designed for HIP: two explicit streams per device (compute +
// communication), stream-ordered zero-fill, no library handles.
#include "nk_common.h"

static thread_local char g_err[512] = "";

void nk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int nk_fail_hip(hipError_t e, const char* what, const char* file, int line) {
    nk_set_error("HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return e == hipErrorOutOfMemory ? NK_ERR_OOM : NK_ERR_HIP;
}

static int grow_scratch(nk_device* dev, void*& block, size_t& have, size_t bytes, const char* what) {
    if (bytes > have) {
        size_t want = bytes < (size_t(64) << 20) ? (size_t(64) << 20) : bytes;
        if (block) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            NK_HIP(hipStreamIsCapturing(dev->compute, &cs));
            NK_CHECK(cs == hipStreamCaptureStatusNone,
                     "the device %s would have to grow (%zu -> %zu bytes) inside a captured region: run one eager step first", what, have, bytes);
            if (dev->graphs_alive > 0) {
                // a captured graph has the old pointer baked into its kernel arguments (split-K slabs, reduction partials,
                // conv tables): keep the block until the last graph of this device is destroyed
                dev->workspace_retired.push_back(block);
            } else {
                NK_HIP(hipDeviceSynchronize());
                NK_HIP(hipFree(block));
            }
            block = nullptr;
            have = 0;
        }
        NK_HIP(hipMalloc(&block, want));
        have = want;
    }
    return NK_OK;
}

int nk_workspace(nk_device* dev, size_t bytes, void** out) {
    const int rc = grow_scratch(dev, dev->workspace, dev->workspace_bytes, bytes, "workspace");
    if (rc) return rc;
    *out = dev->workspace;
    return NK_OK;
}

// A second stream-ordered scratch region, for an OPERAND a call builds for a kernel that uses the workspace itself (the padded copy
// of a folded-padding convolution entry that falls back to pad -> convolution).  Same growth rules as the workspace.
int nk_operand_scratch(nk_device* dev, size_t bytes, void** out) {
    const int rc = grow_scratch(dev, dev->operand_scratch, dev->operand_scratch_bytes, bytes, "operand scratch");
    if (rc) return rc;
    *out = dev->operand_scratch;
    return NK_OK;
}

int nk_prof_start(nk_device* dev, int klass, double flop) {
    if (!dev->prof_on) return NK_OK;
    nk_prof_rec r;
    if (!dev->prof_free.empty()) {
        r = dev->prof_free.back();
        dev->prof_free.pop_back();
    } else {
        NK_HIP(hipEventCreate(&r.start));
        NK_HIP(hipEventCreate(&r.stop));
    }
    r.klass = klass;
    r.flop = flop;
    NK_HIP(hipEventRecord(r.start, dev->compute));
    dev->prof.push_back(r);
    return NK_OK;
}
int nk_prof_stop(nk_device* dev) {
    if (!dev->prof_on || dev->prof.empty()) return NK_OK;
    NK_HIP(hipEventRecord(dev->prof.back().stop, dev->compute));
    return NK_OK;
}

extern "C" {

int nk_profile_begin(nk_device* dev) {
    NK_USE(dev);
    for (auto& r : dev->prof) dev->prof_free.push_back(r);
    dev->prof.clear();
    dev->prof_on = true;
    dev->prof_window = true;
    return NK_OK;
}

int nk_profile_pause(nk_device* dev, int paused) {
    NK_USE(dev);
    // only inside a begin / end window: resuming outside one would switch event pairs on with nothing to drain them
    NK_CHECK(dev->prof_window, "nk_profile_pause outside an nk_profile_begin / nk_profile_end window");
    dev->prof_on = paused == 0;   // the records collected so far stay: nk_profile_end reads them
    return NK_OK;
}

int nk_profile_end(nk_device* dev, int kernel_class, int* launches, double* total_ms, double* total_flop) {
    NK_USE(dev);
    NK_CHECK(launches && total_ms && total_flop, "null output");
    dev->prof_on = false;
    dev->prof_window = false;
    NK_HIP(hipStreamSynchronize(dev->compute));
    int n = 0;
    double ms = 0.0, flop = 0.0;
    for (auto& r : dev->prof) {
        if (r.klass != kernel_class) continue;
        float t = 0.f;
        NK_HIP(hipEventElapsedTime(&t, r.start, r.stop));
        ms += t; flop += r.flop; ++n;
    }
    *launches = n; *total_ms = ms; *total_flop = flop;
    return NK_OK;
}

const char* nk_last_error(void) { return g_err; }
const char* nk_version(void) { return "neuronika_hip 0.1 (gfx950)"; }

int nk_device_count(int* out) {
    NK_CHECK(out != nullptr, "null out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *out = n;
    return NK_OK;
}

int nk_device_create(int idx, nk_device** out) {
    NK_CHECK(out != nullptr, "null out");
    int n = 0;
    NK_HIP(hipGetDeviceCount(&n));
    NK_CHECK(idx >= 0 && idx < n, "device index %d out of range (%d devices)", idx, n);
    NK_HIP(hipSetDevice(idx));
    nk_device* d = new nk_device();
    d->idx = idx;
    NK_HIP(hipStreamCreateWithFlags(&d->compute, hipStreamNonBlocking));
    {   // the all-reduce stream gets the highest priority: its few workgroups must be placed as soon as a GEMM block
        // retires, not after the (long) GEMM grid of the compute stream has been dispatched
        int least = 0, greatest = 0;
        NK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        NK_HIP(hipStreamCreateWithPriority(&d->comm, hipStreamNonBlocking, greatest));
    }
    NK_HIP(hipStreamCreateWithFlags(&d->copy, hipStreamNonBlocking));
    NK_HIP(hipEventCreateWithFlags(&d->fork, hipEventDisableTiming));
    NK_HIP(hipEventCreateWithFlags(&d->join, hipEventDisableTiming));
    hipDeviceProp_t prop;
    NK_HIP(hipGetDeviceProperties(&prop, idx));
    d->num_cus = prop.multiProcessorCount;
    *out = d;
    return NK_OK;
}

int nk_device_destroy(nk_device* dev) {
    if (!dev) return NK_OK;
    NK_HIP(hipSetDevice(dev->idx));
    NK_HIP(hipDeviceSynchronize());
    if (dev->workspace) (void)hipFree(dev->workspace);
    if (dev->operand_scratch) (void)hipFree(dev->operand_scratch);
    for (void* w : dev->workspace_retired) (void)hipFree(w);
    for (auto* v : {&dev->prof, &dev->prof_free})
        for (auto& r : *v) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    (void)hipEventDestroy(dev->fork);
    (void)hipEventDestroy(dev->join);
    (void)hipStreamDestroy(dev->compute);
    (void)hipStreamDestroy(dev->comm);
    (void)hipStreamDestroy(dev->copy);
    delete dev;
    return NK_OK;
}

int nk_device_set_busy_slots(nk_device* dev, int n) {
    NK_CHECK(dev != nullptr && n >= 0 && n <= dev->num_cus, "busy slots: 0 .. number of CUs");
    dev->busy_slots = n;
    return NK_OK;
}

int nk_conv_winograd_launches(nk_device* dev, uint64_t* count) {
    NK_CHECK(dev != nullptr && count != nullptr, "bad nk_conv_winograd_launches arguments");
    *count = dev->wino_launches;
    return NK_OK;
}

int nk_dev_tune(nk_device* dev, int knob, const int* values, int n) {
    NK_CHECK(dev != nullptr && n >= 0 && (n == 0 || values != nullptr), "bad nk_dev_tune arguments");
    switch (knob) {
        case NK_TUNE_GEMM_FORCE:
            NK_CHECK(n <= 6, "NK_TUNE_GEMM_FORCE takes at most 6 values");
            for (int i = 0; i < 6; ++i) dev->tune_gemm[i] = i < n ? values[i] : 0;
            dev->tune_gemm_n = n;
            return NK_OK;
        case NK_TUNE_GEMM_KPAIR:
            NK_CHECK(n <= 1 && (n == 0 || (values[0] >= -1 && values[0] <= 2)), "NK_TUNE_GEMM_KPAIR: -1, 0, 1 or 2");
            dev->tune_kpair = n ? values[0] : -1;
            return NK_OK;
        case NK_TUNE_GEMM_PAIR:
            NK_CHECK(n <= 1 && (n == 0 || (values[0] >= -1 && values[0] <= 1)), "NK_TUNE_GEMM_PAIR: -1, 0 or 1");
            dev->tune_pair = n ? values[0] : -1;
            return NK_OK;
        case NK_TUNE_CONV_NARROW:
            NK_CHECK(n <= 1 && (n == 0 || (values[0] >= 0 && values[0] <= 100)), "NK_TUNE_CONV_NARROW: 0 (off) or a cost in percent, 1..100");
            dev->tune_conv_narrow = n ? values[0] : -1;
            return NK_OK;
        case NK_TUNE_CONV_WINOGRAD:
            NK_CHECK(n <= 4 && (n == 0 || (values[0] >= -1 && values[0] <= 1)) && (n < 2 || values[1] >= -1) && (n < 3 || (values[2] >= -1 && values[2] <= 1)) &&
                         (n < 4 || (values[3] >= -1 && values[3] <= 1)),
                     "NK_TUNE_CONV_WINOGRAD: -1, 0 or 1[, stagger >= -1[, block shape -1, 0 or 1[, kernel gradient -1, 0 or 1]]]");
            dev->tune_conv_winograd = n ? values[0] : -1;
            dev->tune_conv_wino_stagger = n >= 2 ? values[1] : -1;
            dev->tune_conv_wino_shape = n >= 3 ? values[2] : -1;
            dev->tune_conv_wino_dw = n >= 4 ? values[3] : -1;
            return NK_OK;
        case NK_TUNE_CONV_S2DX:
            NK_CHECK(n <= 1 && (n == 0 || (values[0] >= -1 && values[0] <= 3)), "NK_TUNE_CONV_S2DX: -1, 0, 1, 2 (narrow blocks) or 3 (wide blocks)");
            dev->tune_conv_s2dx = n ? values[0] : -1;
            return NK_OK;
        case NK_TUNE_GEMM_CHAIN:
            NK_CHECK(n <= 1 && (n == 0 || values[0] == -1 || values[0] == 0 || (values[0] >= 64 && values[0] % 64 == 0)),
                     "NK_TUNE_GEMM_CHAIN: -1 (rule), 0 (one chain) or a chain length that is a multiple of 64");
            dev->tune_chain = n ? values[0] : -1;
            return NK_OK;
        case NK_TUNE_ATTENTION_OCC:
            NK_CHECK(n <= 1 && (n == 0 || values[0] == 0 || values[0] == 2), "NK_TUNE_ATTENTION_OCC: 0 or 2");
            dev->tune_attn_occ = n ? values[0] : 0;
            return NK_OK;
    }
    nk_set_error("unknown tuning knob %d", knob);
    return NK_ERR_INVALID;
}

int nk_device_sync(nk_device* dev) {
    NK_USE(dev);
    NK_HIP(hipStreamSynchronize(dev->compute));
    NK_HIP(hipStreamSynchronize(dev->comm));
    NK_HIP(hipStreamSynchronize(dev->copy));
    return NK_OK;
}

int nk_device_index(const nk_device* dev) { return dev ? dev->idx : -1; }
void* nk_stream_compute(nk_device* dev) { return dev ? (void*)dev->compute : nullptr; }
void* nk_stream_comm(nk_device* dev) { return dev ? (void*)dev->comm : nullptr; }

int nk_alloc_zeroed(nk_device* dev, size_t n, float** out) {
    NK_USE(dev);
    NK_CHECK(out != nullptr, "null out");
    void* p = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(float);
    NK_HIP(hipMalloc(&p, bytes));
    NK_HIP(hipMemsetAsync(p, 0, bytes, dev->compute));
    *out = (float*)p;
    return NK_OK;
}

int nk_free(nk_device* dev, float* ptr) {
    NK_USE(dev);
    if (ptr) NK_HIP(hipFree(ptr));
    return NK_OK;
}

int nk_upload(nk_device* dev, float* dst, const float* host_src, size_t n) {
    NK_USE(dev);
    if (n == 0) return NK_OK;
    NK_CHECK(dst && host_src, "null pointer in nk_upload");
    NK_HIP(hipMemcpyAsync(dst, host_src, n * sizeof(float), hipMemcpyHostToDevice, dev->compute));
    // pageable host memory: the copy has been staged when the call returns, but keep the
    // contract simple for the host (the reference's from_slice is synchronous too).
    NK_HIP(hipStreamSynchronize(dev->compute));
    return NK_OK;
}

// ---- hipGraph capture of a launch-bound step -------------------------------------------------------
struct nk_graph {
    nk_device* dev;  // must outlive the graph (its workspace pointers are baked into the captured kernel arguments)
    int idx;
    hipStream_t stream;
    hipGraph_t graph;
    hipGraphExec_t exec;
};

int nk_graph_begin(nk_device* dev) {
    NK_USE(dev);
    NK_HIP(hipStreamBeginCapture(dev->compute, hipStreamCaptureModeRelaxed));
    return NK_OK;
}

int nk_graph_end(nk_device* dev, nk_graph** out) {
    NK_USE(dev);
    NK_CHECK(out != nullptr, "null out pointer");
    *out = nullptr;
    hipGraph_t g = nullptr;
    NK_HIP(hipStreamEndCapture(dev->compute, &g));
    NK_CHECK(g != nullptr, "stream capture produced no graph (a synchronising call inside the captured region?)");
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        return nk_fail_hip(e, "hipGraphInstantiate", __FILE__, __LINE__);
    }
    *out = new nk_graph{dev, dev->idx, dev->compute, g, exec};
    ++dev->graphs_alive;
    return NK_OK;
}

int nk_graph_launch(nk_graph* g) {
    NK_CHECK(g != nullptr, "null graph");
    NK_HIP(hipSetDevice(g->idx));
    NK_HIP(hipGraphLaunch(g->exec, g->stream));
    return NK_OK;
}

int nk_graph_destroy(nk_graph* g) {
    if (!g) return NK_OK;
    (void)hipSetDevice(g->idx);
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    if (--g->dev->graphs_alive == 0 && !g->dev->workspace_retired.empty()) {  // nobody can replay into the outgrown blocks now
        (void)hipDeviceSynchronize();
        for (void* w : g->dev->workspace_retired) (void)hipFree(w);
        g->dev->workspace_retired.clear();
    }
    delete g;
    return NK_OK;
}

int nk_host_alloc(size_t bytes, void** out) {
    NK_CHECK(out != nullptr, "null out pointer");
    *out = nullptr;
    if (bytes == 0) return NK_OK;
    NK_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return NK_OK;
}

int nk_host_free(void* p) {
    if (p) NK_HIP(hipHostFree(p));
    return NK_OK;
}

int nk_upload_async(nk_device* dev, float* dst, const float* pinned_src, size_t n) {
    NK_USE(dev);
    if (n == 0) return NK_OK;
    NK_CHECK(dst && pinned_src, "null pointer in nk_upload_async");
    NK_HIP(hipMemcpyAsync(dst, pinned_src, n * sizeof(float), hipMemcpyHostToDevice, dev->copy));
    return NK_OK;
}

int nk_download(nk_device* dev, float* host_dst, const float* src, size_t n) {
    NK_USE(dev);
    if (n == 0) return NK_OK;
    NK_CHECK(host_dst && src, "null pointer in nk_download");
    NK_HIP(hipMemcpyAsync(host_dst, src, n * sizeof(float), hipMemcpyDeviceToHost, dev->compute));
    NK_HIP(hipStreamSynchronize(dev->compute));
    return NK_OK;
}

// d2d copy as a span-walk kernel (nk_common.h): 5.7 TB/s at 1 GiB where the runtime's blit reaches 5.1 (round 6, same box)
__global__ void copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4) {
    nk_span_walk<4>(n4, [&](size_t i) { return nk_load_stream(src + i, true); }, [&](size_t i, const float4& v) { nk_store_stream(dst + i, v); });
}
int nk_copy(nk_device* dev, float* dst, const float* src, size_t n) {
    NK_USE(dev);
    if (n == 0) return NK_OK;
    const bool aligned = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
    const bool overlap = dst < src + n && src < dst + n;
    if (aligned && !overlap && n % 4 == 0 && n >= (size_t(1) << 20)) {
        hipLaunchKernelGGL(copy_kernel, dim3(nk_stream_grid(n / 4, 256)), dim3(256), 0, dev->compute, reinterpret_cast<float4*>(dst),
                           reinterpret_cast<const float4*>(src), n / 4);
        NK_LAUNCH_CHECK();
        return NK_OK;
    }
    NK_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, dev->compute));
    return NK_OK;
}

int nk_event_create(nk_device* dev, nk_event** out) {
    NK_USE(dev);
    NK_CHECK(out != nullptr, "null out");
    nk_event* e = new nk_event{dev->idx, dev->compute, dev->comm, dev->copy, nullptr};
    NK_HIP(hipEventCreate(&e->ev));
    *out = e;
    return NK_OK;
}

int nk_event_destroy(nk_event* ev) {
    if (!ev) return NK_OK;
    (void)hipSetDevice(ev->idx);
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return NK_OK;
}

int nk_event_record(nk_event* ev, int on_comm_stream) {
    NK_CHECK(ev != nullptr, "null event");
    NK_HIP(hipSetDevice(ev->idx));
    NK_CHECK(on_comm_stream >= 0 && on_comm_stream <= 2, "unknown stream %d", on_comm_stream);
    NK_HIP(hipEventRecord(ev->ev, on_comm_stream == 2 ? ev->copy : (on_comm_stream ? ev->comm : ev->compute)));
    return NK_OK;
}

int nk_event_sync(nk_event* ev) {
    NK_CHECK(ev != nullptr, "null event");
    NK_HIP(hipSetDevice(ev->idx));
    NK_HIP(hipEventSynchronize(ev->ev));
    return NK_OK;
}

int nk_event_elapsed_ms(nk_event* start, nk_event* stop, float* ms) {
    NK_CHECK(start && stop && ms, "null argument");
    NK_HIP(hipSetDevice(start->idx));
    NK_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return NK_OK;
}

int nk_stream_wait_event(nk_device* dev, int on_comm_stream, nk_event* ev) {
    NK_USE(dev);
    NK_CHECK(ev != nullptr, "null event");
    NK_CHECK(on_comm_stream >= 0 && on_comm_stream <= 2, "unknown stream %d", on_comm_stream);
    NK_HIP(hipStreamWaitEvent(on_comm_stream == 2 ? dev->copy : (on_comm_stream ? dev->comm : dev->compute), ev->ev, 0));
    return NK_OK;
}

}  // extern "C"
