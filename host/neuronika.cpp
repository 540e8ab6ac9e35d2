// Implementation of the host-side tape mirror (see neuronika.hpp).  Every node's forward() /
// backward() body is ONE call into the C ABI (include/neuronika_hip.h), issued asynchronously
// on the device's compute stream in tape order.
#include "neuronika.hpp"

#include <cstdlib>

#include <algorithm>
#include <charconv>
#include <cstring>
#include <cmath>
#include <random>

namespace neuronika {

void panic(const std::string& msg) { throw Panic(msg); }
void check(int status) {
    if (status != NK_OK) panic(nk_last_error());
}
size_t numel(const Shape& s) {
    size_t n = 1;
    for (int d : s) n *= (size_t)d;
    return n;
}

// =================================================================================================
// Device / HipArray / Gradient
// =================================================================================================
std::shared_ptr<Device> Device::create(int idx) {
    std::shared_ptr<Device> d(new Device());
    check(nk_device_create(idx, &d->h_));
    d->idx_ = idx;
    return d;
}
Device::~Device() {
    if (!h_) return;
    nk_device_sync(h_);
    for (auto& kv : pool_)
        for (float* p : kv.second) nk_free(h_, p);
    nk_device_destroy(h_);
}
void Device::sync() const { check(nk_device_sync(h_)); }
float* Device::alloc_zeroed(size_t n) {
    auto it = pool_.find(n);
    float* p = nullptr;
    if (it != pool_.end() && !it->second.empty()) {
        p = it->second.back();
        it->second.pop_back();
        check(nk_fill(h_, p, n, 0.f));
    } else {
        check(nk_alloc_zeroed(h_, n, &p));
    }
    in_use_ += n * sizeof(float);
    return p;
}
void Device::graph_begin() { check(nk_graph_begin(h_)); }
std::shared_ptr<Graph> Device::graph_end() {
    nk_graph* g = nullptr;
    check(nk_graph_end(h_, &g));
    return std::make_shared<Graph>(g, shared_from_this());
}
Graph::~Graph() { (void)nk_graph_destroy(g_); }  // dev_ (a member) is released after this body: the device outlives its graphs
void Graph::launch() const { check(nk_graph_launch(g_)); }

float* Device::alloc_uninit(size_t n) {
    auto it = pool_.find(n);
    float* p = nullptr;
    if (it != pool_.end() && !it->second.empty()) {
        p = it->second.back();
        it->second.pop_back();
    } else {
        check(nk_alloc_zeroed(h_, n, &p));
    }
    in_use_ += n * sizeof(float);
    return p;
}
void Device::release(float* p, size_t n) {
    if (!p) return;
    pool_[n].push_back(p);
    in_use_ -= n * sizeof(float);
}

HipArray::HipArray(DevicePtr dev, Shape shape)
    : dev_(std::move(dev)), shape_(std::move(shape)), len_(numel(shape_)), ptr_(dev_->alloc_zeroed(len_)) {}
HipArray::HipArray(DevicePtr dev, Shape shape, Uninit)
    : dev_(std::move(dev)), shape_(std::move(shape)), len_(numel(shape_)), ptr_(dev_->alloc_uninit(len_)) {}
HipArray::HipArray(std::shared_ptr<HipArray> parent, size_t offset, Shape shape)
    : dev_(parent->device()), shape_(std::move(shape)), len_(numel(shape_)), ptr_(parent->ptr() + offset), parent_(std::move(parent)) {
    if (offset + len_ > parent_->len()) panic("HipArray view: out of the parent's range");
}
HipArray::~HipArray() {
    if (!parent_) dev_->release(ptr_, len_);
}
std::shared_ptr<HipArray> HipArray::from_host(DevicePtr dev, const Shape& shape, const float* host) {
    auto a = std::make_shared<HipArray>(std::move(dev), shape);
    a->upload(host);
    return a;
}
void HipArray::upload(const float* host) { check(nk_upload(dev_->raw(), ptr_, host, len_)); }
void HipArray::download(float* host) const { check(nk_download(dev_->raw(), host, ptr_, len_)); }
std::vector<float> HipArray::to_vec() const {
    std::vector<float> v(len_);
    download(v.data());
    return v;
}
void HipArray::fill(float v) { check(nk_fill(dev_->raw(), ptr_, len_, v)); }

Gradient::Gradient(DevicePtr dev, Shape shape)
    : dev_(std::move(dev)), shape_(std::move(shape)), array_(std::make_shared<HipArray>(dev_, shape_, HipArray::Uninit{})) {}
Gradient::Gradient(Shared<HipArray> storage, size_t offset, Shape shape)
    : dev_(storage->device()), shape_(std::move(shape)), array_(std::make_shared<HipArray>(storage, offset, shape_)), storage_(std::move(storage)),
      offset_(offset) {}
HipArray& Gradient::borrow() const {
    if (!array_)
        panic("Trying to get a de-allocated gradient. Switch on the gradients first by using `.with_grad()`");
    if (pending_zero_) {
        array_->fill(0.f);
        pending_zero_ = false;
    }
    return *array_;
}
Shared<HipArray> Gradient::exchange_array(Shared<HipArray> external) {
    if (!array_) panic("Trying to get a de-allocated gradient. Switch on the gradients first by using `.with_grad()`");
    if (external->len() != array_->len()) panic("exchange_array: extent mismatch");
    std::swap(array_, external);
    pending_zero_ = false;
    return external;
}
HipArray& Gradient::borrow_first_write(bool& assign) const {
    if (!array_)
        panic("Trying to get a de-allocated gradient. Switch on the gradients first by using `.with_grad()`");
    assign = pending_zero_;
    pending_zero_ = false;
    return *array_;
}
void Gradient::zero() {
    if (!array_)
        panic("Trying to get a de-allocated gradient. Switch on the gradients first by using `.with_grad()`");
    pending_zero_ = true;
}
void Gradient::no_grad() { array_.reset(); }
void Gradient::with_grad() {
    if (!array_) {
        array_ = storage_ ? std::make_shared<HipArray>(storage_, offset_, shape_) : std::make_shared<HipArray>(dev_, shape_, HipArray::Uninit{});
        pending_zero_ = true;
    }
}

// =================================================================================================
// History
// =================================================================================================
template <class T>
void History<T>::merge(const History& other) {
    for (const Item& it : other.path_) {
        bool present = false;
        for (const Item& mine : path_)
            if (mine.ptr == it.ptr) { present = true; break; }
        if (!present) path_.push_back(it);
    }
    std::stable_sort(path_.begin(), path_.end(), [](const Item& a, const Item& b) { return a.order < b.order; });
}
template <class T>
void History<T>::insert(const void* ptr, T op) {
    path_.push_back(Item{ptr, path_.size(), std::move(op)});
    buffer_->clear();
}
template <class T>
std::vector<T> History<T>::to_vec() const {
    std::vector<T> v;
    v.reserve(path_.size());
    for (const Item& it : path_) v.push_back(it.op);
    return v;
}
template class History<ForwardEntry>;
template class History<BackwardEntry>;

// =================================================================================================
// helpers
// =================================================================================================
// Philox key of the next random node (Dropout, fused attention probabilities).  The reference draws from
// `rand::thread_rng()` (dropout/mod.rs:68-70), seeded by the OS; here every node gets its own key from one per-thread
// sequence that `manual_seed` restarts, so a run is reproducible and data-parallel ranks can decorrelate their masks
// (`manual_seed(base + rank)`).
static thread_local uint64_t g_node_seed = 0x9E3779B97F4A7C15ull;
void manual_seed(uint64_t seed) { g_node_seed = seed; }
static uint64_t next_node_seed() {
    const uint64_t s = g_node_seed;
    g_node_seed += 0x632BE59BD9B4E019ull;
    return s;
}

namespace {

nk_device* D(const Shared<HipArray>& a) { return a->device()->raw(); }
Shared<HipArray> zeros_like(const Shared<HipArray>& a, Shape s) { return std::make_shared<HipArray>(a->device(), std::move(s)); }

Shape cobroadcast(const Shape& l, const Shape& r) {  // utils.rs:97-125
    const Shape& big = l.size() >= r.size() ? l : r;
    const Shape& small = l.size() >= r.size() ? r : l;
    Shape out = big;
    const size_t off = big.size() - small.size();
    for (size_t i = 0; i < small.size(); ++i) {
        int& o = out[off + i];
        if (o != small[i]) {
            if (o == 1) o = small[i];
            else if (small[i] != 1) panic("The two tensors have incompatible shape.");
        }
    }
    return out;
}

// ---- forward nodes -------------------------------------------------------------------------------
struct BinaryFwd : Forward {  // node/{addition,subtraction,multiplication,division}/mod.rs
    int op;
    Shared<HipArray> l, r, out;
    BinaryFwd(int op, Shared<HipArray> l, Shared<HipArray> r, Shared<HipArray> out)
        : op(op), l(std::move(l)), r(std::move(r)), out(std::move(out)) {}
    void forward() const override {
        check(nk_binary_fwd(D(out), op, out->ptr(), out->shape().data(), (int)out->shape().size(), l->ptr(),
                            l->shape().data(), (int)l->shape().size(), r->ptr(), r->shape().data(),
                            (int)r->shape().size()));
    }
};
struct BinaryBwd : Backward {  // <Op>Backward{Left,Right}; either side may be absent
    int op;
    Shared<Gradient> lg, rg, g;
    Shared<HipArray> l, r;
    void backward() const override {
        const HipArray& G = g->borrow();
        bool assign = false;
        if (lg) {
            HipArray& d = lg->borrow_first_write(assign);
            check((assign ? nk_binary_bwd_left_assign : nk_binary_bwd_left)(D(l), op, d.ptr(), d.shape().data(), (int)d.shape().size(), G.ptr(),
                                     G.shape().data(), (int)G.shape().size(), r->ptr(), r->shape().data(),
                                     (int)r->shape().size()));
        }
        if (rg) {
            HipArray& d = rg->borrow_first_write(assign);
            check((assign ? nk_binary_bwd_right_assign : nk_binary_bwd_right)(D(l), op, d.ptr(), d.shape().data(), (int)d.shape().size(), G.ptr(),
                                      G.shape().data(), (int)G.shape().size(), l->ptr(), l->shape().data(),
                                      (int)l->shape().size(), r->ptr()));
        }
    }
    void targets(std::vector<const Gradient*>& out) const override {
        if (lg) out.push_back(lg.get());
        if (rg) out.push_back(rg.get());
    }
};

// First-write access to a gradient: `beta` = 0 when its zero fill is still pending (the node assigns), 1 otherwise.
static float* first_write(const Shared<Gradient>& gr, float& beta) {
    bool assign = false;
    HipArray& d = gr->borrow_first_write(assign);
    beta = assign ? 0.f : 1.f;
    return d.ptr();
}

enum class Unary { Relu, Softmax, LogSoftmax, Transpose, Sum, Mean };
struct UnaryFwd : Forward {
    Unary kind;
    int axis;
    Shared<HipArray> x, y;
    UnaryFwd(Unary k, int axis, Shared<HipArray> x, Shared<HipArray> y) : kind(k), axis(axis), x(std::move(x)), y(std::move(y)) {}
    void forward() const override {
        const int nd = (int)x->shape().size();
        switch (kind) {
            case Unary::Relu: check(nk_relu_fwd(D(x), x->ptr(), y->ptr(), x->len())); break;
            case Unary::Softmax: check(nk_softmax_fwd(D(x), x->ptr(), y->ptr(), x->shape().data(), nd, axis)); break;
            case Unary::LogSoftmax: check(nk_log_softmax_fwd(D(x), x->ptr(), y->ptr(), x->shape().data(), nd, axis)); break;
            case Unary::Transpose: check(nk_transpose_fwd(D(x), x->ptr(), y->ptr(), x->shape().data(), nd)); break;
            case Unary::Sum: check(nk_sum_fwd(D(x), x->ptr(), x->len(), y->ptr())); break;
            case Unary::Mean: check(nk_mean_fwd(D(x), x->ptr(), x->len(), y->ptr())); break;
        }
    }
};
struct UnaryBwd : Backward {
    Unary kind;
    int axis;
    Shared<Gradient> dx, g;
    Shared<HipArray> x, y;  // ReLU needs the input, (log)softmax the output
    void backward() const override {
        const HipArray& G = g->borrow();
        if (kind == Unary::Relu || kind == Unary::Sum || kind == Unary::Mean) {
            bool assign = false;
            HipArray& d = dx->borrow_first_write(assign);
            if (kind == Unary::Relu) check((assign ? nk_relu_bwd_assign : nk_relu_bwd)(D(x), d.ptr(), G.ptr(), x->ptr(), d.len()));
            else if (kind == Unary::Sum) check((assign ? nk_sum_bwd_assign : nk_sum_bwd)(d.device()->raw(), d.ptr(), d.len(), G.ptr()));
            else check((assign ? nk_mean_bwd_assign : nk_mean_bwd)(d.device()->raw(), d.ptr(), d.len(), G.ptr()));
            return;
        }
        if (kind == Unary::Transpose) {
            bool assign = false;
            HipArray& d = dx->borrow_first_write(assign);
            check((assign ? nk_transpose_bwd_assign : nk_transpose_bwd)(d.device()->raw(), d.ptr(), G.ptr(), d.shape().data(), (int)d.shape().size()));
            return;
        }
        if (kind == Unary::Softmax || kind == Unary::LogSoftmax) {
            bool assign = false;
            HipArray& d = dx->borrow_first_write(assign);
            auto fn = kind == Unary::Softmax ? (assign ? nk_softmax_bwd_assign : nk_softmax_bwd)
                                             : (assign ? nk_log_softmax_bwd_assign : nk_log_softmax_bwd);
            check(fn(D(y), d.ptr(), G.ptr(), y->ptr(), d.shape().data(), (int)d.shape().size(), axis));
            return;
        }
        panic("UnaryBwd: unknown node kind");
    }
    void targets(std::vector<const Gradient*>& out) const override { out.push_back(dx.get()); }
};

// MatMul family.  kind: 0 = mm, 1 = mm_t, 2 = batched mm over [B,*,*] tiles, 3 = batched mm_t
struct MatMulFwd : Forward {
    int kind;
    Shared<HipArray> a, b, c;
    MatMulFwd(int kind, Shared<HipArray> a, Shared<HipArray> b, Shared<HipArray> c)
        : kind(kind), a(std::move(a)), b(std::move(b)), c(std::move(c)) {}
    void forward() const override {
        const Shape& as = a->shape();
        const Shape& bs = b->shape();
        if (kind == 0) check(nk_mm_fwd(D(a), a->ptr(), b->ptr(), c->ptr(), as[0], as[1], bs[1]));
        else if (kind == 4) check(nk_mv_fwd(D(a), a->ptr(), b->ptr(), c->ptr(), as[0], as[1]));
        else if (kind == 5) check(nk_vm_fwd(D(a), a->ptr(), b->ptr(), c->ptr(), bs[0], bs[1]));
        else if (kind == 6) check(nk_vv_fwd(D(a), a->ptr(), b->ptr(), a->len(), c->ptr()));
        else if (kind == 1) check(nk_mm_t_fwd(D(a), a->ptr(), b->ptr(), c->ptr(), as[0], as[1], bs[0]));
        else if (kind == 2)  // C[b] (n,o) = A[b] (n,m) . B[b] (m,o)
            check(nk_sgemm_batched(D(a), 0, 0, as[1], bs[2], as[2], 1.f, a->ptr(), as[2], (long long)as[1] * as[2], 0,
                                   b->ptr(), bs[2], (long long)bs[1] * bs[2], 0, 0.f, c->ptr(), bs[2],
                                   (long long)as[1] * bs[2], 0, as[0], 1));
        else  // C[b] (n,o) = A[b] (n,m) . B[b] (o,m)^T
            check(nk_sgemm_batched(D(a), 0, 1, as[1], bs[1], as[2], 1.f, a->ptr(), as[2], (long long)as[1] * as[2], 0,
                                   b->ptr(), bs[2], (long long)bs[1] * bs[2], 0, 0.f, c->ptr(), bs[1],
                                   (long long)as[1] * bs[1], 0, as[0], 1));
    }
};
struct MatMulBwd : Backward {
    int kind;
    Shared<HipArray> a, b;
    Shared<Gradient> da, db, g;  // da / db may be null
    void backward() const override {
        const HipArray& G = g->borrow();
        const Shape& as = a->shape();
        const Shape& bs = b->shape();
        nk_device* dev = D(a);
        if (kind == 4) {  // matrix_vector_mul/mod.rs:63-69, :92-102
            if (da) check(nk_mv_bwd_left(dev, da->borrow().ptr(), G.ptr(), b->ptr(), as[0], as[1]));
            if (db) check(nk_mv_bwd_right(dev, db->borrow().ptr(), a->ptr(), G.ptr(), as[0], as[1]));
        } else if (kind == 5) {  // vector_matrix_mul/mod.rs:63-73, :95-101
            if (da) check(nk_vm_bwd_left(dev, da->borrow().ptr(), b->ptr(), G.ptr(), bs[0], bs[1]));
            if (db) check(nk_vm_bwd_right(dev, db->borrow().ptr(), a->ptr(), G.ptr(), bs[0], bs[1]));
        } else if (kind == 6) {  // vector_vector_mul/mod.rs:57-63
            if (da) check(nk_vv_bwd(dev, da->borrow().ptr(), b->ptr(), G.ptr(), a->len()));
            if (db) check(nk_vv_bwd(dev, db->borrow().ptr(), a->ptr(), G.ptr(), a->len()));
        } else if (kind == 0) {  // = nk_mm_bwd_left / nk_mm_bwd_right with beta 0 on a first write
            const int n = as[0], m = as[1], o = bs[1];
            float beta;
            if (da && db && da != db) {  // MatrixMatrixMulBackward::backward: both products as one call (one launch at small sizes)
                float beta_b;
                float* d = first_write(da, beta);
                float* e = first_write(db, beta_b);
                check(nk_mm_bwd(dev, d, e, G.ptr(), a->ptr(), b->ptr(), n, m, o, beta == 0.f, beta_b == 0.f));
                return;
            }
            if (da) { float* d = first_write(da, beta); check(nk_sgemm(dev, 0, 1, n, m, o, 1.f, G.ptr(), o, b->ptr(), o, beta, d, m)); }
            if (db) { float* d = first_write(db, beta); check(nk_sgemm(dev, 1, 0, m, o, n, 1.f, a->ptr(), m, G.ptr(), o, beta, d, o)); }
        } else if (kind == 1) {  // = nk_mm_t_bwd_left / nk_mm_t_bwd_right
            const int n = as[0], m = as[1], o = bs[0];
            float beta;
            if (da && db && da != db) {  // MatrixMatrixMulTBackward::backward as one call
                float beta_b;
                float* d = first_write(da, beta);
                float* e = first_write(db, beta_b);
                check(nk_mm_t_bwd(dev, d, e, G.ptr(), a->ptr(), b->ptr(), n, m, o, beta == 0.f, beta_b == 0.f));
                return;
            }
            if (da) { float* d = first_write(da, beta); check(nk_sgemm(dev, 0, 0, n, m, o, 1.f, G.ptr(), o, b->ptr(), m, beta, d, m)); }
            if (db) { float* d = first_write(db, beta); check(nk_sgemm(dev, 1, 0, o, m, n, 1.f, G.ptr(), o, a->ptr(), m, beta, d, m)); }
        } else if (kind == 2) {
            const int B = as[0], n = as[1], m = as[2], o = bs[2];
            float beta;
            if (da) {  // dA[b] += G[b] . B[b]^T
                float* d = first_write(da, beta);
                check(nk_sgemm_batched(dev, 0, 1, n, m, o, 1.f, G.ptr(), o, (long long)n * o, 0, b->ptr(), o,
                                       (long long)m * o, 0, beta, d, m, (long long)n * m, 0, B, 1));
            }
            if (db) {  // dB[b] += A[b]^T . G[b]
                float* d = first_write(db, beta);
                check(nk_sgemm_batched(dev, 1, 0, m, o, n, 1.f, a->ptr(), m, (long long)n * m, 0, G.ptr(), o,
                                       (long long)n * o, 0, beta, d, o, (long long)m * o, 0, B, 1));
            }
        } else {
            const int B = as[0], n = as[1], m = as[2], o = bs[1];
            float beta;
            if (da) {  // dA[b] += G[b] . B[b]
                float* d = first_write(da, beta);
                check(nk_sgemm_batched(dev, 0, 0, n, m, o, 1.f, G.ptr(), o, (long long)n * o, 0, b->ptr(), m,
                                       (long long)o * m, 0, beta, d, m, (long long)n * m, 0, B, 1));
            }
            if (db) {  // dB[b] += G[b]^T . A[b]
                float* d = first_write(db, beta);
                check(nk_sgemm_batched(dev, 1, 0, o, m, n, 1.f, G.ptr(), o, (long long)n * o, 0, a->ptr(), m,
                                       (long long)n * m, 0, beta, d, m, (long long)o * m, 0, B, 1));
            }
        }
    }
    void targets(std::vector<const Gradient*>& out) const override {
        if (da) out.push_back(da.get());
        if (db) out.push_back(db.get());
    }
};

struct ConvFwd : Forward {  // node/convolution/mod.rs:296-355 (+ the module's broadcast bias Addition when b is set)
    Shared<HipArray> x, w, b, y;
    std::vector<int> stride, dilation;
    std::vector<int> fold;  // module node with its Zero padding folded in: x is the UNPADDED input (nk_conv_bias_fwd_padded)
    int groups;
    void forward() const override {
        const int nd = (int)x->shape().size() - 2;
        if (!fold.empty())
            check(nk_conv_bias_fwd_padded(D(x), nd, x->ptr(), x->shape().data(), fold.data(), w->ptr(), w->shape().data(), b ? b->ptr() : nullptr,
                                          y->ptr(), stride.data(), dilation.data(), groups));
        else if (b)
            check(nk_conv_bias_fwd(D(x), nd, x->ptr(), x->shape().data(), w->ptr(), w->shape().data(), b->ptr(), y->ptr(),
                                   stride.data(), dilation.data(), groups));
        else
            check(nk_conv_fwd(D(x), nd, x->ptr(), x->shape().data(), w->ptr(), w->shape().data(), y->ptr(), stride.data(),
                              dilation.data(), groups));
    }
};
struct ConvBwd : Backward {  // ConvolutionBackward{Input,Kernel}  :357-510
    Shared<HipArray> x, w;
    Shared<Gradient> dx, dw, db, g;  // dx may be null (input is a non-differentiable Var); db only for the fused module node
    std::vector<int> stride, dilation;
    std::vector<int> padding;  // fused module node with Zero padding: x is the padded input, dx the UNPADDED input's gradient
    bool x_unpadded = false;   // ... and with the padding folded into forward and kernel gradient too: x is the UNPADDED input
    int groups;
    void backward() const override {
        const HipArray& G = g->borrow();
        const int nd = (int)x->shape().size() - 2;
        bool assign = false;
        if (dx) {
            HipArray& d = dx->borrow_first_write(assign);
            if (!padding.empty())  // ConvolutionBackwardInput + PadBackward in one kernel
                check((assign ? nk_conv_bwd_input_padded_assign : nk_conv_bwd_input_padded)(
                    D(x), nd, d.ptr(), d.shape().data(), padding.data(), G.ptr(), w->ptr(), w->shape().data(), stride.data(),
                    dilation.data(), groups));
            else
                check((assign ? nk_conv_bwd_input_assign : nk_conv_bwd_input)(D(x), nd, d.ptr(), x->shape().data(), G.ptr(), w->ptr(),
                                                                              w->shape().data(), stride.data(), dilation.data(), groups));
        }
        if (dw && x_unpadded) {  // the Pad node folded into the kernel-gradient pass (db optional)
            bool assign_b = false;
            HipArray& d = dw->borrow_first_write(assign);
            HipArray* b = db ? &db->borrow_first_write(assign_b) : nullptr;
            check(nk_conv_bwd_kernel_bias_padded(D(x), nd, d.ptr(), b ? b->ptr() : nullptr, w->shape().data(), G.ptr(), x->ptr(), x->shape().data(),
                                                 padding.data(), stride.data(), dilation.data(), groups, assign ? 1 : 0, assign_b ? 1 : 0));
        } else if (dw && db) {  // kernel and bias gradient of the fused module node in one pass over G
            bool assign_b = false;
            HipArray& d = dw->borrow_first_write(assign);
            HipArray& b = db->borrow_first_write(assign_b);
            check(nk_conv_bwd_kernel_bias(D(x), nd, d.ptr(), b.ptr(), w->shape().data(), G.ptr(), x->ptr(), x->shape().data(), stride.data(),
                                          dilation.data(), groups, assign ? 1 : 0, assign_b ? 1 : 0));
        } else if (dw) {
            HipArray& d = dw->borrow_first_write(assign);
            check((assign ? nk_conv_bwd_kernel_assign : nk_conv_bwd_kernel)(D(x), nd, d.ptr(), w->shape().data(), G.ptr(), x->ptr(),
                                                                            x->shape().data(), stride.data(), dilation.data(), groups));
        } else if (db) {  // AdditionBackwardRight of the bias: sum of G over every axis the (Cout,1,..) bias lacks
            HipArray& d = db->borrow_first_write(assign);
            check((assign ? nk_unbroadcast_assign : nk_unbroadcast_add)(D(x), d.ptr(), d.shape().data(), (int)d.shape().size(), G.ptr(),
                                                                        G.shape().data(), (int)G.shape().size()));
        }
    }
    void targets(std::vector<const Gradient*>& out) const override {
        if (dx) out.push_back(dx.get());
        if (dw) out.push_back(dw.get());
        if (db) out.push_back(db.get());
    }
};

struct PadFwd : Forward {
    Shared<HipArray> x, y;
    std::vector<int> padding;
    PaddingMode mode;
    void forward() const override {
        const int nd = (int)x->shape().size() - 2;
        switch (mode.kind) {
            case PaddingMode::Reflective:
                check(nk_pad_reflective_fwd(D(x), nd, x->ptr(), x->shape().data(), y->ptr(), padding.data()));
                break;
            case PaddingMode::Replicative:
                check(nk_pad_replicative_fwd(D(x), nd, x->ptr(), x->shape().data(), y->ptr(), padding.data()));
                break;
            default:
                check(nk_pad_const_fwd(D(x), nd, x->ptr(), x->shape().data(), y->ptr(), padding.data(), mode.value));
        }
    }
};
struct PadBwd : Backward {
    Shared<Gradient> dx, g;
    std::vector<int> padding;
    void backward() const override {
        bool assign = false;
        HipArray& d = dx->borrow_first_write(assign);
        check((assign ? nk_pad_bwd_assign : nk_pad_bwd)(d.device()->raw(), (int)d.shape().size() - 2, d.ptr(), d.shape().data(),
                                                        g->borrow().ptr(), padding.data()));
    }
    void targets(std::vector<const Gradient*>& out) const override { out.push_back(dx.get()); }
};

struct DropoutFwd : Forward {  // node/dropout/mod.rs:17-79
    Shared<HipArray> x, y, noise;
    double p;
    Shared<bool> status;
    uint64_t seed;
    Shared<uint64_t> calls;  // Philox offset advances on every forward (noise is resampled)
    void forward() const override {
        const uint64_t offset = (*calls) * ((x->len() + 7) / 8);  // 8 draws per Philox call (nk_common.h)
        check(nk_dropout_fwd(D(x), x->ptr(), y->ptr(), noise->ptr(), x->len(), p, *status ? 1 : 0, seed, offset));
        ++(*calls);  // only a forward that was issued consumes its Philox range (a refused capture throws above)
    }
};
struct DropoutBwd : Backward {
    Shared<Gradient> dx, g;
    Shared<HipArray> noise;
    double p;
    Shared<bool> status;
    void backward() const override {
        bool assign = false;
        HipArray& d = dx->borrow_first_write(assign);
        check((assign ? nk_dropout_bwd_assign : nk_dropout_bwd)(d.device()->raw(), d.ptr(), g->borrow().ptr(), noise->ptr(), d.len(), p,
                                                                *status ? 1 : 0));
    }
    void targets(std::vector<const Gradient*>& out) const override { out.push_back(dx.get()); }
};

// Fused attention probabilities: Multiplication(scalar) + Softmax(last axis) + Dropout
struct AttnProbsFwd : Forward {
    Shared<HipArray> x, probs, out;
    float scale;
    double p;
    Shared<bool> status;
    uint64_t seed;
    Shared<uint64_t> calls, last_offset;  // the backward node regenerates the mask of the LAST forward
    void forward() const override {
        const int L = x->shape().back();
        const long long rows = (long long)(x->len() / (size_t)L);
        const uint64_t offset = (*calls) * ((x->len() + 7) / 8);  // 8 draws per Philox call (nk_common.h)
        *last_offset = offset;
        check(nk_scale_softmax_dropout_fwd(D(x), x->ptr(), probs ? probs->ptr() : nullptr, out->ptr(), nullptr, rows, L, scale, p,
                                           *status ? 1 : 0, seed, offset));
        ++(*calls);  // only a forward that was issued consumes its Philox range (a refused capture throws above)
    }
};
struct AttnProbsBwd : Backward {
    Shared<Gradient> dx, g;
    Shared<HipArray> probs, scores;  // probs null: recomputed from the scores (the forward did not store them)
    float scale;
    double p;
    Shared<bool> status;
    uint64_t seed;
    Shared<uint64_t> last_offset;
    void backward() const override {
        bool assign = false;
        HipArray& d = dx->borrow_first_write(assign);
        const int L = d.shape().back();
        const long long rows = (long long)(d.len() / (size_t)L);
        if (!probs)
            check(nk_scale_softmax_dropout_bwd_from_scores(d.device()->raw(), d.ptr(), g->borrow().ptr(), scores->ptr(), nullptr, rows, L,
                                                           scale, p, *status ? 1 : 0, seed, *last_offset, assign ? 1 : 0));
        else
            check((assign ? nk_scale_softmax_dropout_bwd_assign : nk_scale_softmax_dropout_bwd)(d.device()->raw(), d.ptr(), g->borrow().ptr(), probs->ptr(), nullptr,
                                               rows, L, scale, p, *status ? 1 : 0, seed, *last_offset));
    }
    void targets(std::vector<const Gradient*>& out) const override { out.push_back(dx.get()); }
};

// Pointwise unary nodes: node/{negation,exp,logn,sqrt,sigmoid,tanh,softplus,leaky_relu,power}
struct PointwiseFwd : Forward {
    int op, iparam;
    Shared<HipArray> x, y;
    void forward() const override { check(nk_unary_fwd(D(x), op, x->ptr(), y->ptr(), x->len(), iparam)); }
};
struct PointwiseBwd : Backward {
    int op, iparam;
    Shared<Gradient> dx, g;
    Shared<HipArray> ref;  // the buffer the reference node keeps: operand data or node data
    void backward() const override {
        bool assign = false;
        HipArray& d = dx->borrow_first_write(assign);
        check((assign ? nk_unary_bwd_assign : nk_unary_bwd)(d.device()->raw(), op, d.ptr(), g->borrow().ptr(), ref ? ref->ptr() : nullptr,
                                                            d.len(), iparam));
    }
    void targets(std::vector<const Gradient*>& out) const override { out.push_back(dx.get()); }
};
// Unsqueeze: node/unsqueeze/mod.rs:35-40 (copy into the reshaped buffer), :70-78 (dx += g)
struct UnsqueezeFwd : Forward {
    Shared<HipArray> x, y;
    void forward() const override { check(nk_copy(D(x), y->ptr(), x->ptr(), x->len())); }
};
struct UnsqueezeBwd : Backward {
    Shared<Gradient> dx, g;
    void backward() const override {
        HipArray& d = dx->borrow();
        const int n = (int)d.len();
        check(nk_unbroadcast_add(d.device()->raw(), d.ptr(), &n, 1, g->borrow().ptr(), &n, 1));
    }
    void targets(std::vector<const Gradient*>& out) const override { out.push_back(dx.get()); }
};

struct ChunkFwd : Forward {
    Shared<HipArray> x, y;
    int chunk_no;
    void forward() const override {
        check(nk_chunk_fwd(D(x), x->ptr(), x->shape().data(), y->ptr(), y->shape().data(), (int)x->shape().size(), chunk_no));
    }
};
struct ChunkBwd : Backward {
    Shared<Gradient> dx, g;
    int chunk_no;
    void backward() const override {
        HipArray& d = dx->borrow();
        const HipArray& G = g->borrow();
        check(nk_chunk_bwd(d.device()->raw(), d.ptr(), d.shape().data(), G.ptr(), G.shape().data(), (int)d.shape().size(), chunk_no));
    }
    void targets(std::vector<const Gradient*>& out) const override { out.push_back(dx.get()); }
};

struct CatFwd : Forward {  // node/multi_concatenate/mod.rs
    std::vector<Shared<HipArray>> operands;
    Shared<HipArray> out;
    int axis;
    bool stack = false;  // node/multi_stack/mod.rs:37-47: every operand fills one index of a NEW axis
    void forward() const override {
        int off = 0;
        for (const auto& o : operands) {
            const int len = stack ? 1 : o->shape()[axis];
            check(nk_concat_fwd_part(D(out), o->ptr(), out->ptr(), out->shape().data(), (int)out->shape().size(), axis, off, len));
            off += len;
        }
    }
};
struct CatBwd : Backward {
    std::vector<Shared<Gradient>> operands;
    Shared<Gradient> g;
    int axis;
    bool stack = false;  // node/multi_stack/mod.rs:76-86
    void backward() const override {
        const HipArray& G = g->borrow();
        int off = 0;
        for (const auto& o : operands) {
            bool assign = false;
            HipArray& d = o->borrow_first_write(assign);
            const int len = stack ? 1 : d.shape()[axis];
            check((assign ? nk_concat_bwd_part_assign : nk_concat_bwd_part)(d.device()->raw(), d.ptr(), G.ptr(), G.shape().data(), (int)G.shape().size(), axis, off, len));
            off += len;
        }
    }
    void targets(std::vector<const Gradient*>& out) const override {
        for (const auto& o : operands) out.push_back(o.get());
    }
};

struct HeadsFwd : Forward {
    bool split;
    Shared<HipArray> x, y;
    int B, S, H, dh;
    void forward() const override {
        check(split ? nk_split_heads_fwd(D(x), x->ptr(), y->ptr(), B, S, H, dh) : nk_merge_heads_fwd(D(x), x->ptr(), y->ptr(), B, S, H, dh));
    }
};
struct HeadsBwd : Backward {
    bool split;
    Shared<Gradient> dx, g;
    int B, S, H, dh;
    void backward() const override {
        bool assign = false;
        HipArray& d = dx->borrow_first_write(assign);
        auto fn = split ? (assign ? nk_split_heads_bwd_assign : nk_split_heads_bwd)
                        : (assign ? nk_merge_heads_bwd_assign : nk_merge_heads_bwd);
        check(fn(d.device()->raw(), d.ptr(), g->borrow().ptr(), B, S, H, dh));
    }
    void targets(std::vector<const Gradient*>& out) const override { out.push_back(dx.get()); }
};

struct MseFwd : Forward {  // node/squared_error/mod.rs:42-59
    Shared<HipArray> x, t, out;
    Reduction red;
    void forward() const override { check(nk_mse_fwd(D(x), x->ptr(), t->ptr(), x->len(), (int)red, out->ptr())); }
};
struct MseBwd : Backward {
    Shared<HipArray> x, t;
    Shared<Gradient> dx, g;
    Reduction red;
    void backward() const override {
        bool assign = false;
        HipArray& d = dx->borrow_first_write(assign);
        check((assign ? nk_mse_bwd_assign : nk_mse_bwd)(D(x), d.ptr(), g->borrow().ptr(), x->ptr(), t->ptr(), x->len(), (int)red));
    }
    void targets(std::vector<const Gradient*>& out) const override { out.push_back(dx.get()); }
};

// Per-(batch, head) attention products on the (B*S, H*dh) projection layout: the B*H Chunk((S,dh)) tiles of the composed
// module (SURVEY 8a note) are addressed as strided GEMM operands instead of being copied out and back.
struct HeadsGeom {
    int B, S, H, dh;
    int d() const { return H * dh; }
    long long so() const { return (long long)S * H * dh; }  // sample stride in the flat layout
    long long po() const { return (long long)H * S * S; }   // sample stride of the (B*H, S, S) tensor
    long long pi() const { return (long long)S * S; }
};
struct HeadsScoresFwd : Forward {  // C_bh(S,S) = Q_bh(S,dh) . K_bh(S,dh)^T
    HeadsGeom hg;
    Shared<HipArray> q, k, c;
    void forward() const override {
        check(nk_sgemm_batched(D(q), 0, 1, hg.S, hg.S, hg.dh, 1.f, q->ptr(), hg.d(), hg.so(), hg.dh, k->ptr(), hg.d(), hg.so(), hg.dh, 0.f,
                               c->ptr(), hg.S, hg.po(), hg.pi(), hg.B, hg.H));
    }
};
struct HeadsScoresBwd : Backward {
    HeadsGeom hg;
    Shared<HipArray> q, k;
    Shared<Gradient> dq, dk, g;
    void backward() const override {
        const HipArray& G = g->borrow();
        nk_device* dev = D(q);
        float beta;
        {   // dQ_bh += G_bh . K_bh      (every element of dQ belongs to exactly one head block: a first write may assign)
            float* d = first_write(dq, beta);
            check(nk_sgemm_batched(dev, 0, 0, hg.S, hg.dh, hg.S, 1.f, G.ptr(), hg.S, hg.po(), hg.pi(), k->ptr(), hg.d(), hg.so(), hg.dh, beta,
                                   d, hg.d(), hg.so(), hg.dh, hg.B, hg.H));
        }
        {   // dK_bh += G_bh^T . Q_bh
            float* d = first_write(dk, beta);
            check(nk_sgemm_batched(dev, 1, 0, hg.S, hg.dh, hg.S, 1.f, G.ptr(), hg.S, hg.po(), hg.pi(), q->ptr(), hg.d(), hg.so(), hg.dh, beta,
                                   d, hg.d(), hg.so(), hg.dh, hg.B, hg.H));
        }
    }
    void targets(std::vector<const Gradient*>& out) const override { out.push_back(dq.get()); out.push_back(dk.get()); }
};
struct HeadsContextFwd : Forward {  // O_bh(S,dh) = P_bh(S,S) . V_bh(S,dh), written into the flat layout
    HeadsGeom hg;
    Shared<HipArray> p, v, o;
    void forward() const override {
        check(nk_sgemm_batched(D(p), 0, 0, hg.S, hg.dh, hg.S, 1.f, p->ptr(), hg.S, hg.po(), hg.pi(), v->ptr(), hg.d(), hg.so(), hg.dh, 0.f,
                               o->ptr(), hg.d(), hg.so(), hg.dh, hg.B, hg.H));
    }
};
struct HeadsContextBwd : Backward {
    HeadsGeom hg;
    Shared<HipArray> p, v;
    Shared<Gradient> dp, dv, g;
    void backward() const override {
        const HipArray& G = g->borrow();  // dO, flat layout
        nk_device* dev = D(p);
        float beta;
        {   // dP_bh += dO_bh . V_bh^T
            float* d = first_write(dp, beta);
            check(nk_sgemm_batched(dev, 0, 1, hg.S, hg.S, hg.dh, 1.f, G.ptr(), hg.d(), hg.so(), hg.dh, v->ptr(), hg.d(), hg.so(), hg.dh, beta,
                                   d, hg.S, hg.po(), hg.pi(), hg.B, hg.H));
        }
        {   // dV_bh += P_bh^T . dO_bh
            float* d = first_write(dv, beta);
            check(nk_sgemm_batched(dev, 1, 0, hg.S, hg.dh, hg.S, 1.f, p->ptr(), hg.S, hg.po(), hg.pi(), G.ptr(), hg.d(), hg.so(), hg.dh, beta,
                                   d, hg.d(), hg.so(), hg.dh, hg.B, hg.H));
        }
    }
    void targets(std::vector<const Gradient*>& out) const override { out.push_back(dp.get()); out.push_back(dv.get()); }
};

// heads_scores -> attention_probs -> heads_context as one node (nk_attention_fwd / nk_attention_bwd)
struct HeadsAttentionFwd : Forward {
    HeadsGeom hg;
    Shared<HipArray> q, k, v, scores, stats, mask, o;  // mask: the dropout draws, 1 bit per score (u32 words in an f32 array)
    float scale;
    double p;
    Shared<bool> status;
    uint64_t seed;
    Shared<uint64_t> calls;  // each forward draws a fresh mask (the Philox offset advances), as AttnProbsFwd
    void forward() const override {
        const uint64_t sp = ((uint64_t)hg.S + 31) / 32 * 32;  // the draws are indexed in the padded (B*H, SP, SP) tensor (SP = S unless S is ragged)
        const uint64_t offset = (*calls) * (((uint64_t)hg.B * hg.H * sp * sp + 7) / 8);  // draws per forward, 8 per Philox call
        // scores / stats / mask are null in a graph without gradients: nothing is kept, no (B*H, S, S) tensor exists
        check(nk_attention_fwd(D(q), q->ptr(), k->ptr(), v->ptr(), scores ? scores->ptr() : nullptr, stats ? stats->ptr() : nullptr,
                               mask ? reinterpret_cast<uint32_t*>(mask->ptr()) : nullptr, o->ptr(), hg.B, hg.S, hg.H, hg.dh, scale, p,
                               *status ? 1 : 0, seed, offset));
        ++(*calls);  // only a forward that was issued consumes its Philox range (a refused capture throws above)
    }
};
struct HeadsAttentionBwd : Backward {
    HeadsGeom hg;
    Shared<HipArray> q, k, v, scores, stats, mask, o;
    Shared<HipArray> ds, dropped;  // (B*H, S, S) scratch written by the first kernel, read by the dK / dV products
    Shared<Gradient> dq, dk, dv, g;
    float scale;
    double p;
    Shared<bool> status;
    void backward() const override {
        const HipArray& G = g->borrow();  // dO, flat layout
        nk_device* dev = D(q);
        float bq, bk, bv;
        float* gq = first_write(dq, bq);
        float* gk = first_write(dk, bk);
        float* gv = first_write(dv, bv);
        // dS and Pd are written by the fused kernel, dQ comes out of it; dK / dV are the two batched products on dS / Pd
        check(nk_attention_bwd(dev, gq, gk, gv, ds->ptr(), dropped->ptr(), G.ptr(), o->ptr(), scores->ptr(), stats->ptr(),
                               reinterpret_cast<const uint32_t*>(mask->ptr()), q->ptr(), k->ptr(), v->ptr(), hg.B, hg.S, hg.H, hg.dh, scale,
                               p, *status ? 1 : 0, bq == 0.f ? 1 : 0, bk == 0.f ? 1 : 0, bv == 0.f ? 1 : 0));
    }
    void targets(std::vector<const Gradient*>& out) const override {
        out.push_back(dq.get()); out.push_back(dk.get()); out.push_back(dv.get());
    }
};

// nn::MultiheadAttention with PACKED projections: x -> [Q | K | V] = x . [Wq; Wk; Wv]^T + [bq; bk; bv] (ONE GEMM, N = 3d) -> the
// fused attention core reading Q, K, V as column blocks of that output -> context.  One forward and one backward node for what
// the unpacked module runs as three Linear nodes + the attention node (same values bit for bit in the forward; the backward's
// input gradient is one K = 3d chain instead of three K = d chains added up).
struct QkvAttentionFwd : Forward {
    HeadsGeom hg;
    Shared<HipArray> x, w, b, qkv, scores, stats, mask, o;  // w (3d, d), b (3d): the packed storage; qkv (B*S, 3d)
    float scale;
    double p;
    Shared<bool> status;
    uint64_t seed;
    Shared<uint64_t> calls;
    void forward() const override {
        const int n = hg.B * hg.S, d = hg.d();
        check(nk_linear_fwd(D(x), x->ptr(), w->ptr(), b->ptr(), qkv->ptr(), n, d, 3 * d));
        const uint64_t sp = ((uint64_t)hg.S + 31) / 32 * 32;
        const uint64_t offset = (*calls) * (((uint64_t)hg.B * hg.H * sp * sp + 7) / 8);
        check(nk_attention_qkv_fwd(D(x), qkv->ptr(), scores ? scores->ptr() : nullptr, stats ? stats->ptr() : nullptr,
                                   mask ? reinterpret_cast<uint32_t*>(mask->ptr()) : nullptr, o->ptr(), hg.B, hg.S, hg.H, hg.dh, scale, p,
                                   *status ? 1 : 0, seed, offset));
        ++(*calls);
    }
};
struct QkvAttentionBwd : Backward {
    HeadsGeom hg;
    Shared<HipArray> x, w, qkv, scores, stats, mask, o;
    Shared<HipArray> dqkv, ds, dropped;  // scratch owned by the node: (B*S, 3d) and two (B*H, SP, SP)
    Shared<HipArray> gw_all, gb_all;     // packed gradient storage: (3d, d), (3d)
    Shared<Gradient> dx, gw[3], gb[3], g;  // dx null for a non-differentiable input; gw / gb: views of gw_all / gb_all
    float scale;
    double p;
    Shared<bool> status;
    // the three views are written by ONE kernel: it may assign only when all three still wait for their zero fill
    static float packed_beta(const Shared<Gradient> (&v)[3]) {
        bool all = true;
        for (const auto& q : v) all = all && q->zero_pending();
        bool assign = false;
        for (const auto& q : v) {
            if (all) (void)q->borrow_first_write(assign);  // hands the pending fill to the kernel below (beta = 0)
            else (void)q->borrow();                        // materialises a pending view, the kernel accumulates
        }
        return all ? 0.f : 1.f;
    }
    void backward() const override {
        const HipArray& G = g->borrow();  // dO (B*S, d)
        nk_device* dev = D(x);
        const int n = hg.B * hg.S, d = hg.d();
        check(nk_attention_qkv_bwd(dev, dqkv->ptr(), ds->ptr(), dropped->ptr(), G.ptr(), o->ptr(), scores->ptr(), stats->ptr(),
                                   reinterpret_cast<const uint32_t*>(mask->ptr()), qkv->ptr(), hg.B, hg.S, hg.H, hg.dh, scale, p,
                                   *status ? 1 : 0, 1));
        float beta;
        if (dx) {  // dX (+)= [dQ | dK | dV] . [Wq; Wk; Wv]: one NN product, K = 3d
            float* q = first_write(dx, beta);
            check(nk_sgemm(dev, 0, 0, n, d, 3 * d, 1.f, dqkv->ptr(), 3 * d, w->ptr(), d, beta, q, d));
        }
        {
            const int gs[2] = {n, 3 * d}, o3 = 3 * d;
            const bool assign = packed_beta(gb) == 0.f;
            check((assign ? nk_unbroadcast_assign : nk_unbroadcast_add)(dev, gb_all->ptr(), &o3, 1, dqkv->ptr(), gs, 2));
        }
        beta = packed_beta(gw);  // [dWq; dWk; dWv] (+)= [dQ | dK | dV]^T . x: one TN product, M = 3d
        check(nk_sgemm(dev, 1, 0, 3 * d, d, n, 1.f, dqkv->ptr(), 3 * d, x->ptr(), d, beta, gw_all->ptr(), d));
    }
    void targets(std::vector<const Gradient*>& out) const override {
        if (dx) out.push_back(dx.get());
        for (const auto& q : gw) out.push_back(q.get());
        for (const auto& q : gb) out.push_back(q.get());
    }
};

// Scalar criteria: node/{absolute_error,bce,bce_with_logits,kldiv,nll}/mod.rs.  kind -1 = NLL.
struct LossFwd : Forward {
    int kind;
    Shared<HipArray> x, t, out;
    Reduction red;
    void forward() const override {
        const Shape& s = x->shape();
        if (kind < 0) check(nk_nll_fwd(D(x), x->ptr(), t->ptr(), s.data(), (int)s.size(), (int)red, out->ptr()));
        else check(nk_loss_fwd(D(x), kind, x->ptr(), t->ptr(), s.data(), (int)s.size(), (int)red, out->ptr()));
    }
};
struct LossBwd : Backward {
    int kind;
    Shared<HipArray> x, t;
    Shared<Gradient> dx, g;
    Reduction red;
    void backward() const override {
        HipArray& d = dx->borrow();
        const Shape& s = d.shape();
        if (kind < 0) check(nk_nll_bwd(D(x), d.ptr(), g->borrow().ptr(), t->ptr(), s.data(), (int)s.size(), (int)red));
        else check(nk_loss_bwd(D(x), kind, d.ptr(), g->borrow().ptr(), x->ptr(), t->ptr(), s.data(), (int)s.size(), (int)red));
    }
    void targets(std::vector<const Gradient*>& out) const override { out.push_back(dx.get()); }
};

// `Linear::forward` as one node: mm_t + broadcast Addition (neuronika-nn/src/lib.rs:443-446); `relu`: followed by the ReLU
// node (node/relu/mod.rs:29-38) in the same epilogue
struct LinearFwd : Forward {
    Shared<HipArray> x, w, b, y;
    bool relu = false;
    void forward() const override {
        check((relu ? nk_linear_relu_fwd : nk_linear_fwd)(D(x), x->ptr(), w->ptr(), b->ptr(), y->ptr(), x->shape()[0], x->shape()[1], w->shape()[0]));
    }
};
struct LinearBwd : Backward {
    Shared<HipArray> x, w;
    Shared<HipArray> y;              // set for the fused Linear+ReLU node: its output, the mask of its own gradient
    Shared<Gradient> dx, dw, db, g;  // dx null for a non-differentiable input
    mutable Shared<HipArray> masked_;  // (y > 0) * dL/dy of a pass whose writers stored plain values; owned by the node (see below)
    void backward() const override {
        // Linear+ReLU: g must hold dL/dz = (y > 0) * dL/dy.  When every writer of g on this tape applied the mask while
        // storing (`premasked`, decided by VarDiff::run_backward), it already does; otherwise mask it in place now
        // into a scratch copy (the traffic of an in-place pass): g itself keeps dL/dy, what `grad()` of this variable shows
        // in the reference's graph - also for a root, whose gradient is the seed
        // The scratch belongs to the NODE: allocated the first time a pass needs it and kept for the node's life, so that a pass
        // never allocates after the first one (a hipGraph captured from a warm step bakes in a pointer that stays this node's -
        // the pool cannot hand it to another tensor between replays).
        const bool mask_now = y && !g->premasked();
        if (mask_now) {
            const HipArray& Gy = g->borrow();
            if (!masked_ || masked_->shape() != Gy.shape()) masked_ = std::make_shared<HipArray>(x->device(), Gy.shape(), HipArray::Uninit{});
            check(nk_relu_bwd_assign(D(x), masked_->ptr(), Gy.ptr(), y->ptr(), Gy.len()));
        }
        const HipArray& G = mask_now ? *masked_ : g->borrow();
        nk_device* dev = D(x);
        const int n = x->shape()[0], m = x->shape()[1], o = w->shape()[0];
        // MatrixMatrixMulTBackward (left, right) and AdditionBackwardRight write three different buffers, so their
        // order is free: the bias gradient goes first (the data-parallel exchange then sends the small gradients of
        // the whole model as one group while the last weight-gradient GEMMs still run), the weight gradient last
        float beta;  // nk_mm_t_bwd_left / nk_mm_t_bwd_right, with beta 0 when the gradient's zero fill is still pending
        if (dx) {
            float* d = first_write(dx, beta);
            if (dx->premasked())  // x is the output of a Linear+ReLU node: store (x > 0) * (G . W), its pre-activation gradient
                check(nk_linear_bwd_input_relu(dev, d, G.ptr(), w->ptr(), x->ptr(), n, m, o, beta == 0.f ? 1 : 0));
            else
                check(nk_sgemm(dev, 0, 0, n, m, o, 1.f, G.ptr(), o, w->ptr(), m, beta, d, m));
        }
        {
            // (the bias gradient goes first and on its own: under the data-parallel hook the small gradients of the whole model
            //  leave as one group as soon as the last of them is final - in front of this layer's weight-gradient GEMMs.
            //  Summing it on the way in the weight-gradient GEMM - its A operand is G - was built and measured in round 3:
            //  the 16 adds per k-tile cost the TN GEMM 4 %, more than the 12 us reduction they replace)
            const int gs[2] = {n, o};
            bool assign = false;
            HipArray& d = db->borrow_first_write(assign);
            check((assign ? nk_unbroadcast_assign : nk_unbroadcast_add)(dev, d.ptr(), &o, 1, G.ptr(), gs, 2));
            if (BackwardHook* hook = parts_hook(db.get())) hook->grad_part_ready(db.get(), 0, (size_t)o);
        }
        {
            float* d = first_write(dw, beta);
            BackwardHook* hook = parts_hook(dw.get());  // null unless this node is the last writer of dW and the hook wants pieces
            // data-parallel exchange at row-block granularity (only for the weight gradient the hook asks for - by default
            // the one that becomes final last): each half of dW (still >= 512 tiles of 128x128, a full wave of resident
            // blocks) goes to the all-reduce as soon as it is issued, so only half a gradient's exchange is left exposed
            // behind the last GEMM of the backward pass
            const int h = o / 2;
            if (hook && o % 256 == 0 && (long long)(h / 128) * ((m + 127) / 128) >= 512) {
                for (int r0 = 0; r0 < o; r0 += h) {
                    check(nk_sgemm(dev, 1, 0, h, m, n, 1.f, G.ptr() + r0, o, x->ptr(), m, beta, d + (size_t)r0 * m, m));
                    hook->grad_part_ready(dw.get(), (size_t)r0 * m, (size_t)h * m);
                }
            } else {
                check(nk_sgemm(dev, 1, 0, o, m, n, 1.f, G.ptr(), o, x->ptr(), m, beta, d, m));
            }
        }
    }
    void targets(std::vector<const Gradient*>& out) const override {
        if (dx) out.push_back(dx.get());
        out.push_back(dw.get());
        out.push_back(db.get());
    }
    void premask_targets(std::vector<const Gradient*>& out) const override {
        // only when the mask source IS this node's input buffer (the gradient belongs to the node that produced x)
        if (dx && dx->premask_source() && dx->premask_source().get() == x.get()) out.push_back(dx.get());
    }
};

template <class Op>
BackwardEntry entry(Shared<Op> op, Shared<Gradient> grad) {
    return BackwardEntry{std::move(op), std::move(grad)};
}

Shape mm_shape(const Shape& a, const Shape& b, int kind) {  // utils.rs:46-55 `DotDim`
    if (kind == 4) {
        if (a.size() != 2 || b.size() != 1) panic("mv: matrix and vector expected");
        if (a[1] != b[0]) panic("Shapes are incompatible for matrix-vector multiplication.");
        return Shape{a[0]};
    }
    if (kind == 5) {
        if (a.size() != 1 || b.size() != 2) panic("vm: vector and matrix expected");
        if (a[0] != b[0]) panic("Shapes are incompatible for vector-matrix multiplication.");
        return Shape{b[1]};
    }
    if (kind == 6) {
        if (a.size() != 1 || b.size() != 1) panic("vv: vectors expected");
        if (a[0] != b[0]) panic("Shapes are incompatible for vector-vector multiplication.");
        return Shape{};
    }
    if (kind <= 1) {
        if (a.size() != 2 || b.size() != 2) panic("mm: matrices expected");
        const int inner_b = kind == 0 ? b[0] : b[1];
        if (a[1] != inner_b) panic("Shapes are incompatible for matrix multiplication.");
        return Shape{a[0], kind == 0 ? b[1] : b[0]};
    }
    if (a.size() != 3 || b.size() != 3 || a[0] != b[0]) panic("bmm: [B,n,m] x [B,*,*] expected");
    const int inner_b = kind == 2 ? b[1] : b[2];
    if (a[2] != inner_b) panic("Shapes are incompatible for matrix multiplication.");
    return Shape{a[0], a[1], kind == 2 ? b[2] : b[1]};
}

Var matmul_var(int kind, const Var& a, const Var& b) {
    History<ForwardEntry> h = a.history;
    h.merge(b.history);
    auto out = zeros_like(a.data, mm_shape(a.shape(), b.shape(), kind));
    return Var::node(out, std::make_shared<MatMulFwd>(kind, a.data, b.data, out), std::move(h));
}
VarDiff matmul_diff(int kind, const Var& a, const Shared<Gradient>& da, const History<BackwardEntry>* ha, const Var& b,
                    const Shared<Gradient>& db, const History<BackwardEntry>* hb) {
    Var var = matmul_var(kind, a, b);
    History<BackwardEntry> h;
    if (ha) h = *ha;
    if (hb) h.merge(*hb);
    auto grad = std::make_shared<Gradient>(var.device(), var.shape());
    auto op = std::make_shared<MatMulBwd>();
    op->kind = kind; op->a = a.data; op->b = b.data; op->da = da; op->db = db; op->g = grad;
    return VarDiff::node(std::move(var), grad, entry(op, grad), std::move(h));
}

Var binary_var(int op, const Var& l, const Var& r) {
    History<ForwardEntry> h = l.history;
    h.merge(r.history);
    auto out = zeros_like(l.data, cobroadcast(l.shape(), r.shape()));
    return Var::node(out, std::make_shared<BinaryFwd>(op, l.data, r.data, out), std::move(h));
}
VarDiff binary_diff(int op, const Var& l, const Shared<Gradient>& lg, const History<BackwardEntry>* lh, const Var& r,
                    const Shared<Gradient>& rg, const History<BackwardEntry>* rh) {
    Var var = binary_var(op, l, r);
    History<BackwardEntry> h;
    if (lh) h = *lh;
    if (rh) h.merge(*rh);
    auto grad = std::make_shared<Gradient>(var.device(), var.shape());
    auto bw = std::make_shared<BinaryBwd>();
    bw->op = op; bw->lg = lg; bw->rg = rg; bw->g = grad; bw->l = l.data; bw->r = r.data;
    return VarDiff::node(std::move(var), grad, entry(bw, grad), std::move(h));
}

Var unary_var(Unary k, int axis, const Var& x, Shape out_shape) {
    auto y = zeros_like(x.data, std::move(out_shape));
    return Var::node(y, std::make_shared<UnaryFwd>(k, axis, x.data, y), x.history);
}
VarDiff unary_diff(Unary k, int axis, const VarDiff& x, Shape out_shape) {
    Var var = unary_var(k, axis, x.var, std::move(out_shape));
    auto grad = std::make_shared<Gradient>(var.device(), var.shape());
    auto bw = std::make_shared<UnaryBwd>();
    bw->kind = k; bw->axis = axis; bw->dx = x.grad; bw->g = grad; bw->x = x.var.data; bw->y = var.data;
    return VarDiff::node(std::move(var), grad, entry(bw, grad), x.history);
}

void check_axis(const Shape& s, int axis) {
    if (axis < 0 || axis >= (int)s.size()) panic("axis out of bounds");
}

Shape conv_out_shape(const Shape& in, const Shape& k, const std::vector<int>& stride, const std::vector<int>& dil, int groups) {
    // check_conv_args / check_groups_args / conv_out_shape  utils.rs:207-237, 427-497
    const int nd = (int)in.size() - 2;
    if ((int)stride.size() != nd) panic("Invalid stride for " + std::to_string(nd) + "d conv.");
    if ((int)dil.size() != nd) panic("Invalid dilation for " + std::to_string(nd) + "d conv.");
    if (k.size() != in.size()) panic("Invalid kernel shape for " + std::to_string(nd) + "d conv");
    if (in[1] % groups != 0) panic("In channels " + std::to_string(in[1]) + " is not divisible by groups " + std::to_string(groups));
    if (k[0] % groups != 0) panic("Out channels " + std::to_string(k[0]) + " is not divisible by groups " + std::to_string(groups));
    Shape out{in[0], k[0]};
    for (int d = 0; d < nd; ++d) {
        if (in[2 + d] < (k[2 + d] - 1) * dil[d] + 1) panic("The kernel size can't be greater than actual input size.");
        out.push_back((in[2 + d] - dil[d] * (k[2 + d] - 1) - 1) / stride[d] + 1);
    }
    return out;
}

}  // namespace

// =================================================================================================
// Var
// =================================================================================================
Var Var::leaf(Shared<HipArray> array) {
    Var v;
    v.data = std::move(array);
    return v;
}
Var Var::node(Shared<HipArray> data, Shared<Forward> op, History<ForwardEntry> h) {
    const void* ptr = op.get();
    h.insert(ptr, ForwardEntry{std::move(op), std::make_shared<bool>(false)});
    Var v;
    v.data = std::move(data);
    v.history = std::move(h);
    return v;
}
VarDiff Var::requires_grad() const { return VarDiff::leaf(*this, std::make_shared<Gradient>(device(), shape())); }
void Var::forward() const {
    auto& buffer = history.buffer_mut();
    if (buffer.empty()) buffer = history.to_vec();
    else
        for (auto& e : buffer) *e.computed = false;
    for (auto& e : buffer)
        if (!*e.computed) {
            e.op->forward();
            *e.computed = true;
        }
}
float Var::item() const {
    if (data->len() != 1) panic("item(): not a scalar");
    float v;
    data->download(&v);
    return v;
}
Var Var::sum() const { return unary_var(Unary::Sum, 0, *this, {}); }
Var Var::mean() const { return unary_var(Unary::Mean, 0, *this, {}); }
Var Var::relu() const { return unary_var(Unary::Relu, 0, *this, shape()); }
static Var pointwise_var(int op, int iparam, const Var& x) {
    auto n = std::make_shared<PointwiseFwd>();
    n->op = op; n->iparam = iparam; n->x = x.data; n->y = zeros_like(x.data, x.shape());
    auto y = n->y;
    return Var::node(y, n, x.history);
}
static VarDiff pointwise_diff(int op, int iparam, const VarDiff& x) {
    Var v = pointwise_var(op, iparam, x.var);
    auto g = std::make_shared<Gradient>(v.device(), v.shape());
    auto bw = std::make_shared<PointwiseBwd>();
    bw->op = op; bw->iparam = iparam; bw->dx = x.grad; bw->g = g;
    const bool keeps_output = op == NK_EXP || op == NK_SQRT || op == NK_SIGMOID || op == NK_TANH;
    if (op != NK_NEG) bw->ref = keeps_output ? v.data : x.var.data;
    return VarDiff::node(std::move(v), g, entry(bw, g), x.history);
}
Var Var::neg() const { return pointwise_var(NK_NEG, 0, *this); }
Var Var::pow(int e) const { return pointwise_var(NK_POW, e, *this); }
Var Var::sqrt() const { return pointwise_var(NK_SQRT, 0, *this); }
Var Var::leaky_relu() const { return pointwise_var(NK_LEAKY_RELU, 0, *this); }
Var Var::softplus() const { return pointwise_var(NK_SOFTPLUS, 0, *this); }
Var Var::sigmoid() const { return pointwise_var(NK_SIGMOID, 0, *this); }
Var Var::tanh() const { return pointwise_var(NK_TANH, 0, *this); }
Var Var::ln() const { return pointwise_var(NK_LN, 0, *this); }
Var Var::exp() const { return pointwise_var(NK_EXP, 0, *this); }
Var Var::unsqueeze(int axis) const {
    if (axis < 0 || axis > (int)shape().size()) panic("unsqueeze: axis out of bounds");
    Shape s = shape();
    s.insert(s.begin() + axis, 1);
    auto n = std::make_shared<UnsqueezeFwd>();
    n->x = data; n->y = zeros_like(data, s);
    auto y = n->y;
    return Var::node(y, n, history);
}
Var Var::softmax(int axis) const { check_axis(shape(), axis); return unary_var(Unary::Softmax, axis, *this, shape()); }
Var Var::log_softmax(int axis) const { check_axis(shape(), axis); return unary_var(Unary::LogSoftmax, axis, *this, shape()); }
Var Var::t() const { return unary_var(Unary::Transpose, 0, *this, Shape(shape().rbegin(), shape().rend())); }
Var Var::dropout(double p, Shared<bool> status) const {
    if (!(p >= 0.0 && p <= 1.0)) panic("Wrong probability received: " + std::to_string(p) + ".");
    auto op = std::make_shared<DropoutFwd>();
    op->x = data; op->y = zeros_like(data, shape()); op->noise = zeros_like(data, shape());
    op->p = p; op->status = std::move(status);
    op->seed = next_node_seed();
    op->calls = std::make_shared<uint64_t>(0);
    auto y = op->y;
    return Var::node(y, op, history);
}
std::vector<Var> Var::chunks(const Shape& chunk_size) const {
    if (chunk_size.size() != shape().size()) panic("chunks: rank mismatch");
    size_t n = 1;
    for (size_t i = 0; i < chunk_size.size(); ++i) {
        if (chunk_size[i] <= 0) panic("chunks: chunk size must be positive");
        n *= (size_t)(shape()[i] / chunk_size[i]);
    }
    std::vector<Var> out;
    for (size_t i = 0; i < n; ++i) {
        auto op = std::make_shared<ChunkFwd>();
        op->x = data; op->y = zeros_like(data, chunk_size); op->chunk_no = (int)i;
        auto y = op->y;
        out.push_back(Var::node(y, op, history));
    }
    return out;
}
Var Var::cat(const std::vector<Var>& variables, int axis) const {
    check_axis(shape(), axis);
    History<ForwardEntry> h = history;
    auto op = std::make_shared<CatFwd>();
    op->operands.push_back(data);
    Shape s = shape();
    for (const Var& v : variables) {
        h.merge(v.history);
        op->operands.push_back(v.data);
        for (size_t i = 0; i < s.size(); ++i)
            if ((int)i != axis && v.shape()[i] != s[i]) panic("cat: incompatible shapes");
        s[axis] += v.shape()[axis];
    }
    op->axis = axis;
    op->out = zeros_like(data, s);
    auto y = op->out;
    return Var::node(y, op, std::move(h));
}
Var Var::stack(const std::vector<Var>& variables, int axis) const {
    if (axis < 0 || axis > (int)shape().size()) panic("stack: axis out of bounds");
    History<ForwardEntry> h = history;
    auto op = std::make_shared<CatFwd>();
    op->operands.push_back(data);
    for (const Var& v : variables) {
        if (v.shape() != shape()) panic("stack: all the variables must have the same shape");
        h.merge(v.history);
        op->operands.push_back(v.data);
    }
    Shape s = shape();
    s.insert(s.begin() + axis, (int)variables.size() + 1);
    op->axis = axis; op->stack = true;
    op->out = zeros_like(data, s);
    auto y = op->out;
    return Var::node(y, op, std::move(h));
}
static Var loss_var(int kind, const Var& x, const Var& target, Reduction reduction) {
    if (kind < 0) {  // nll: target has the input's shape without the class axis (var.rs:661)
        Shape want = x.shape();
        if (want.size() < 2) panic("nll: input of shape (minibatch, C, ...) expected");
        want.erase(want.begin() + 1);
        if (target.shape() != want) panic("nll: target must have shape (minibatch, d1, ..., dk)");
    } else if (target.shape() != x.shape()) {
        panic("loss: input and target shapes differ");
    }
    History<ForwardEntry> h = x.history;
    h.merge(target.history);
    auto op = std::make_shared<LossFwd>();
    op->kind = kind; op->x = x.data; op->t = target.data; op->out = zeros_like(x.data, {}); op->red = reduction;
    auto y = op->out;
    return Var::node(y, op, std::move(h));
}
static VarDiff loss_diff(int kind, const VarDiff& x, const Var& target, Reduction reduction) {
    Var out = loss_var(kind, x.var, target, reduction);
    auto g = std::make_shared<Gradient>(x.device(), Shape{});
    auto bw = std::make_shared<LossBwd>();
    bw->kind = kind; bw->x = x.var.data; bw->t = target.data; bw->dx = x.grad; bw->g = g; bw->red = reduction;
    return VarDiff::node(std::move(out), g, entry(bw, g), x.history);
}
Var Var::mae(const Var& t, Reduction r) const { return loss_var(NK_LOSS_MAE, *this, t, r); }
Var Var::bce(const Var& t, Reduction r) const { return loss_var(NK_LOSS_BCE, *this, t, r); }
Var Var::bce_with_logits(const Var& t, Reduction r) const { return loss_var(NK_LOSS_BCE_WITH_LOGITS, *this, t, r); }
Var Var::kldiv(const Var& t, Reduction r) const { return loss_var(NK_LOSS_KLDIV, *this, t, r); }
Var Var::nll(const Var& t, Reduction r) const { return loss_var(-1, *this, t, r); }
VarDiff VarDiff::mae(const Var& t, Reduction r) const { return loss_diff(NK_LOSS_MAE, *this, t, r); }
VarDiff VarDiff::bce(const Var& t, Reduction r) const { return loss_diff(NK_LOSS_BCE, *this, t, r); }
VarDiff VarDiff::bce_with_logits(const Var& t, Reduction r) const { return loss_diff(NK_LOSS_BCE_WITH_LOGITS, *this, t, r); }
VarDiff VarDiff::kldiv(const Var& t, Reduction r) const { return loss_diff(NK_LOSS_KLDIV, *this, t, r); }
VarDiff VarDiff::nll(const Var& t, Reduction r) const { return loss_diff(-1, *this, t, r); }
Var Var::mv(const Var& rhs) const { return matmul_var(4, *this, rhs); }
VarDiff Var::mv(const VarDiff& rhs) const { return matmul_diff(4, *this, nullptr, nullptr, rhs.var, rhs.grad, &rhs.history); }
Var Var::vm(const Var& rhs) const { return matmul_var(5, *this, rhs); }
VarDiff Var::vm(const VarDiff& rhs) const { return matmul_diff(5, *this, nullptr, nullptr, rhs.var, rhs.grad, &rhs.history); }
Var Var::vv(const Var& rhs) const { return matmul_var(6, *this, rhs); }
VarDiff Var::vv(const VarDiff& rhs) const { return matmul_diff(6, *this, nullptr, nullptr, rhs.var, rhs.grad, &rhs.history); }
VarDiff VarDiff::mv(const Var& rhs) const { return matmul_diff(4, var, grad, &history, rhs, nullptr, nullptr); }
VarDiff VarDiff::mv(const VarDiff& rhs) const { return matmul_diff(4, var, grad, &history, rhs.var, rhs.grad, &rhs.history); }
VarDiff VarDiff::vm(const Var& rhs) const { return matmul_diff(5, var, grad, &history, rhs, nullptr, nullptr); }
VarDiff VarDiff::vm(const VarDiff& rhs) const { return matmul_diff(5, var, grad, &history, rhs.var, rhs.grad, &rhs.history); }
VarDiff VarDiff::vv(const Var& rhs) const { return matmul_diff(6, var, grad, &history, rhs, nullptr, nullptr); }
VarDiff VarDiff::vv(const VarDiff& rhs) const { return matmul_diff(6, var, grad, &history, rhs.var, rhs.grad, &rhs.history); }
Var Var::mse(const Var& target, Reduction reduction) const {
    if (target.shape() != shape()) panic("mse: input and target shapes differ");
    History<ForwardEntry> h = history;
    h.merge(target.history);
    auto op = std::make_shared<MseFwd>();
    op->x = data; op->t = target.data; op->out = zeros_like(data, {}); op->red = reduction;
    auto y = op->out;
    return Var::node(y, op, std::move(h));
}
Var Var::pad(const std::vector<int>& padding, float value) const { return pad(padding, PaddingMode::constant(value)); }
Var Var::pad(const std::vector<int>& padding, PaddingMode mode) const {
    if (padding.size() + 2 != shape().size()) panic("pad: one padding per spatial dimension expected");
    Shape s = shape();
    for (size_t i = 0; i < padding.size(); ++i) s[2 + i] += 2 * padding[i];
    auto op = std::make_shared<PadFwd>();
    op->x = data; op->y = zeros_like(data, s); op->padding = padding; op->mode = mode;
    auto y = op->y;
    return Var::node(y, op, history);
}
Var Var::mm(const Var& rhs) const { return matmul_var(0, *this, rhs); }
VarDiff Var::mm(const VarDiff& rhs) const { return matmul_diff(0, *this, nullptr, nullptr, rhs.var, rhs.grad, &rhs.history); }
Var Var::mm_t(const Var& rhs) const { return matmul_var(1, *this, rhs); }
VarDiff Var::mm_t(const VarDiff& rhs) const { return matmul_diff(1, *this, nullptr, nullptr, rhs.var, rhs.grad, &rhs.history); }
static void check_heads(const Shape& flat, const Shape& cube, int B, int S, int H, int dh, bool with_cube) {
    if (B <= 0 || S <= 0 || H <= 0 || dh <= 0) panic("heads: non-positive geometry");
    if (flat != Shape{B * S, H * dh}) panic("heads: the projection operand must have shape (B*S, H*dh)");
    if (with_cube && cube != Shape{B * H, S, S}) panic("heads: the probabilities must have shape (B*H, S, S)");
}
Var Var::heads_scores(const Var& keys, int B, int S, int H, int dh) const {
    check_heads(shape(), {}, B, S, H, dh, false);
    check_heads(keys.shape(), {}, B, S, H, dh, false);
    History<ForwardEntry> h = history;
    h.merge(keys.history);
    auto op = std::make_shared<HeadsScoresFwd>();
    op->hg = {B, S, H, dh}; op->q = data; op->k = keys.data; op->c = zeros_like(data, Shape{B * H, S, S});
    auto y = op->c;
    return Var::node(y, op, std::move(h));
}
Var Var::heads_context(const Var& values, int B, int S, int H, int dh) const {
    check_heads(values.shape(), shape(), B, S, H, dh, true);
    History<ForwardEntry> h = history;
    h.merge(values.history);
    auto op = std::make_shared<HeadsContextFwd>();
    op->hg = {B, S, H, dh}; op->p = data; op->v = values.data; op->o = zeros_like(data, Shape{B * S, H * dh});
    auto y = op->o;
    return Var::node(y, op, std::move(h));
}
VarDiff VarDiff::heads_scores(const VarDiff& keys, int B, int S, int H, int dh) const {
    Var out = var.heads_scores(keys.var, B, S, H, dh);
    History<BackwardEntry> h = history;
    h.merge(keys.history);
    auto g = std::make_shared<Gradient>(out.device(), out.shape());
    auto bw = std::make_shared<HeadsScoresBwd>();
    bw->hg = {B, S, H, dh}; bw->q = var.data; bw->k = keys.var.data; bw->dq = grad; bw->dk = keys.grad; bw->g = g;
    return VarDiff::node(std::move(out), g, entry(bw, g), std::move(h));
}
VarDiff VarDiff::heads_context(const VarDiff& values, int B, int S, int H, int dh) const {
    Var out = var.heads_context(values.var, B, S, H, dh);
    History<BackwardEntry> h = history;
    h.merge(values.history);
    auto g = std::make_shared<Gradient>(out.device(), out.shape());
    auto bw = std::make_shared<HeadsContextBwd>();
    bw->hg = {B, S, H, dh}; bw->p = var.data; bw->v = values.var.data; bw->dp = grad; bw->dv = values.grad; bw->g = g;
    return VarDiff::node(std::move(out), g, entry(bw, g), std::move(h));
}
bool Var::attention_core_supported(int S, int dh, double p) { return nk_attention_supported(S, dh, p, 1) != 0; }
static Var heads_attention_node(const Var& q, const Var& keys, const Var& values, int B, int S, int H, int dh, float scale, double p,
                                Shared<bool> status, bool keep) {
    if (!(p >= 0.0 && p <= 1.0)) panic("Wrong probability received: " + std::to_string(p) + ".");
    check_heads(q.shape(), {}, B, S, H, dh, false);
    check_heads(keys.shape(), {}, B, S, H, dh, false);
    check_heads(values.shape(), {}, B, S, H, dh, false);
    if (!Var::attention_core_supported(S, dh, p)) panic("heads_attention: the fused kernels take dh in {32, 64, 128} and p < 1");
    History<ForwardEntry> h = q.history;
    h.merge(keys.history);
    h.merge(values.history);
    auto op = std::make_shared<HeadsAttentionFwd>();
    op->hg = {B, S, H, dh}; op->q = q.data; op->k = keys.data; op->v = values.data;
    if (keep) {  // what the backward node reads; a graph without gradients keeps nothing
        const int SP = (S + 31) / 32 * 32;  // the scratch tensors are padded to whole 32 x 32 tiles (include/neuronika_hip.h)
        op->scores = zeros_like(q.data, Shape{B * H, SP, SP});
        op->stats = zeros_like(q.data, Shape{B * H, SP, 2});
        op->mask = zeros_like(q.data, Shape{B * H, SP, SP / 32});
    }
    op->o = zeros_like(q.data, Shape{B * S, H * dh});
    op->scale = scale; op->p = p; op->status = std::move(status);
    op->seed = next_node_seed();
    op->calls = std::make_shared<uint64_t>(0);
    auto y = op->o;
    return Var::node(y, op, std::move(h));
}
Var Var::heads_attention(const Var& keys, const Var& values, int B, int S, int H, int dh, float scale, double p,
                         Shared<bool> status) const {
    return heads_attention_node(*this, keys, values, B, S, H, dh, scale, p, std::move(status), false);
}
VarDiff VarDiff::heads_attention(const VarDiff& keys, const VarDiff& values, int B, int S, int H, int dh, float scale, double p,
                                 Shared<bool> status) const {
    Var out = heads_attention_node(var, keys.var, values.var, B, S, H, dh, scale, p, status, true);
    auto fwd = std::dynamic_pointer_cast<HeadsAttentionFwd>(out.history.to_vec().back().op);
    History<BackwardEntry> h = history;
    h.merge(keys.history);
    h.merge(values.history);
    auto g = std::make_shared<Gradient>(out.device(), out.shape());
    auto bw = std::make_shared<HeadsAttentionBwd>();
    bw->hg = fwd->hg; bw->q = fwd->q; bw->k = fwd->k; bw->v = fwd->v; bw->scores = fwd->scores; bw->stats = fwd->stats; bw->mask = fwd->mask; bw->o = fwd->o;
    bw->ds = zeros_like(fwd->scores, fwd->scores->shape());
    bw->dropped = zeros_like(fwd->scores, fwd->scores->shape());
    bw->dq = grad; bw->dk = keys.grad; bw->dv = values.grad; bw->g = g;
    bw->scale = scale; bw->p = p; bw->status = status;
    return VarDiff::node(std::move(out), g, entry(bw, g), std::move(h));
}
Var Var::bmm(const Var& rhs) const { return matmul_var(2, *this, rhs); }
Var Var::bmm_t(const Var& rhs) const { return matmul_var(3, *this, rhs); }
Var Var::attention_probs(float scale, double p, Shared<bool> status, bool store_probs) const {
    if (!(p >= 0.0 && p <= 1.0)) panic("Wrong probability received: " + std::to_string(p) + ".");
    if (shape().empty()) panic("attention_probs: at least one axis expected");
    auto op = std::make_shared<AttnProbsFwd>();
    op->x = data; op->out = zeros_like(data, shape());
    if (store_probs) op->probs = zeros_like(data, shape());
    op->scale = scale; op->p = p; op->status = std::move(status);
    op->seed = next_node_seed();
    op->calls = std::make_shared<uint64_t>(0);
    op->last_offset = std::make_shared<uint64_t>(0);
    auto y = op->out;
    return Var::node(y, op, history);
}
Var Var::convolution(const Var& input, const std::vector<int>& stride, const std::vector<int>& dilation, int groups) const {
    Shape kfull = shape();  // kernel (Cout, Cin/groups, k...) checked against the input
    Shape kcheck = kfull;
    kcheck[1] *= groups;
    (void)kcheck;
    Shape out = conv_out_shape(input.shape(), kfull, stride, dilation, groups);
    History<ForwardEntry> h = history;
    h.merge(input.history);
    auto op = std::make_shared<ConvFwd>();
    op->x = input.data; op->w = data; op->y = zeros_like(data, out); op->stride = stride; op->dilation = dilation; op->groups = groups;
    auto y = op->y;
    return Var::node(y, op, std::move(h));
}
static Var heads_var(bool split, const Var& x, int B, int S, int H, int dh) {
    auto op = std::make_shared<HeadsFwd>();
    op->split = split; op->x = x.data; op->B = B; op->S = S; op->H = H; op->dh = dh;
    const Shape want = split ? Shape{B * S, H * dh} : Shape{B * H, S, dh};
    if (x.shape() != want) panic("split/merge heads: unexpected input shape");
    op->y = zeros_like(x.data, split ? Shape{B * H, S, dh} : Shape{B * S, H * dh});
    auto y = op->y;
    return Var::node(y, op, x.history);
}
Var Var::split_heads(int B, int S, int H, int dh) const { return heads_var(true, *this, B, S, H, dh); }
Var Var::merge_heads(int B, int S, int H, int dh) const { return heads_var(false, *this, B, S, H, dh); }

// =================================================================================================
// VarDiff
// =================================================================================================
VarDiff VarDiff::leaf(Var var, Shared<Gradient> grad) {
    VarDiff v;
    v.var = std::move(var);
    v.grad = std::move(grad);
    return v;
}
VarDiff VarDiff::node(Var var, Shared<Gradient> grad, BackwardEntry op, History<BackwardEntry> h) {
    const void* ptr = op.op.get();
    h.insert(ptr, std::move(op));
    VarDiff v;
    v.var = std::move(var);
    v.grad = std::move(grad);
    v.history = std::move(h);
    return v;
}
void VarDiff::zero_grad() const { grad->zero(); }
void VarDiff::forward() const {
    var.forward();
    auto& buffer = history.buffer_mut();
    if (buffer.empty()) buffer = history.to_vec();
}
static thread_local BackwardHook* g_active_hook = nullptr;
// gradients whose LAST writer on the tape is the node whose backward() is running now
static thread_local const std::vector<const Gradient*>* g_final_here = nullptr;
BackwardHook* parts_hook(const Gradient* g) {
    if (!g_active_hook || !g_final_here) return nullptr;
    if (std::find(g_final_here->begin(), g_final_here->end(), g) == g_final_here->end()) return nullptr;
    return g_active_hook->wants_parts(g) ? g_active_hook : nullptr;
}
namespace {
struct ActiveHookScope {
    explicit ActiveHookScope(BackwardHook* h) { g_active_hook = h; }
    ~ActiveHookScope() { g_active_hook = nullptr; g_final_here = nullptr; }
};
}  // namespace

void VarDiff::backward(float seed, BackwardHook* hook) const {
    if (var.history.len() != var.history.buffer_len()) panic("Perhaps you forgot to call .forward()?");
    {
        bool assign = false;
        grad->borrow_first_write(assign).fill(seed);  // `grad_mut().fill(seed)` vardiff.rs:133
    }
    run_backward(hook);
}
void VarDiff::backward_from(const Var& seed, BackwardHook* hook) const {
    if (var.history.len() != var.history.buffer_len()) panic("Perhaps you forgot to call .forward()?");
    if (seed.shape() != shape()) panic("backward_from: the seed must have the shape of the root");
    // the root's backward nodes read the root gradient through `borrow()` when they run: let them see the seed's buffer
    const bool was_pending = grad->zero_pending();
    Shared<HipArray> own = grad->exchange_array(seed.data);
    struct Restore {
        const Shared<Gradient>& g; Shared<HipArray>& own; bool pending;
        ~Restore() { g->exchange_array(own); if (pending) g->zero(); }
    } restore{grad, own, was_pending};
    run_backward(hook);
}
// Pre-masked gradients (fused Linear+ReLU): a gradient that names a mask source is written pre-masked in THIS pass iff
// every tape node that writes it can do so.
static void decide_premasked(const std::vector<BackwardEntry>& buffer) {
    std::unordered_map<const Gradient*, std::pair<int, int>> count;  // gradient -> (writers, mask-capable writers)
    std::vector<const Gradient*> ts;
    for (const BackwardEntry& e : buffer) {
        if (const auto* own = dynamic_cast<const Gradient*>(e.grad.get()))
            if (own->premask_source()) count[own];  // a root has no writers: it is seeded with plain values
        ts.clear();
        e.op->targets(ts);
        for (const Gradient* g : ts)
            if (g->premask_source()) ++count[g].first;
        ts.clear();
        e.op->premask_targets(ts);
        for (const Gradient* g : ts) ++count[g].second;
    }
    for (const auto& kv : count) kv.first->set_premasked(kv.second.first > 0 && kv.second.first == kv.second.second);
}
void VarDiff::run_backward(BackwardHook* hook) const {
    auto& buffer = history.buffer_mut();
    decide_premasked(buffer);
    if (!hook) {
        for (auto it = buffer.rbegin(); it != buffer.rend(); ++it) it->op->backward();
        return;
    }
    ActiveHookScope scope(hook);
    // a gradient is final once the last node (in reverse order) that accumulates into it ran
    std::unordered_map<const Gradient*, size_t> last;
    std::vector<const Gradient*> ts, final_here;
    for (size_t i = 0; i < buffer.size(); ++i) {  // reverse execution order: index 0 runs last
        ts.clear();
        buffer[i].op->targets(ts);
        for (const Gradient* g : ts)
            if (!last.count(g)) last[g] = i;  // smallest index = last to run
    }
    for (size_t k = buffer.size(); k-- > 0;) {
        ts.clear();
        buffer[k].op->targets(ts);
        final_here.clear();
        for (const Gradient* g : ts)
            if (last[g] == k && std::find(final_here.begin(), final_here.end(), g) == final_here.end()) final_here.push_back(g);
        g_final_here = &final_here;  // what `parts_hook` answers for while this node runs
        buffer[k].op->backward();
        g_final_here = nullptr;
        for (const Gradient* g : final_here) hook->grad_ready(g);
    }
}
void VarDiff::no_grad() const {
    auto& buffer = history.buffer_mut();
    if (buffer.empty()) buffer = history.to_vec();
    for (auto& e : buffer) e.grad->no_grad();
}
void VarDiff::with_grad() const {
    auto& buffer = history.buffer_mut();
    if (buffer.empty()) buffer = history.to_vec();
    for (auto& e : buffer) e.grad->with_grad();
}

VarDiff VarDiff::sum() const { return unary_diff(Unary::Sum, 0, *this, {}); }
VarDiff VarDiff::mean() const { return unary_diff(Unary::Mean, 0, *this, {}); }
static thread_local bool g_relu_peephole = true;
namespace nn {
bool set_relu_peephole(bool on) { const bool was = g_relu_peephole; g_relu_peephole = on; return was; }
VarDiff linear_relu_from_origin(const LinearOrigin& o);
}  // namespace nn
VarDiff VarDiff::relu() const& { return unary_diff(Unary::Relu, 0, *this, shape()); }
VarDiff VarDiff::relu() && {
    // `lin.forward(x).relu()`: one Linear+ReLU node over the Linear's operands (see `linear_origin`); the temporary never runs
    if (linear_origin && g_relu_peephole) return nn::linear_relu_from_origin(*linear_origin);
    return unary_diff(Unary::Relu, 0, *this, shape());
}
VarDiff VarDiff::neg() const { return pointwise_diff(NK_NEG, 0, *this); }
VarDiff VarDiff::pow(int e) const { return pointwise_diff(NK_POW, e, *this); }
VarDiff VarDiff::sqrt() const { return pointwise_diff(NK_SQRT, 0, *this); }
VarDiff VarDiff::leaky_relu() const { return pointwise_diff(NK_LEAKY_RELU, 0, *this); }
VarDiff VarDiff::softplus() const { return pointwise_diff(NK_SOFTPLUS, 0, *this); }
VarDiff VarDiff::sigmoid() const { return pointwise_diff(NK_SIGMOID, 0, *this); }
VarDiff VarDiff::tanh() const { return pointwise_diff(NK_TANH, 0, *this); }
VarDiff VarDiff::ln() const { return pointwise_diff(NK_LN, 0, *this); }
VarDiff VarDiff::exp() const { return pointwise_diff(NK_EXP, 0, *this); }
VarDiff VarDiff::unsqueeze(int axis) const {
    Var v = var.unsqueeze(axis);
    auto g = std::make_shared<Gradient>(v.device(), v.shape());
    auto bw = std::make_shared<UnsqueezeBwd>();
    bw->dx = grad; bw->g = g;
    return VarDiff::node(std::move(v), g, entry(bw, g), history);
}
VarDiff VarDiff::softmax(int axis) const { check_axis(shape(), axis); return unary_diff(Unary::Softmax, axis, *this, shape()); }
VarDiff VarDiff::log_softmax(int axis) const { check_axis(shape(), axis); return unary_diff(Unary::LogSoftmax, axis, *this, shape()); }
VarDiff VarDiff::t() const { return unary_diff(Unary::Transpose, 0, *this, Shape(shape().rbegin(), shape().rend())); }
VarDiff VarDiff::dropout(double p, Shared<bool> status) const {
    Var v = var.dropout(p, status);
    // the forward node owns the noise buffer; share it with the backward node (var.rs:375-393)
    auto fwd = std::dynamic_pointer_cast<DropoutFwd>(v.history.to_vec().back().op);
    auto g = std::make_shared<Gradient>(device(), shape());
    auto bw = std::make_shared<DropoutBwd>();
    bw->dx = grad; bw->g = g; bw->noise = fwd->noise; bw->p = p; bw->status = status;
    return VarDiff::node(std::move(v), g, entry(bw, g), history);
}
std::vector<VarDiff> VarDiff::chunks(const Shape& chunk_size) const {
    std::vector<Var> vs = var.chunks(chunk_size);
    std::vector<VarDiff> out;
    for (size_t i = 0; i < vs.size(); ++i) {
        auto g = std::make_shared<Gradient>(device(), chunk_size);
        auto bw = std::make_shared<ChunkBwd>();
        bw->dx = grad; bw->g = g; bw->chunk_no = (int)i;
        out.push_back(VarDiff::node(std::move(vs[i]), g, entry(bw, g), history));
    }
    return out;
}
VarDiff VarDiff::cat(const std::vector<VarDiff>& vars, int axis) const {
    std::vector<Var> vs;
    History<BackwardEntry> h = history;
    auto bw = std::make_shared<CatBwd>();
    bw->operands.push_back(grad);
    for (const VarDiff& v : vars) {
        vs.push_back(v.var);
        h.merge(v.history);
        bw->operands.push_back(v.grad);
    }
    Var out = var.cat(vs, axis);
    auto g = std::make_shared<Gradient>(device(), out.shape());
    bw->g = g; bw->axis = axis;
    return VarDiff::node(std::move(out), g, entry(bw, g), std::move(h));
}
VarDiff VarDiff::stack(const std::vector<VarDiff>& vars, int axis) const {
    std::vector<Var> vs;
    History<BackwardEntry> h = history;
    auto bw = std::make_shared<CatBwd>();
    bw->operands.push_back(grad);
    for (const VarDiff& v : vars) {
        vs.push_back(v.var);
        h.merge(v.history);
        bw->operands.push_back(v.grad);
    }
    Var out = var.stack(vs, axis);
    auto g = std::make_shared<Gradient>(device(), out.shape());
    bw->g = g; bw->axis = axis; bw->stack = true;
    return VarDiff::node(std::move(out), g, entry(bw, g), std::move(h));
}
VarDiff VarDiff::mse(const Var& target, Reduction reduction) const {
    Var out = var.mse(target, reduction);
    auto g = std::make_shared<Gradient>(device(), Shape{});
    auto bw = std::make_shared<MseBwd>();
    bw->x = var.data; bw->t = target.data; bw->dx = grad; bw->g = g; bw->red = reduction;
    return VarDiff::node(std::move(out), g, entry(bw, g), history);
}
VarDiff VarDiff::pad(const std::vector<int>& padding, float value) const { return pad(padding, PaddingMode::constant(value)); }
VarDiff VarDiff::pad(const std::vector<int>& padding, PaddingMode mode) const {
    Var out = var.pad(padding, mode);
    auto g = std::make_shared<Gradient>(device(), out.shape());
    auto bw = std::make_shared<PadBwd>();
    bw->dx = grad; bw->g = g; bw->padding = padding;
    return VarDiff::node(std::move(out), g, entry(bw, g), history);
}
VarDiff VarDiff::mm(const Var& rhs) const { return matmul_diff(0, var, grad, &history, rhs, nullptr, nullptr); }
VarDiff VarDiff::mm(const VarDiff& rhs) const { return matmul_diff(0, var, grad, &history, rhs.var, rhs.grad, &rhs.history); }
VarDiff VarDiff::mm_t(const Var& rhs) const { return matmul_diff(1, var, grad, &history, rhs, nullptr, nullptr); }
VarDiff VarDiff::mm_t(const VarDiff& rhs) const { return matmul_diff(1, var, grad, &history, rhs.var, rhs.grad, &rhs.history); }
VarDiff VarDiff::bmm(const VarDiff& rhs) const { return matmul_diff(2, var, grad, &history, rhs.var, rhs.grad, &rhs.history); }
VarDiff VarDiff::bmm_t(const VarDiff& rhs) const { return matmul_diff(3, var, grad, &history, rhs.var, rhs.grad, &rhs.history); }

VarDiff VarDiff::attention_probs(float scale, double p, Shared<bool> status, bool store_probs) const {
    Var v = var.attention_probs(scale, p, status, store_probs);
    auto fwd = std::dynamic_pointer_cast<AttnProbsFwd>(v.history.to_vec().back().op);
    auto g = std::make_shared<Gradient>(device(), shape());
    auto bw = std::make_shared<AttnProbsBwd>();
    bw->dx = grad; bw->g = g; bw->probs = fwd->probs; bw->scores = fwd->x; bw->scale = scale; bw->p = p; bw->status = status;
    bw->seed = fwd->seed; bw->last_offset = fwd->last_offset;
    return VarDiff::node(std::move(v), g, entry(bw, g), history);
}
static VarDiff conv_diff(const VarDiff& kernel, const Var& input, const Shared<Gradient>& dx,
                         const History<BackwardEntry>* hx, const std::vector<int>& stride,
                         const std::vector<int>& dilation, int groups, const VarDiff* bias = nullptr,
                         const std::vector<int>* crop = nullptr) {
    Var out = kernel.var.convolution(input, stride, dilation, groups);
    History<BackwardEntry> h = kernel.history;
    if (hx) h.merge(*hx);
    if (bias) {  // one node for `convolution(..) + bias`: the bias joins the forward node and both histories
        Shape want{out.shape()[1]};
        want.insert(want.end(), out.shape().size() - 2, 1);
        if (bias->shape() != want) panic("conv bias must have shape (out_channels, 1, ...)");
        auto fw = std::dynamic_pointer_cast<ConvFwd>(out.history.to_vec().back().op);
        fw->b = bias->var.data;
        out.history.merge(bias->var.history);
        h.merge(bias->history);
    }
    auto g = std::make_shared<Gradient>(out.device(), out.shape());
    auto bw = std::make_shared<ConvBwd>();
    bw->x = input.data; bw->w = kernel.var.data; bw->dx = dx; bw->dw = kernel.grad; bw->g = g;
    if (bias) bw->db = bias->grad;
    if (crop) bw->padding = *crop;  // `input` is the zero-padded copy of the Var whose gradient is dx
    bw->stride = stride; bw->dilation = dilation; bw->groups = groups;
    return VarDiff::node(std::move(out), g, entry(bw, g), std::move(h));
}
// The Conv module node WITHOUT its Pad node: forward and kernel gradient read the unpadded input through the library's folded
// entry points (nk_conv_bias_fwd_padded, nk_conv_bwd_kernel_bias_padded), the input gradient through nk_conv_bwd_input_padded as
// before.  Only built after nk_conv_padding_folds said yes for this geometry (ConvNd::forward).
static VarDiff conv_folded(const VarDiff& kernel, const Var& input, const Shared<Gradient>& dx, const History<BackwardEntry>* hx,
                           const std::vector<int>& padding, const std::vector<int>& stride, const std::vector<int>& dilation, int groups,
                           const VarDiff& bias) {
    Shape padded = input.shape();
    for (size_t i = 0; i < padding.size(); ++i) padded[2 + i] += 2 * padding[i];
    const Shape out = conv_out_shape(padded, kernel.shape(), stride, dilation, groups);
    Shape want{out[1]};
    want.insert(want.end(), out.size() - 2, 1);
    if (bias.shape() != want) panic("conv bias must have shape (out_channels, 1, ...)");
    History<ForwardEntry> fh = kernel.var.history;
    fh.merge(input.history);
    fh.merge(bias.var.history);
    auto fw = std::make_shared<ConvFwd>();
    fw->x = input.data; fw->w = kernel.var.data; fw->b = bias.var.data; fw->y = zeros_like(kernel.var.data, out);
    fw->stride = stride; fw->dilation = dilation; fw->groups = groups; fw->fold = padding;
    auto y = fw->y;
    Var outv = Var::node(y, fw, std::move(fh));
    History<BackwardEntry> h = kernel.history;
    if (hx) h.merge(*hx);
    h.merge(bias.history);
    auto g = std::make_shared<Gradient>(outv.device(), outv.shape());
    auto bw = std::make_shared<ConvBwd>();
    bw->x = input.data; bw->w = kernel.var.data; bw->dx = dx; bw->dw = kernel.grad; bw->db = bias.grad; bw->g = g;
    bw->padding = padding; bw->x_unpadded = true;
    bw->stride = stride; bw->dilation = dilation; bw->groups = groups;
    return VarDiff::node(std::move(outv), g, entry(bw, g), std::move(h));
}
static bool padding_folds(const Var& input, const VarDiff& weight, const std::vector<int>& padding, const std::vector<int>& stride,
                          const std::vector<int>& dilation, int groups) {
    int folds = 0;
    const int nd = (int)input.shape().size() - 2;
    check(nk_conv_padding_folds(input.device()->raw(), nd, input.shape().data(), padding.data(), weight.shape().data(), stride.data(),
                                dilation.data(), groups, &folds));
    return folds != 0;
}
VarDiff VarDiff::convolution(const Var& input, const std::vector<int>& stride, const std::vector<int>& dilation, int groups) const {
    return conv_diff(*this, input, nullptr, nullptr, stride, dilation, groups);
}
VarDiff VarDiff::convolution(const VarDiff& input, const std::vector<int>& stride, const std::vector<int>& dilation, int groups) const {
    return conv_diff(*this, input.var, input.grad, &input.history, stride, dilation, groups);
}
static VarDiff heads_diff(bool split, const VarDiff& x, int B, int S, int H, int dh) {
    Var out = split ? x.var.split_heads(B, S, H, dh) : x.var.merge_heads(B, S, H, dh);
    auto g = std::make_shared<Gradient>(out.device(), out.shape());
    auto bw = std::make_shared<HeadsBwd>();
    bw->split = split; bw->dx = x.grad; bw->g = g; bw->B = B; bw->S = S; bw->H = H; bw->dh = dh;
    return VarDiff::node(std::move(out), g, entry(bw, g), x.history);
}
VarDiff VarDiff::split_heads(int B, int S, int H, int dh) const { return heads_diff(true, *this, B, S, H, dh); }
VarDiff VarDiff::merge_heads(int B, int S, int H, int dh) const { return heads_diff(false, *this, B, S, H, dh); }

// ---- arithmetic operators -------------------------------------------------------------------------
static Var scalar_leaf(const DevicePtr& dev, float v) {  // the f32 is wrapped in an Ix0 leaf (var.rs:746-838)
    auto a = std::make_shared<HipArray>(dev, Shape{});
    a->fill(v);
    return Var::leaf(a);
}
#define NK_DEFINE_BINARY(OP, CODE)                                                                          \
    Var operator OP(const Var& l, const Var& r) { return binary_var(CODE, l, r); }                          \
    VarDiff operator OP(const Var& l, const VarDiff& r) { return binary_diff(CODE, l, nullptr, nullptr, r.var, r.grad, &r.history); } \
    VarDiff operator OP(const VarDiff& l, const Var& r) { return binary_diff(CODE, l.var, l.grad, &l.history, r, nullptr, nullptr); } \
    VarDiff operator OP(const VarDiff& l, const VarDiff& r) { return binary_diff(CODE, l.var, l.grad, &l.history, r.var, r.grad, &r.history); } \
    Var operator OP(const Var& l, float r) { return binary_var(CODE, l, scalar_leaf(l.device(), r)); }      \
    VarDiff operator OP(const VarDiff& l, float r) { return binary_diff(CODE, l.var, l.grad, &l.history, scalar_leaf(l.device(), r), nullptr, nullptr); }
NK_DEFINE_BINARY(+, NK_ADD)
NK_DEFINE_BINARY(-, NK_SUB)
NK_DEFINE_BINARY(*, NK_MUL)
NK_DEFINE_BINARY(/, NK_DIV)
#undef NK_DEFINE_BINARY

// ---- leaf constructors ----------------------------------------------------------------------------
Var from_host(DevicePtr dev, const Shape& shape, const float* host) { return Var::leaf(HipArray::from_host(std::move(dev), shape, host)); }
Var zeros(DevicePtr dev, const Shape& shape) { return Var::leaf(std::make_shared<HipArray>(std::move(dev), shape)); }
Var full(DevicePtr dev, const Shape& shape, float value) {
    auto a = std::make_shared<HipArray>(std::move(dev), shape);
    a->fill(value);
    return Var::leaf(a);
}
Var ones(DevicePtr dev, const Shape& shape) { return full(std::move(dev), shape, 1.f); }
static std::vector<float> uniform(size_t n, float lo, float hi, uint64_t seed) {
    std::mt19937_64 rng(seed);
    std::vector<float> v(n);
    for (float& x : v) x = lo + (hi - lo) * (float)((rng() >> 40) * (1.0 / 16777216.0));
    return v;
}
Var rand(DevicePtr dev, const Shape& shape, uint64_t seed) {
    const auto v = uniform(numel(shape), 0.f, 1.f, seed);
    return from_host(std::move(dev), shape, v.data());
}

// ndarray's `Linspace`: element i = start + step * i with step = (end - start) / (n - 1)
static std::vector<float> linspace_host(float start, float end, int n) {
    if (n < 0) panic("linspace: negative length");
    std::vector<float> v((size_t)n);
    const float step = n > 1 ? (end - start) / (float)(n - 1) : 0.f;
    for (int i = 0; i < n; ++i) v[(size_t)i] = start + step * (float)i;
    return v;
}
Var eye(DevicePtr dev, int n) {
    if (n < 0) panic("eye: negative size");
    std::vector<float> h((size_t)n * n, 0.f);
    for (int i = 0; i < n; ++i) h[(size_t)i * n + i] = 1.f;
    return from_host(std::move(dev), Shape{n, n}, h.data());
}
Var linspace(DevicePtr dev, float start, float end, int n) {
    const auto v = linspace_host(start, end, n);
    return from_host(std::move(dev), Shape{n}, v.data());
}
Var logspace(DevicePtr dev, float base, float start, float end, int n) {
    auto v = linspace_host(start, end, n);
    const float sign = base < 0.f ? -1.f : 1.f, b = std::fabs(base);
    for (float& e : v) e = sign * std::pow(b, e);
    return from_host(std::move(dev), Shape{n}, v.data());
}
Var geomspace(DevicePtr dev, float start, float end, int n) {
    if (start == 0.f || end == 0.f || (start < 0.f) != (end < 0.f))
        panic("geomspace: the reference returns None (an endpoint is zero or the endpoints differ in sign)");
    auto v = linspace_host(std::log(std::fabs(start)), std::log(std::fabs(end)), n);
    const float sign = start < 0.f ? -1.f : 1.f;
    for (float& e : v) e = sign * std::exp(e);
    return from_host(std::move(dev), Shape{n}, v.data());
}
Var range(DevicePtr dev, float start, float end, float step) {
    const float cnt = std::ceil((end - start) / step);
    const int n = (cnt > 0.f && std::isfinite(cnt)) ? (int)cnt : 0;  // `as usize` saturates negatives / NaN to 0
    std::vector<float> v((size_t)n);
    for (int i = 0; i < n; ++i) v[(size_t)i] = start + step * (float)i;
    return from_host(std::move(dev), Shape{n}, v.data());
}

// =================================================================================================
// nn
// =================================================================================================
namespace nn {

static VarDiff uniform_param(const DevicePtr& dev, const Shape& s, float k, uint64_t seed) {
    const auto v = uniform(numel(s), -k, k, seed);
    return from_host(dev, s, v.data()).requires_grad();
}
namespace init {
float calculate_gain(const std::string& nl) {
    if (nl == "linear" || nl == "sigmoid") return 1.f;
    if (nl == "tanh") return 5.f / 3.f;
    if (nl == "relu") return std::sqrt(2.f);
    if (nl == "leaky_relu") return std::sqrt(2.f / (1.f + 0.01f * 0.01f));
    panic("error: unsupported nonlinearity: " + nl);
}
std::pair<float, float> calculate_fan_in_fan_out(const VarDiff& param) {
    const Shape& s = param.shape();
    if (s.size() < 2) panic("index out of bounds: fan in / fan out need at least 2 dimensions");  // `shape[1]`
    size_t fan_in = (size_t)s[1], fan_out = (size_t)s[0];
    if (s.size() > 2) {
        size_t numel = 0;  // init.rs:55: `.skip(2).sum()` - the reference SUMS the trailing extents
        for (size_t i = 2; i < s.size(); ++i) numel += (size_t)s[i];
        fan_in *= numel; fan_out *= numel;
    }
    return {(float)fan_in, (float)fan_out};
}
void constant(const VarDiff& p, float v) { p.var.data->fill(v); }
void zeros(const VarDiff& p) { p.var.data->fill(0.f); }
void ones(const VarDiff& p) { p.var.data->fill(1.f); }
void eye(const VarDiff& p) {
    const Shape& s = p.shape();
    if (s.size() != 2) panic("eye: a 2-dimensional parameter is expected");
    std::vector<float> h(numel(s), 0.f);
    for (int i = 0; i < std::min(s[0], s[1]); ++i) h[(size_t)i * s[1] + i] = 1.f;
    p.var.data->upload(h.data());
}
void dirac(const VarDiff& p, int groups) {
    const Shape& s = p.shape();
    if (s.size() < 3 || s.size() > 5) panic("error: only 3, 4 and 5 dimensional parameters are supported.");
    if (groups < 1 || s[0] % groups != 0) panic("error: output channels must be divisible by groups.");
    std::vector<float> h = p.to_vec();  // only the selected elements are set (init.rs:152-169), the rest is kept
    const int opg = s[0] / groups, min_dim = std::min(opg, s[1]);
    std::vector<size_t> stride(s.size(), 1);
    for (int i = (int)s.size() - 2; i >= 0; --i) stride[i] = stride[i + 1] * (size_t)s[i + 1];
    for (int g = 0; g < groups; ++g)
        for (int d = 0; d < min_dim; ++d) {
            size_t off = (size_t)(g * opg + d) * stride[0] + (size_t)d * stride[1];
            for (size_t i = 2; i < s.size(); ++i) off += (size_t)(s[i] / 2) * stride[i];
            h[off] = 1.f;
        }
    p.var.data->upload(h.data());
}
void uniform(const VarDiff& p, float low, float high, uint64_t seed) {
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<float> d(low, high);
    std::vector<float> h(numel(p.shape()));
    for (float& v : h) v = d(rng);
    p.var.data->upload(h.data());
}
void normal(const VarDiff& p, float mean, float std, uint64_t seed) {
    if (!(std >= 0.f) || !std::isfinite(std)) panic("called `Result::unwrap()` on an `Err` value: BadVariance");  // Normal::new(..).unwrap()
    std::mt19937_64 rng(seed);
    std::normal_distribution<float> d(mean, std);
    std::vector<float> h(numel(p.shape()));
    for (float& v : h) v = std == 0.f ? mean : d(rng);
    p.var.data->upload(h.data());
}
void xavier_uniform(const VarDiff& p, float gain, uint64_t seed) {
    const auto f = calculate_fan_in_fan_out(p);
    const float sd = gain * std::sqrt(2.f / (f.first + f.second)), a = std::sqrt(3.f) * sd;
    uniform(p, -a, a, seed);
}
void xavier_normal(const VarDiff& p, float gain, uint64_t seed) {
    const auto f = calculate_fan_in_fan_out(p);
    normal(p, 0.f, gain * std::sqrt(2.f / (f.first + f.second)), seed);
}
}  // namespace init

Linear::Linear(DevicePtr dev, int in_features, int out_features, uint64_t seed)
    : weight(uniform_param(dev, {out_features, in_features}, 1.f / std::sqrt((float)in_features), seed)),
      bias(uniform_param(dev, {out_features}, 1.f / std::sqrt((float)in_features), seed + 1)) {}
static VarDiff linear_node(const Linear& l, const Var& x, const Shared<Gradient>& dx, const History<BackwardEntry>* hx, bool relu = false) {
    const Shape& xs = x.shape();
    const Shape& ws = l.weight.shape();
    if (xs.size() != 2 || ws.size() != 2 || xs[1] != ws[1]) panic("Shapes are incompatible for matrix multiplication.");
    if (l.bias.shape() != Shape{ws[0]}) panic("Linear: bias must have shape (out_features)");
    History<ForwardEntry> h = x.history;
    h.merge(l.weight.var.history);
    h.merge(l.bias.var.history);
    auto fw = std::make_shared<LinearFwd>();
    fw->x = x.data; fw->w = l.weight.var.data; fw->b = l.bias.var.data; fw->y = zeros_like(x.data, Shape{xs[0], ws[0]});
    fw->relu = relu;
    auto y = fw->y;
    Var var = Var::node(y, fw, std::move(h));
    History<BackwardEntry> hb;
    if (hx) hb = *hx;
    hb.merge(l.weight.history);
    hb.merge(l.bias.history);
    auto g = std::make_shared<Gradient>(var.device(), var.shape());
    if (relu) g->set_premask_source(y);
    auto bw = std::make_shared<LinearBwd>();
    bw->x = x.data; bw->w = l.weight.var.data; bw->dx = dx; bw->dw = l.weight.grad; bw->db = l.bias.grad; bw->g = g;
    if (relu) bw->y = y;
    VarDiff out = VarDiff::node(std::move(var), g, entry(bw, g), std::move(hb));
    if (!relu) {
        auto origin = std::make_shared<LinearOrigin>();
        origin->weight = l.weight; origin->bias = l.bias; origin->input = x; origin->input_grad = dx;
        origin->differentiable_input = hx != nullptr;
        if (hx) origin->input_history = *hx;
        out.linear_origin = std::move(origin);
    }
    return out;
}
VarDiff linear_relu_from_origin(const LinearOrigin& o) {
    const Linear l(o.weight, o.bias);
    return linear_node(l, o.input, o.input_grad, o.differentiable_input ? &o.input_history : nullptr, true);
}
VarDiff Linear::forward(const Var& input) const {
    return fused ? linear_node(*this, input, nullptr, nullptr) : input.mm_t(weight) + bias;
}
VarDiff Linear::forward(const VarDiff& input) const {
    return fused ? linear_node(*this, input.var, input.grad, &input.history) : input.mm_t(weight) + bias;
}
VarDiff Linear::forward_relu(const Var& input) const {
    return fused ? linear_node(*this, input, nullptr, nullptr, true) : (input.mm_t(weight) + bias).relu();
}
VarDiff Linear::forward_relu(const VarDiff& input) const {
    return fused ? linear_node(*this, input.var, input.grad, &input.history, true) : (input.mm_t(weight) + bias).relu();
}

LSTMCell::LSTMCell(DevicePtr dev, int input_size, int hidden_size, uint64_t seed)
    : weight_ih(uniform_param(dev, {4 * hidden_size, input_size}, 1.f / std::sqrt((float)hidden_size), seed)),
      weight_hh(uniform_param(dev, {4 * hidden_size, hidden_size}, 1.f / std::sqrt((float)hidden_size), seed + 1)),
      bias_ih(uniform_param(dev, {4 * hidden_size}, 1.f / std::sqrt((float)hidden_size), seed + 2)),
      bias_hh(uniform_param(dev, {4 * hidden_size}, 1.f / std::sqrt((float)hidden_size), seed + 3)) {}
template <class I>
static std::pair<VarDiff, VarDiff> lstm_step(const LSTMCell& c, const std::pair<VarDiff, VarDiff>& state, const I& input) {
    const VarDiff& cell_state = state.first;
    const VarDiff& hidden = state.second;
    VarDiff gates = hidden.mm_t(c.weight_hh) + c.bias_hh + input.mm_t(c.weight_ih) + c.bias_ih;
    const Shape gs = gates.shape();
    auto ch = gates.chunks({gs[0], gs[1] / 4});
    VarDiff input_gate = ch[0].sigmoid(), forget_gate = ch[1].tanh(), cell_state_gate = ch[2].sigmoid(),
            output_gate = ch[3].sigmoid();
    VarDiff new_cell_state = forget_gate * cell_state + (input_gate * cell_state_gate);
    VarDiff new_hidden = output_gate * new_cell_state.tanh();
    return {new_cell_state, new_hidden};
}
std::pair<VarDiff, VarDiff> LSTMCell::forward(const std::pair<VarDiff, VarDiff>& st, const Var& in) const { return lstm_step(*this, st, in); }
std::pair<VarDiff, VarDiff> LSTMCell::forward(const std::pair<VarDiff, VarDiff>& st, const VarDiff& in) const { return lstm_step(*this, st, in); }

GRUCell::GRUCell(DevicePtr dev, int input_size, int hidden_size, uint64_t seed)
    : weight_ih(uniform_param(dev, {3 * hidden_size, input_size}, 1.f / std::sqrt((float)hidden_size), seed)),
      weight_hh(uniform_param(dev, {3 * hidden_size, hidden_size}, 1.f / std::sqrt((float)hidden_size), seed + 1)),
      bias_ih(uniform_param(dev, {3 * hidden_size}, 1.f / std::sqrt((float)hidden_size), seed + 2)),
      bias_hh(uniform_param(dev, {3 * hidden_size}, 1.f / std::sqrt((float)hidden_size), seed + 3)) {}
template <class I>
static VarDiff gru_step(const GRUCell& c, const VarDiff& hidden, const I& input) {
    VarDiff igates = input.mm_t(c.weight_ih) + c.bias_ih;
    VarDiff hgates = hidden.mm_t(c.weight_hh) + c.bias_hh;
    const Shape gs = hgates.shape();
    auto ci = igates.chunks({gs[0], gs[1] / 3}), chh = hgates.chunks({gs[0], gs[1] / 3});
    VarDiff reset_gate = (chh[0] + ci[0]).sigmoid();
    VarDiff input_gate = (chh[1] + ci[1]).sigmoid();
    VarDiff new_gate = (ci[2] + (chh[2] * reset_gate)).tanh();
    return (hidden - new_gate) * input_gate + new_gate;
}
VarDiff GRUCell::forward(const VarDiff& h, const Var& in) const { return gru_step(*this, h, in); }
VarDiff GRUCell::forward(const VarDiff& h, const VarDiff& in) const { return gru_step(*this, h, in); }

static Shape conv_weight_shape(int out_channels, int in_per_group, const std::vector<int>& kernel) {
    Shape w{out_channels, in_per_group};
    w.insert(w.end(), kernel.begin(), kernel.end());
    return w;
}
static Shape conv_bias_shape(int out_channels, int nd) {
    Shape b{out_channels};
    b.insert(b.end(), nd, 1);
    return b;
}
static float conv_init_bound(int in_per_group, const std::vector<int>& kernel) {
    int fan = in_per_group;
    for (int k : kernel) fan *= k;
    return std::sqrt(1.f / (float)fan);
}
ConvNd::ConvNd(int nd, DevicePtr dev, int in_channels, int out_channels, std::vector<int> kernel, std::vector<int> padding_,
               PaddingMode mode, std::vector<int> stride_, std::vector<int> dilation_, int groups_, uint64_t seed)
    : weight(uniform_param(dev, conv_weight_shape(out_channels, in_channels / groups_, kernel),
                           conv_init_bound(in_channels / groups_, kernel), seed)),
      bias(uniform_param(dev, conv_bias_shape(out_channels, nd), conv_init_bound(in_channels / groups_, kernel), seed + 1)),
      padding(std::move(padding_)), stride(std::move(stride_)), dilation(std::move(dilation_)), padding_mode(mode),
      groups(groups_) {
    if ((int)kernel.size() != nd || (int)padding.size() != nd || (int)stride.size() != nd || (int)dilation.size() != nd)
        panic("Conv" + std::to_string(nd) + "d: kernel/padding/stride/dilation need " + std::to_string(nd) + " entries");
}
VarDiff ConvNd::forward(const Var& input) const {
    const bool zero_pad = padding_mode.kind == PaddingMode::Zero || (padding_mode.kind == PaddingMode::Constant && padding_mode.value == 0.f);
    if (fused && fold_padding && zero_pad && padding_folds(input, weight, padding, stride, dilation, groups))
        return conv_folded(weight, input, nullptr, nullptr, padding, stride, dilation, groups, bias);
    const Var padded = input.pad(padding, padding_mode);
    if (!fused) return weight.convolution(padded, stride, dilation, groups) + bias;
    return conv_diff(weight, padded, nullptr, nullptr, stride, dilation, groups, &bias);
}
VarDiff ConvNd::forward(const VarDiff& input) const {
    bool any_pad = false;
    for (int p : padding) any_pad = any_pad || p != 0;
    if (fused && any_pad && (padding_mode.kind == PaddingMode::Zero || (padding_mode.kind == PaddingMode::Constant && padding_mode.value == 0.f))) {
        // pad -> convolution -> + bias with the Pad node's backward folded into the convolution's: the padded copy is a
        // forward-only node, the convolution's input gradient lands in input.grad directly (no padded gradient buffer) - or, where
        // the library's kernels read the unpadded input, no Pad node at all
        if (fold_padding && padding_folds(input.var, weight, padding, stride, dilation, groups))
            return conv_folded(weight, input.var, input.grad, &input.history, padding, stride, dilation, groups, bias);
        const Var padded = input.var.pad(padding, padding_mode);
        return conv_diff(weight, padded, input.grad, &input.history, stride, dilation, groups, &bias, &padding);
    }
    const VarDiff padded = input.pad(padding, padding_mode);
    if (!fused) return weight.convolution(padded, stride, dilation, groups) + bias;
    return conv_diff(weight, padded.var, padded.grad, &padded.history, stride, dilation, groups, &bias);
}

// a Linear whose weight / bias (and their gradients) are rows [row0, row0 + d) of the packed storage, initialised as
// `Linear(dev, d, d, seed)` would be
static Linear packed_linear(const Shared<HipArray>& w_all, const Shared<HipArray>& b_all, const Shared<HipArray>& gw_all,
                            const Shared<HipArray>& gb_all, int row0, int d, uint64_t seed) {
    const float k = 1.f / std::sqrt((float)d);
    auto w = std::make_shared<HipArray>(w_all, (size_t)row0 * d, Shape{d, d});
    auto b = std::make_shared<HipArray>(b_all, (size_t)row0, Shape{d});
    w->upload(uniform((size_t)d * d, -k, k, seed).data());
    b->upload(uniform((size_t)d, -k, k, seed + 1).data());
    return Linear(VarDiff::leaf(Var::leaf(w), std::make_shared<Gradient>(gw_all, (size_t)row0 * d, Shape{d, d})),
                  VarDiff::leaf(Var::leaf(b), std::make_shared<Gradient>(gb_all, (size_t)row0, Shape{d})));
}
static Linear placeholder_linear(const DevicePtr& dev) { return Linear(zeros(dev, {1, 1}).requires_grad(), zeros(dev, {1}).requires_grad()); }
MultiheadAttention::MultiheadAttention(DevicePtr dev, int d_model_, int heads_, double p, uint64_t seed)
    : q(placeholder_linear(dev)), k(placeholder_linear(dev)), v(placeholder_linear(dev)), o(dev, d_model_, d_model_, seed + 6),
      d_model(d_model_), heads(heads_), drop(p) {
    if (d_model % heads != 0) panic("d_model must be divisible by heads");
    if (d_model % 4 != 0) {
        // the bias views of a packed allocation start d and 2d floats in: not 16-byte aligned unless d % 4 == 0, and kernels
        // that take whole float4s (nk_fill with a value, the pointwise kernels) refuse such a buffer.  Three ordinary layers.
        q = Linear(dev, d_model, d_model, seed); k = Linear(dev, d_model, d_model, seed + 2); v = Linear(dev, d_model, d_model, seed + 4);
        packed_qkv = false;
        return;
    }
    wqkv_ = std::make_shared<HipArray>(dev, Shape{3 * d_model, d_model});
    bqkv_ = std::make_shared<HipArray>(dev, Shape{3 * d_model});
    gwqkv_ = std::make_shared<HipArray>(dev, Shape{3 * d_model, d_model}, HipArray::Uninit{});
    gbqkv_ = std::make_shared<HipArray>(dev, Shape{3 * d_model}, HipArray::Uninit{});
    q = packed_linear(wqkv_, bqkv_, gwqkv_, gbqkv_, 0, d_model, seed);
    k = packed_linear(wqkv_, bqkv_, gwqkv_, gbqkv_, d_model, d_model, seed + 2);
    v = packed_linear(wqkv_, bqkv_, gwqkv_, gbqkv_, 2 * d_model, d_model, seed + 4);
}
MultiheadAttention::MultiheadAttention(Linear q_, Linear k_, Linear v_, Linear o_, int heads_, double p)
    : q(std::move(q_)), k(std::move(k_)), v(std::move(v_)), o(std::move(o_)), d_model(q.weight.shape()[1]), heads(heads_), drop(p) {
    if (d_model % heads != 0) panic("d_model must be divisible by heads");
    packed_qkv = false;
}
static VarDiff qkv_attention_node(const MultiheadAttention& m, const Shared<HipArray>& w, const Shared<HipArray>& b, const Shared<HipArray>& gw,
                                  const Shared<HipArray>& gb, const VarDiff& x, int B, int S, int H, int dh, float scale) {
    const int d = H * dh;
    History<ForwardEntry> hf = x.var.history;
    for (const Linear* l : {&m.q, &m.k, &m.v}) { hf.merge(l->weight.var.history); hf.merge(l->bias.var.history); }
    auto fw = std::make_shared<QkvAttentionFwd>();
    fw->hg = {B, S, H, dh}; fw->x = x.var.data; fw->w = w; fw->b = b;
    fw->qkv = zeros_like(x.var.data, Shape{B * S, 3 * d});
    const int SP = (S + 31) / 32 * 32;
    fw->scores = zeros_like(x.var.data, Shape{B * H, SP, SP});
    fw->stats = zeros_like(x.var.data, Shape{B * H, SP, 2});
    fw->mask = zeros_like(x.var.data, Shape{B * H, SP, SP / 32});
    fw->o = zeros_like(x.var.data, Shape{B * S, d});
    fw->scale = scale; fw->p = m.drop.p; fw->status = m.drop.status;
    fw->seed = next_node_seed();
    fw->calls = std::make_shared<uint64_t>(0);
    Var out = Var::node(fw->o, fw, std::move(hf));
    History<BackwardEntry> hb = x.history;
    for (const Linear* l : {&m.q, &m.k, &m.v}) { hb.merge(l->weight.history); hb.merge(l->bias.history); }
    auto g = std::make_shared<Gradient>(out.device(), out.shape());
    auto bw = std::make_shared<QkvAttentionBwd>();
    bw->hg = fw->hg; bw->x = fw->x; bw->w = w; bw->qkv = fw->qkv; bw->scores = fw->scores; bw->stats = fw->stats; bw->mask = fw->mask; bw->o = fw->o;
    bw->dqkv = zeros_like(fw->qkv, fw->qkv->shape());
    bw->ds = zeros_like(fw->scores, fw->scores->shape());
    bw->dropped = zeros_like(fw->scores, fw->scores->shape());
    bw->gw_all = gw; bw->gb_all = gb;
    bw->dx = x.grad;
    const Linear* ls[3] = {&m.q, &m.k, &m.v};
    for (int i = 0; i < 3; ++i) { bw->gw[i] = ls[i]->weight.grad; bw->gb[i] = ls[i]->bias.grad; }
    bw->g = g; bw->scale = scale; bw->p = m.drop.p; bw->status = m.drop.status;
    return VarDiff::node(std::move(out), g, entry(bw, g), std::move(hb));
}
VarDiff MultiheadAttention::forward(const VarDiff& x, int batch) const {
    const int rows = x.shape()[0], S = rows / batch, dh = d_model / heads;
    if (rows % batch != 0 || x.shape()[1] != d_model) panic("MultiheadAttention: bad input shape");
    const float scale = 1.f / std::sqrt((float)dh);
    // q / k / v are public members: the packed path is only valid while they still ARE the views of the packed storage
    // (after `mha.q = Linear(...)` or a swap with deserialised layers the three-node path below runs on the new layers)
    auto still_packed = [&]() {
        const Linear* ls[3] = {&q, &k, &v};
        const size_t dd = (size_t)d_model * d_model;
        for (int i = 0; i < 3; ++i) {
            if (ls[i]->weight.var.data->ptr() != wqkv_->ptr() + i * dd || ls[i]->bias.var.data->ptr() != bqkv_->ptr() + (size_t)i * d_model) return false;
            if (ls[i]->weight.shape() != Shape{d_model, d_model} || ls[i]->bias.shape() != Shape{d_model}) return false;
            if (!ls[i]->weight.grad->is_view_of(gwqkv_, i * dd) || !ls[i]->bias.grad->is_view_of(gbqkv_, (size_t)i * d_model)) return false;
        }
        return true;
    };
    if (packed_qkv && wqkv_ && strided_heads && fused && fused_core && q.fused && k.fused && v.fused && Var::attention_core_supported(S, dh, drop.p) &&
        still_packed())
        return o.forward(qkv_attention_node(*this, wqkv_, bqkv_, gwqkv_, gbqkv_, x, batch, S, heads, dh, scale));
    if (strided_heads && dh % 4 == 0) {  // attention GEMMs address the heads inside the projection layout: no copies
        const VarDiff Qf = q.forward(x), Kf = k.forward(x), Vf = v.forward(x);
        if (fused && fused_core && Var::attention_core_supported(S, dh, drop.p))
            return o.forward(Qf.heads_attention(Kf, Vf, batch, S, heads, dh, scale, drop.p, drop.status));
        const VarDiff scores = Qf.heads_scores(Kf, batch, S, heads, dh);
        const VarDiff P = (fused && S % 4 == 0 && S <= 2048) ? scores.attention_probs(scale, drop.p, drop.status)
                                                             : drop.forward((scores * scale).softmax(2));
        return o.forward(P.heads_context(Vf, batch, S, heads, dh));
    }
    const VarDiff Q = q.forward(x).split_heads(batch, S, heads, dh);
    const VarDiff K = k.forward(x).split_heads(batch, S, heads, dh);
    const VarDiff V = v.forward(x).split_heads(batch, S, heads, dh);
    const VarDiff P = (fused && S % 4 == 0 && S <= 2048)
                          ? Q.bmm_t(K).attention_probs(scale, drop.p, drop.status)
                          : drop.forward((Q.bmm_t(K) * scale).softmax(2));
    const VarDiff O = P.bmm(V).merge_heads(batch, S, heads, dh);
    return o.forward(O);
}

}  // namespace nn

// =================================================================================================
// serde (neuronika-variable/src/serde.rs:10-58): ndarray's {"v":1,"dim":[..],"data":[..]} wire format
// =================================================================================================
namespace serde {

namespace {
struct Parser {
    const std::string& s;
    size_t i = 0;
    explicit Parser(const std::string& s) : s(s) {}
    [[noreturn]] void fail(const char* what) const { panic(std::string("json: ") + what + " at offset " + std::to_string(i)); }
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i; }
    bool eat(char c) { ws(); if (i < s.size() && s[i] == c) { ++i; return true; } return false; }
    void expect(char c) { if (!eat(c)) fail("unexpected character"); }
    std::string string() {
        expect('"');
        std::string out;
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\') {
                if (++i >= s.size()) fail("bad escape");
                switch (s[i]) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': fail("\\u escapes are not needed for this schema");
                    default: out += s[i];
                }
                ++i;
            } else {
                out += s[i++];
            }
        }
        if (i >= s.size()) fail("unterminated string");
        ++i;
        return out;
    }
    Json value() {
        ws();
        if (i >= s.size()) fail("unexpected end");
        Json j;
        const char c = s[i];
        if (c == '{') {
            ++i; j.kind = Json::Object;
            if (eat('}')) return j;
            do {
                ws();
                std::string k = string();
                expect(':');
                j.members.emplace_back(std::move(k), value());
            } while (eat(','));
            expect('}');
        } else if (c == '[') {
            ++i; j.kind = Json::Array;
            if (eat(']')) return j;
            do { j.items.push_back(value()); } while (eat(','));
            expect(']');
        } else if (c == '"') {
            j.kind = Json::String; j.text = string();
        } else if (s.compare(i, 4, "true") == 0) { j.kind = Json::Bool; j.b = true; i += 4;
        } else if (s.compare(i, 5, "false") == 0) { j.kind = Json::Bool; i += 5;
        } else if (s.compare(i, 4, "null") == 0) { i += 4;
        } else {
            const size_t b = i;
            while (i < s.size() && (std::isdigit((unsigned char)s[i]) || s[i] == '-' || s[i] == '+' || s[i] == '.' || s[i] == 'e' || s[i] == 'E')) ++i;
            if (b == i) fail("unexpected token");
            j.kind = Json::Number; j.text = s.substr(b, i - b);
        }
        return j;
    }
};

void write_f32(std::string& out, float v) {
    if (!std::isfinite(v)) { out += "null"; return; }  // serde_json writes non-finite floats as null
    char buf[32];
    auto r = std::to_chars(buf, buf + sizeof buf, v);  // shortest representation that round-trips
    std::string t(buf, r.ptr);
    if (t.find_first_of(".en") == std::string::npos) t += ".0";  // serde_json keeps floats recognisable ("1.0")
    out += t;
}
float read_f32(const Json& j) {
    if (j.kind == Json::Null) return std::numeric_limits<float>::quiet_NaN();
    if (j.kind != Json::Number) panic("json: number expected in \"data\"");
    float v = 0.f;
    auto r = std::from_chars(j.text.data(), j.text.data() + j.text.size(), v);  // correctly rounded, like Rust's parser
    if (r.ec == std::errc::result_out_of_range) return j.text[0] == '-' ? -INFINITY : INFINITY;
    if (r.ec != std::errc() || r.ptr != j.text.data() + j.text.size()) panic("json: bad number '" + j.text + "'");
    return v;
}
std::string array_json(const Shape& shape, const std::vector<float>& data) {
    std::string out = "{\"v\":1,\"dim\":[";
    for (size_t i = 0; i < shape.size(); ++i) { if (i) out += ','; out += std::to_string(shape[i]); }
    out += "],\"data\":[";
    for (size_t i = 0; i < data.size(); ++i) { if (i) out += ','; write_f32(out, data[i]); }
    out += "]}";
    return out;
}
}  // namespace

const Json& Json::at(const std::string& key) const {
    for (const auto& m : members) if (m.first == key) return m.second;
    panic("json: missing field `" + key + "`");
}
bool Json::has(const std::string& key) const {
    for (const auto& m : members) if (m.first == key) return true;
    return false;
}
Json parse(const std::string& text) {
    Parser p(text);
    Json j = p.value();
    p.ws();
    if (p.i != text.size()) p.fail("trailing characters");
    return j;
}

std::string to_json(const Var& v) { return array_json(v.shape(), v.to_vec()); }
std::string to_json(const VarDiff& v) { return array_json(v.shape(), v.to_vec()); }

Var var_from_json(DevicePtr dev, const Json& j) {
    if (j.kind != Json::Object) panic("json: ndarray object expected");
    const Json& ver = j.at("v");
    if (ver.kind != Json::Number || ver.text != "1") panic("json: unknown array version");  // ndarray's ARRAY_FORMAT_VERSION
    const Json& dim = j.at("dim");
    Shape shape;
    if (dim.kind == Json::Array) {
        for (const Json& d : dim.items) {
            if (d.kind != Json::Number) panic("json: bad dim");
            shape.push_back(std::stoi(d.text));
        }
    } else if (dim.kind == Json::Number) {  // Ix1 may also be written as a bare integer by serde tuples of one
        shape.push_back(std::stoi(dim.text));
    } else {
        panic("json: bad dim");
    }
    const Json& data = j.at("data");
    if (data.kind != Json::Array) panic("json: bad data");
    if (data.items.size() != numel(shape)) panic("json: data length does not match dim");  // ndarray: "data and dimension must match in size"
    std::vector<float> host(data.items.size());
    for (size_t i = 0; i < host.size(); ++i) host[i] = read_f32(data.items[i]);
    return from_host(std::move(dev), shape, host.data());
}
Var var_from_json(DevicePtr dev, const std::string& text) { return var_from_json(std::move(dev), parse(text)); }
VarDiff vardiff_from_json(DevicePtr dev, const Json& j) { return var_from_json(std::move(dev), j).requires_grad(); }
VarDiff vardiff_from_json(DevicePtr dev, const std::string& text) { return vardiff_from_json(std::move(dev), parse(text)); }

std::string to_json(const nn::Linear& l) { return "{\"weight\":" + to_json(l.weight) + ",\"bias\":" + to_json(l.bias) + "}"; }
nn::Linear linear_from_json(DevicePtr dev, const Json& j) {
    return nn::Linear(vardiff_from_json(dev, j.at("weight")), vardiff_from_json(dev, j.at("bias")));
}
nn::Linear linear_from_json(DevicePtr dev, const std::string& text) { return linear_from_json(std::move(dev), parse(text)); }

}  // namespace serde

// =================================================================================================
// optim
// =================================================================================================
namespace optim {

void Optimizer::register_param(const VarDiff& p) {
    params_.push_back(p);
    std::vector<Shared<HipArray>> st;
    for (int i = 0; i < nstate_; ++i) st.push_back(std::make_shared<HipArray>(p.device(), p.shape()));
    state_.push_back(std::move(st));
    steps_.push_back(0);
}
void Optimizer::step() {
    for (size_t i = 0; i < params_.size(); ++i) optimize(params_[i], state_[i], ++steps_[i]);
}
void Optimizer::zero_grad() const {
    for (const VarDiff& p : params_) p.zero_grad();
}

SGD::SGD(float lr, Penalty penalty, float momentum, float dampening, bool nesterov)
    : Optimizer(lr, penalty, momentum > 1.1920929e-7f ? 1 : 0), momentum_(momentum), dampening_(dampening), nesterov_(nesterov) {}
void SGD::step() {
    // one launch per device for all registered parameters (their updates are independent; optimizer.rs:81-86)
    std::vector<float*> w, g, v;
    std::vector<size_t> n;
    std::vector<bool> done(params_.size(), false);
    for (size_t i = 0; i < params_.size(); ++i) {
        if (done[i]) continue;
        const DevicePtr dev = params_[i].device();
        w.clear(); g.clear(); v.clear(); n.clear();
        for (size_t k = i; k < params_.size(); ++k) {
            if (done[k] || params_[k].device().get() != dev.get()) continue;
            // a parameter registered twice is updated twice, one after the other (the reference iterates its list,
            // optimizer.rs:81-86): the second registration waits for a later launch instead of racing in this one
            if (std::find(w.begin(), w.end(), params_[k].var.data->ptr()) != w.end()) continue;
            done[k] = true;
            ++steps_[k];
            HipArray& gr = params_[k].grad->borrow();
            w.push_back(params_[k].var.data->ptr()); g.push_back(gr.ptr());
            v.push_back(state_[k].empty() ? nullptr : state_[k][0]->ptr()); n.push_back(gr.len());
        }
        check(nk_sgd_step_multi(dev->raw(), (int)w.size(), w.data(), g.data(), v.data(), n.data(), lr_, momentum_, dampening_,
                                nesterov_ ? 1 : 0, penalty_.l1, penalty_.l2));
    }
}
void SGD::optimize(const VarDiff& p, std::vector<Shared<HipArray>>& st, int) {
    HipArray& g = p.grad->borrow();
    check(nk_sgd_step(p.device()->raw(), p.var.data->ptr(), g.ptr(), st.empty() ? nullptr : st[0]->ptr(), g.len(), lr_,
                      momentum_, dampening_, nesterov_ ? 1 : 0, penalty_.l1, penalty_.l2));
}

Adam::Adam(float lr, float beta1, float beta2, float eps, Penalty penalty, bool amsgrad)
    : Optimizer(lr, penalty, amsgrad ? 3 : 2), beta1_(beta1), beta2_(beta2), eps_(eps), amsgrad_(amsgrad) {}
void Adam::optimize(const VarDiff& p, std::vector<Shared<HipArray>>& st, int step) {
    HipArray& g = p.grad->borrow();
    check(nk_adam_step(p.device()->raw(), p.var.data->ptr(), g.ptr(), st[0]->ptr(), st[1]->ptr(),
                       amsgrad_ ? st[2]->ptr() : nullptr, g.len(), lr_, beta1_, beta2_, eps_, step, penalty_.l1, penalty_.l2));
}

Adagrad::Adagrad(float lr, float lr_decay, float eps, Penalty penalty)
    : Optimizer(lr, penalty, 1), lr_decay_(lr_decay), eps_(eps) {}
void Adagrad::optimize(const VarDiff& p, std::vector<Shared<HipArray>>& st, int step) {
    HipArray& g = p.grad->borrow();
    check(nk_adagrad_step(p.device()->raw(), p.var.data->ptr(), g.ptr(), st[0]->ptr(), g.len(), lr_, lr_decay_, eps_, step,
                          penalty_.l1, penalty_.l2));
}

RMSProp::RMSProp(float lr, float alpha, float eps, float momentum, bool centered, Penalty penalty)
    : Optimizer(lr, penalty, 3), alpha_(alpha), eps_(eps), momentum_(momentum), centered_(centered) {}
void RMSProp::optimize(const VarDiff& p, std::vector<Shared<HipArray>>& st, int) {
    HipArray& g = p.grad->borrow();
    const bool mom = momentum_ > 1.1920929e-7f;
    check(nk_rmsprop_step(p.device()->raw(), p.var.data->ptr(), g.ptr(), st[0]->ptr(), centered_ ? st[1]->ptr() : nullptr,
                          mom ? st[2]->ptr() : nullptr, g.len(), lr_, alpha_, eps_, momentum_, penalty_.l1, penalty_.l2));
}

namespace lr_scheduler {
void LRScheduler::step() {
    last_lr_ = current_lr_;  // prepare_step, lr_scheduler/mod.rs:51-59
    epoch_ += 1;
    float lr = current_lr_;
    if (update(lr)) {
        current_lr_ = lr;
        opt_.set_lr(current_lr_);
    }
}
bool StepLR::update(float& lr) {
    if (step_size_ == 0) panic("attempt to calculate the remainder with a divisor of zero");  // `rem_euclid(0)`
    if (epoch_ % step_size_ != 0) return false;
    lr = last_lr_ * gamma_;
    return true;
}
bool MultiStepLR::update(float& lr) {
    if (std::find(milestones_.begin(), milestones_.end(), epoch_) == milestones_.end()) return false;
    lr = last_lr_ * gamma_;
    return true;
}
}  // namespace lr_scheduler

}  // namespace optim

// =================================================================================================
// dp
// =================================================================================================
namespace dp {

std::string Communicator::unique_id() {
    std::string id(NK_COMM_ID_BYTES, '\0');
    check(nk_comm_unique_id(&id[0]));
    return id;
}
Communicator::Communicator(DevicePtr dev, int nranks, int rank, const std::string& id)
    : dev_(std::move(dev)), rank_(rank), size_(nranks) {
    if (id.size() != NK_COMM_ID_BYTES) panic("communicator id must be 128 bytes");
    check(nk_comm_init_rank(dev_->raw(), nranks, rank, id.data(), &h_));
}
Communicator::Communicator(DevicePtr dev, int nranks, int channels, double gbps) : dev_(std::move(dev)), rank_(0), size_(nranks) {
    check(nk_comm_init_replicas(dev_->raw(), nranks, channels, gbps, &h_));
}
std::shared_ptr<Communicator> Communicator::replicas(DevicePtr dev, int nranks, int channels, double gbps) {
    return std::shared_ptr<Communicator>(new Communicator(std::move(dev), nranks, channels, gbps));
}
Communicator::~Communicator() { nk_comm_destroy(h_); }

GradientSync::GradientSync(std::shared_ptr<Communicator> comm, const std::vector<VarDiff>& params, size_t small_elems)
    : comm_(std::move(comm)), small_elems_(small_elems) {
    for (const VarDiff& p : params) {
        if (!params_.emplace(p.grad.get(), p.grad).second) continue;  // registered twice: one exchange
        const size_t len = numel(p.shape());
        bytes_ += len * sizeof(float);
        n_small_ += len < small_elems_;
        for (int k = 0; k < 2; ++k) {  // up to two pieces per gradient in flight
            nk_event* ev = nullptr;
            check(nk_event_create(comm_->device()->raw(), &ev));
            events_.push_back(ev);
        }
    }
    nk_event* ev = nullptr;  // one more for the group of small gradients
    check(nk_event_create(comm_->device()->raw(), &ev));
    events_.push_back(ev);
}
GradientSync::~GradientSync() {
    if (busy_told_) nk_device_set_busy_slots(comm_->device()->raw(), 0);
    for (nk_event* e : events_) nk_event_destroy(e);
}
bool GradientSync::wants_parts(const Gradient* g) const {
    if (!active()) return false;
    auto it = params_.find(g);
    if (it == params_.end()) return false;
    // a small gradient is always welcome early (it is handed over whole: nothing is split for it) - the group of small
    // gradients leaves as soon as the last of them is final
    if (numel(it->second->shape()) < small_elems_) return parts_ != Parts::None;
    switch (parts_) {
        case Parts::All: return true;
        case Parts::None: return false;
        default: return g == split_next_;
    }
}
void GradientSync::flush_small() {
    if (small_pending_.empty()) return;
    std::vector<float*> bufs;
    std::vector<size_t> counts;
    for (const Gradient* g : small_pending_) {
        HipArray& a = params_.at(g)->borrow();
        bufs.push_back(a.ptr());
        counts.push_back(a.len());
        elems_ += a.len();
    }
    small_pending_.clear();
    nk_event* ev = events_[next_event_++ % events_.size()];
    check(nk_event_record(ev, 0));  // every pending small gradient was final before this point of the compute stream
    check(nk_allreduce_sum_group_async(comm_->raw(), bufs.data(), counts.data(), (int)bufs.size(), ev));
    ++issued_;
}
void GradientSync::grad_part_ready(const Gradient* g, size_t offset, size_t count) {
    auto it = params_.find(g);
    if (it == params_.end() || !active() || count == 0) return;
    HipArray& a = it->second->borrow();
    if (offset + count > a.len()) panic("grad_part_ready: piece exceeds the gradient");
    if (parts_done_[g] + count > a.len()) panic("grad_part_ready: pieces overlap");
    parts_done_[g] += count;
    if (a.len() < small_elems_) {  // a small gradient travels whole, with the group
        if (count != a.len()) panic("grad_part_ready: a small gradient must be handed over whole");
        small_pending_.push_back(g);
        if (small_pending_.size() == n_small_) flush_small();
        return;
    }
    last_large_ = g;
    nk_event* ev = events_[next_event_++ % events_.size()];
    check(nk_event_record(ev, 0));  // everything up to the launch that finished this piece
    check(nk_allreduce_sum_async(comm_->raw(), a.ptr() + offset, count, ev));
    elems_ += count;
    ++issued_;
    if (busy_slots_ > 0 && !busy_told_) {  // the exchange's workgroups share the chip with every launch from here to join()
        check(nk_device_set_busy_slots(comm_->device()->raw(), busy_slots_));
        busy_told_ = true;
    }
}
void GradientSync::grad_ready(const Gradient* g) {
    auto it = params_.find(g);
    if (it == params_.end()) return;
    if (!active()) return;  // nothing to exchange
    HipArray& a = it->second->borrow();
    auto pd = parts_done_.find(g);
    if (pd != parts_done_.end() && pd->second != 0) {
        const size_t done = pd->second;
        pd->second = 0;
        if (done == a.len()) return;  // already handed over piece by piece
        panic("GradientSync: a gradient was only partly exchanged piecewise");
    }
    grad_part_ready(g, 0, a.len());
    parts_done_[g] = 0;
}
void GradientSync::join() {
    flush_small();  // small gradients of parameters some of whose siblings never became final this pass
    split_next_ = last_large_;  // (every rank runs the same tape: the same gradient on every rank, so the collectives still pair up)
    last_large_ = nullptr;
    next_event_ = 0;
    for (auto& kv : parts_done_) kv.second = 0;
    if (busy_told_) {
        check(nk_device_set_busy_slots(comm_->device()->raw(), 0));
        busy_told_ = false;
    }
    if (active()) check(nk_comm_join(comm_->raw()));
}

void all_reduce_gradients(const Communicator& comm, const std::vector<VarDiff>& params) {
    if (comm.size() == 1) return;
    for (const VarDiff& p : params) {
        HipArray& a = p.grad->borrow();
        check(nk_allreduce_sum_async(comm.raw(), a.ptr(), a.len(), nullptr));
    }
    check(nk_comm_join(comm.raw()));
}

}  // namespace dp

}  // namespace neuronika
