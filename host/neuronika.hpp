// Host-side mirror of neuronika's Var / VarDiff tape for the HIP backend (C++17).
//
// The reference's host code is Rust; no Rust toolchain exists in this environment, so the
// host side above the C ABI (include/neuronika_hip.h) is written in C++ with the reference's
// own names, argument meaning and error behaviour:
//
//   reference (neuronika-variable/src)              here
//   ------------------------------------------------------------------------------------------
//   autograd.rs:7-25    trait Forward / Backward     struct Forward / Backward
//   gradient.rs:8-79    NoGrad, Gradient             struct NoGrad, class Gradient
//   history.rs:9-125    HistoryId, History<T>        class History<T>
//   var.rs:34-128       Var<D>                       class Var   (rank is a run-time Shape)
//   vardiff.rs:35-168   VarDiff<D>                   class VarDiff
//   cuda/device.rs, cuda/cuarray.rs                  class Device, class HipArray
//   lib.rs:29-36        enum Reduction               enum class Reduction
//   neuronika-nn/src/lib.rs:406-448  Linear          nn::Linear   (+ Conv2d, Dropout, MHA)
//   neuronika-optim/src/optimizer.rs, sgd/mod.rs     optim::SGD
//   (net-new)                                         dp::Communicator / dp::GradientSync
//
// Semantics kept from the reference: graph construction allocates ZEROED output / gradient
// buffers and computes nothing; `.forward()` runs the forward tape in insertion order and
// overwrites every node output; `.backward(seed)` fills the root gradient with `seed` and runs
// the backward tape in reverse, every node accumulating (`+=`) into its operands' gradients;
// intermediate gradients are NOT re-zeroed between calls (vardiff.rs:125-141) — use
// `no_grad()` + `with_grad()` (gradient.rs:64-79) to drop and re-create them.
// Errors: the reference panics; here `neuronika::Panic` (a std::runtime_error) is thrown.
// Threading: like `Rc<RefCell<..>>` graphs, one host thread per graph/device.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "neuronika_hip.h"

namespace neuronika {

using Shape = std::vector<int>;

struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] void panic(const std::string& msg);
void check(int status);  // non-zero C-ABI status -> Panic(nk_last_error())

size_t numel(const Shape& s);

// ---------------------------------------------------------------------------------------------
// Device + device array (reference template: cuda/device.rs:11-58, cuda/cuarray.rs:10-117)
// ---------------------------------------------------------------------------------------------
class Device : public std::enable_shared_from_this<Device> {
   public:
    static std::shared_ptr<Device> create(int idx);
    ~Device();
    nk_device* raw() const { return h_; }
    int index() const { return idx_; }
    void sync() const;
    // Stream-ordered caching allocator: blocks released by graph buffers are re-used for later
    // allocations of the same size (all work runs in order on the compute stream).
    float* alloc_zeroed(size_t n);
    float* alloc_uninit(size_t n);  // contents undefined (a gradient whose zero fill is still pending)
    void release(float* p, size_t n);
    size_t bytes_in_use() const { return in_use_; }
    // hipGraph capture of a launch-bound step (see nk_graph_begin in the C header for the rules)
    void graph_begin();
    std::shared_ptr<class Graph> graph_end();

   private:
    Device() = default;
    nk_device* h_ = nullptr;
    int idx_ = 0;
    std::unordered_map<size_t, std::vector<float*>> pool_;
    size_t in_use_ = 0;
};
using DevicePtr = std::shared_ptr<Device>;

class Graph {  // a captured, replayable sequence of launches on one device
   public:
    Graph(nk_graph* g, std::shared_ptr<Device> dev) : dev_(std::move(dev)), g_(g) {}
    ~Graph();
    Graph(const Graph&) = delete;
    Graph& operator=(const Graph&) = delete;
    void launch() const;

   private:
    std::shared_ptr<Device> dev_;  // a graph keeps its device alive (captured kernels hold the device's workspace addresses)
    nk_graph* g_;
};

class HipArray {
   public:
    HipArray(DevicePtr dev, Shape shape);  // `CuArray::zeroed`
    struct Uninit {};
    HipArray(DevicePtr dev, Shape shape, Uninit);  // undefined contents
    // `shape` elements of `parent` starting `offset` elements in: shares the parent's buffer and keeps it alive (the packed
    // projection weights of nn::MultiheadAttention: three parameters, one allocation)
    HipArray(std::shared_ptr<HipArray> parent, size_t offset, Shape shape);
    ~HipArray();
    HipArray(const HipArray&) = delete;
    HipArray& operator=(const HipArray&) = delete;

    static std::shared_ptr<HipArray> from_host(DevicePtr dev, const Shape& shape, const float* host);
    void upload(const float* host);              // `CuArray::from_ndarray`
    void download(float* host) const;            // `CuArray::as_ndarray`
    std::vector<float> to_vec() const;
    void fill(float v);
    size_t len() const { return len_; }
    const Shape& shape() const { return shape_; }
    float* ptr() const { return ptr_; }
    const DevicePtr& device() const { return dev_; }

   private:
    DevicePtr dev_;
    Shape shape_;
    size_t len_;
    float* ptr_;
    std::shared_ptr<HipArray> parent_;  // set for a view: the buffer belongs to the parent
};
template <class T>
using Shared = std::shared_ptr<T>;  // utils.rs:9  `Shared<T> = Rc<RefCell<T>>`

// ---------------------------------------------------------------------------------------------
// autograd.rs / gradient.rs
// ---------------------------------------------------------------------------------------------
struct Forward {
    virtual ~Forward() = default;
    virtual void forward() const = 0;
};
class Gradient;
struct Backward {
    virtual ~Backward() = default;
    virtual void backward() const = 0;
    // gradients this node accumulates into (used to tell when a leaf gradient is final)
    virtual void targets(std::vector<const Gradient*>& out) const = 0;
    // the subset of `targets` this node can write PRE-MASKED when the gradient asks for it (`Gradient::premask_source`):
    // a Linear node's input gradient, whose GEMM epilogue applies the ReLU mask of the node that produced the input
    virtual void premask_targets(std::vector<const Gradient*>&) const {}
};
// Called by VarDiff::backward right after the LAST tape node writing into a gradient has been
// issued: the hook of the data-parallel exchange (dp::GradientSync).
struct BackwardHook {
    virtual ~BackwardHook() = default;
    virtual void grad_ready(const Gradient* g) = 0;
    // Optional finer grain: a node that finishes a gradient before its own `backward()` returns (Linear: the bias
    // gradient first, then the weight gradient in row blocks) reports each finished contiguous piece, so the exchange
    // starts while the node is still issuing launches.  A node obtains the hook through `parts_hook(g)` only, which
    // answers null unless THIS node is the last writer of g on the tape (a shared / tied parameter keeps accumulating
    // in a later node; its exchange must wait for `grad_ready`).  `grad_ready` still follows once the node is done.
    virtual bool wants_parts(const Gradient*) const { return false; }
    virtual void grad_part_ready(const Gradient*, size_t /*offset*/, size_t /*count*/) {}
};
// The hook of the backward pass being issued on this thread if it wants pieces of `g` AND the node now running is
// the last tape node that writes g; null otherwise (also outside `backward(seed, hook)`).
BackwardHook* parts_hook(const Gradient* g);
struct NoGrad {
    virtual ~NoGrad() = default;
    virtual void no_grad() = 0;
    virtual void with_grad() = 0;
};

class Gradient : public NoGrad {
   public:
    // `ndarray_zeros` (gradient.rs:47-54).  The zero fill is LAZY: the buffer is allocated without a memset and
    // `zero_pending()` stays true until somebody looks at it.  `borrow()` (read / `+=` access) materialises the
    // zeros first; `borrow_first_write(assign)` hands a pending fill to a node that can ASSIGN instead of `+=`
    // (`0 + v`, same values, no memset and no read of the destination).
    Gradient(DevicePtr dev, Shape shape);
    // A gradient whose buffer is `shape` elements of `storage` starting at `offset` (packed parameters: the gradients of
    // several parameters in one allocation, so that ONE GEMM can write them all).  Same lazy zero fill, per view.
    Gradient(Shared<HipArray> storage, size_t offset, Shape shape);
    HipArray& borrow() const;              // panics when de-allocated
    HipArray& borrow_first_write(bool& assign) const;
    void zero();                           // `zero_grad` (vardiff.rs:100-102), lazily
    bool zero_pending() const { return pending_zero_; }
    Shared<HipArray> array() const { return array_; }
    // true for a view gradient over `storage` starting `offset` elements in (packed parameters)
    bool is_view_of(const Shared<HipArray>& storage, size_t offset) const { return storage_ && storage_.get() == storage.get() && offset_ == offset; }
    // Stand `external` in for this gradient's buffer (returns the previous one) without touching device memory: how
    // `VarDiff::backward_from` presents an upstream gradient tensor to the root's backward nodes (they read through
    // `borrow()` at run time).  The buffer counts as written.
    Shared<HipArray> exchange_array(Shared<HipArray> external);
    const Shape& shape() const { return shape_; }
    void no_grad() override;
    void with_grad() override;
    // Fused Linear+ReLU (`nn::Linear::forward_relu`): the gradient of its output y holds dL/dz of the PRE-activation.  The
    // node names y as the mask source; `VarDiff::backward` decides per pass (`set_premasked`) whether every node that
    // writes this gradient on the tape can apply `(y > 0) *` itself while storing (`Backward::premask_targets`); if not,
    // the writers store plain contributions and the owner masks the buffer in place before using it (idempotent).
    void set_premask_source(Shared<HipArray> y) { premask_src_ = std::move(y); }
    const Shared<HipArray>& premask_source() const { return premask_src_; }
    void set_premasked(bool on) const { premasked_ = on; }
    bool premasked() const { return premasked_; }

   private:
    DevicePtr dev_;
    Shape shape_;
    Shared<HipArray> array_;
    mutable bool pending_zero_ = true;
    Shared<HipArray> premask_src_;
    mutable bool premasked_ = false;
    Shared<HipArray> storage_;  // set for a view gradient
    size_t offset_ = 0;
};

// ---------------------------------------------------------------------------------------------
// history.rs — the tape.  Entries are identified by node address and ordered by the size of
// the history at insertion; merging keeps one entry per address.
// ---------------------------------------------------------------------------------------------
template <class T>
class History {
   public:
    void merge(const History& other);
    void insert(const void* ptr, T op);
    size_t len() const { return path_.size(); }
    size_t buffer_len() const { return buffer_->size(); }
    std::vector<T> to_vec() const;
    std::vector<T>& buffer_mut() const { return *buffer_; }

   private:
    struct Item {
        const void* ptr;
        size_t order;
        T op;
    };
    std::vector<Item> path_;  // sorted by order (stable)
    // the reference clones the RefCell<Vec> with the History; a fresh buffer per copy is
    // equivalent because insert() truncates it and forward() repopulates an empty one.
    std::shared_ptr<std::vector<T>> buffer_ = std::make_shared<std::vector<T>>();

   public:
    History() = default;
    History(const History& o) : path_(o.path_), buffer_(std::make_shared<std::vector<T>>(*o.buffer_)) {}
    History& operator=(const History& o) {
        path_ = o.path_;
        buffer_ = std::make_shared<std::vector<T>>(*o.buffer_);
        return *this;
    }
};

struct ForwardEntry {
    Shared<Forward> op;
    Shared<bool> computed;
};
struct BackwardEntry {
    Shared<Backward> op;
    Shared<NoGrad> grad;
};

enum class Reduction { Sum = 0, Mean = 1 };  // lib.rs:29-36

class VarDiff;

// ---------------------------------------------------------------------------------------------
// var.rs — non-differentiable variable
// ---------------------------------------------------------------------------------------------
// `PaddingMode` implementors (node/pad/{zero,constant,reflective,replicative}/mod.rs)
struct PaddingMode {
    enum Kind { Zero, Constant, Reflective, Replicative } kind = Zero;
    float value = 0.f;
    static PaddingMode zero() { return {Zero, 0.f}; }
    static PaddingMode constant(float v) { return {Constant, v}; }
    static PaddingMode reflective() { return {Reflective, 0.f}; }
    static PaddingMode replicative() { return {Replicative, 0.f}; }
};

class Var {
   public:
    Shared<HipArray> data;
    History<ForwardEntry> history;

    static Var leaf(Shared<HipArray> array);                                          // var.rs:46
    static Var node(Shared<HipArray> data, Shared<Forward> op, History<ForwardEntry> h);  // var.rs:53
    const Shape& shape() const { return data->shape(); }
    DevicePtr device() const { return data->device(); }

    VarDiff requires_grad() const;  // var.rs:103
    void forward() const;           // var.rs:110-128
    float item() const;             // var.rs:133
    std::vector<float> to_vec() const { return data->to_vec(); }

    Var sum() const;                                  // var.rs:201
    Var mean() const;                                 // var.rs:209
    Var relu() const;                                 // var.rs:243
    // pointwise nodes (var.rs:222-302) + negation (`-x`) and unsqueeze (var.rs:425)
    Var neg() const;
    Var pow(int exp) const;
    Var sqrt() const;
    Var leaky_relu() const;
    Var softplus() const;
    Var sigmoid() const;
    Var tanh() const;
    Var ln() const;
    Var exp() const;
    Var unsqueeze(int axis) const;
    Var softmax(int axis) const;                      // var.rs:318
    Var log_softmax(int axis) const;                  // var.rs:338
    Var t() const;                                    // var.rs:347
    Var dropout(double p, Shared<bool> status) const; // var.rs:375
    std::vector<Var> chunks(const Shape& chunk_size) const;             // var.rs:401
    Var cat(const std::vector<Var>& variables, int axis) const;         // var.rs:564
    Var stack(const std::vector<Var>& variables, int axis) const;       // var.rs:614 (MultiStack)
    Var mse(const Var& target, Reduction reduction) const;              // var.rs:454
    Var mae(const Var& target, Reduction reduction) const;              // var.rs:433
    Var bce(const Var& target, Reduction reduction) const;              // var.rs:484
    Var bce_with_logits(const Var& target, Reduction reduction) const;  // var.rs:513
    Var kldiv(const Var& target, Reduction reduction) const;            // var.rs:542
    Var nll(const Var& target, Reduction reduction) const;              // var.rs:671
    Var mv(const Var& rhs) const;                                        // var.rs:1098 (matrix . vector)
    VarDiff mv(const VarDiff& rhs) const;
    Var vm(const Var& rhs) const;                                        // var.rs:1129 (vector . matrix)
    VarDiff vm(const VarDiff& rhs) const;
    Var vv(const Var& rhs) const;                                        // var.rs:1160 (dot product)
    VarDiff vv(const VarDiff& rhs) const;
    Var pad(const std::vector<int>& padding, float value = 0.f) const;  // var.rs:726 (Zero/Constant)
    Var pad(const std::vector<int>& padding, PaddingMode mode) const;   // any `PaddingMode` (pad/mod.rs:17-27)
    Var mm(const Var& rhs) const;                                        // var.rs:1034
    VarDiff mm(const VarDiff& rhs) const;
    Var mm_t(const Var& rhs) const;                                      // var.rs:1065
    VarDiff mm_t(const VarDiff& rhs) const;                              // var.rs:1081
    // `kernel.convolution(input, stride, dilation, groups)`  var.rs:1296-1371
    Var convolution(const Var& input, const std::vector<int>& stride, const std::vector<int>& dilation,
                    int groups) const;
    // composed-attention glue (Chunk((S,dh)) tiles / cat): SURVEY.md 8a note
    Var split_heads(int B, int S, int H, int dh) const;
    Var merge_heads(int B, int S, int H, int dh) const;
    // batched (b,h) matrix products over [B*H, S, *] tiles = B*H `mm` / `mm_t` nodes
    Var bmm(const Var& rhs) const;
    Var bmm_t(const Var& rhs) const;
    // the same per-(batch, head) products taken DIRECTLY on the (B*S, H*dh) projection layout (strided GEMM operands:
    // row stride H*dh, head offset h*dh) - no head split / merge copies:
    //   heads_scores : self = Q (B*S, H*dh), keys K (B*S, H*dh)      -> (B*H, S, S)   Q_bh . K_bh^T
    //   heads_context: self = P (B*H, S, S), values V (B*S, H*dh)    -> (B*S, H*dh)   P_bh . V_bh
    Var heads_scores(const Var& keys, int B, int S, int H, int dh) const;
    Var heads_context(const Var& values, int B, int S, int H, int dh) const;
    // dropout(softmax(self * scale, last axis), p): the Multiplication + Softmax + Dropout nodes of
    // the attention probabilities as ONE node (same values, one pass over the score tensor)
    // (store_probs = false: the backward pass recomputes the probabilities from the scores, bit-identically)
    Var attention_probs(float scale, double p, Shared<bool> status, bool store_probs = false) const;
    // the whole per-(sample, head) chain heads_scores -> attention_probs -> heads_context as ONE node on the fused
    // attention kernels (nk_attention_fwd): self = Q, all three operands (B*S, H*dh) -> (B*S, H*dh).  The score tile
    // stays on chip between the two products; see attention_core_supported() for the shapes it takes.
    Var heads_attention(const Var& keys, const Var& values, int B, int S, int H, int dh, float scale, double p,
                        Shared<bool> status) const;
    static bool attention_core_supported(int S, int dh, double p);
};

// ---------------------------------------------------------------------------------------------
// vardiff.rs — differentiable variable
// ---------------------------------------------------------------------------------------------
namespace nn { struct LinearOrigin; }
class VarDiff {
    void run_backward(BackwardHook* hook) const;

   public:
    Var var;
    Shared<Gradient> grad;
    History<BackwardEntry> history;
    // Set on the output of a fused `nn::Linear::forward`: how to build the same product again.  `relu()` on such a
    // variable AS A TEMPORARY - the reference's spelling `lin.forward(x).relu()` (neuronika-nn/src/lib.rs:441-447; Rust's
    // `relu(self)` CONSUMES its operand, vardiff.rs:282-288) - answers the ONE Linear+ReLU node over the Linear's own operands
    // instead of a ReLU node over this output (graph-build peephole, `nn::set_relu_peephole`): the consumed pre-activation is
    // not observable in the reference either.  On a variable that is KEPT (`h = lin.forward(x); y = h.relu()` - Rust's
    // `h.clone().relu()`) `relu()` is the ReLU node over h: h is in y's history, its data and gradient are what the reference shows.
    Shared<const nn::LinearOrigin> linear_origin;

    static VarDiff leaf(Var var, Shared<Gradient> grad);                                        // vardiff.rs:48
    static VarDiff node(Var var, Shared<Gradient> grad, BackwardEntry op, History<BackwardEntry> h);  // :56
    const Shape& shape() const { return var.shape(); }
    DevicePtr device() const { return var.device(); }
    Shared<HipArray> data() const { return var.data; }

    void zero_grad() const;         // vardiff.rs:100
    void forward() const;           // vardiff.rs:106-116
    void backward(float seed, BackwardHook* hook = nullptr) const;  // vardiff.rs:125-141
    // The same pass seeded with an upstream gradient TENSOR (root gradient = `seed`, same shape) instead of a
    // scalar fill: what an enclosing graph would hand to this sub-graph's root.  Lets a benchmark time a module's
    // own backward nodes without `(y * G).sum()` scaffolding.  The seed's buffer stands in for the root gradient while
    // the tape runs (no copy); afterwards the root gradient is its own buffer again, holding what it held before.
    void backward_from(const Var& seed, BackwardHook* hook = nullptr) const;
    void no_grad() const;           // vardiff.rs:145
    void with_grad() const;         // vardiff.rs:157
    float item() const { return var.item(); }
    std::vector<float> to_vec() const { return var.to_vec(); }
    std::vector<float> grad_to_vec() const { return grad->borrow().to_vec(); }

    VarDiff sum() const;
    VarDiff mean() const;
    VarDiff relu() const&;  // the ReLU node over this (kept) variable
    VarDiff relu() &&;      // `relu(self)` on a temporary: may fold into the Linear that produced it (see `linear_origin`)
    VarDiff neg() const;
    VarDiff pow(int exp) const;
    VarDiff sqrt() const;
    VarDiff leaky_relu() const;
    VarDiff softplus() const;
    VarDiff sigmoid() const;
    VarDiff tanh() const;
    VarDiff ln() const;
    VarDiff exp() const;
    VarDiff unsqueeze(int axis) const;
    VarDiff softmax(int axis) const;
    VarDiff log_softmax(int axis) const;
    VarDiff t() const;
    VarDiff dropout(double p, Shared<bool> status) const;
    std::vector<VarDiff> chunks(const Shape& chunk_size) const;
    VarDiff cat(const std::vector<VarDiff>& vars, int axis) const;
    VarDiff stack(const std::vector<VarDiff>& vars, int axis) const;
    VarDiff mse(const Var& target, Reduction reduction) const;
    VarDiff mae(const Var& target, Reduction reduction) const;              // vardiff.rs:474
    VarDiff bce(const Var& target, Reduction reduction) const;              // vardiff.rs:525
    VarDiff bce_with_logits(const Var& target, Reduction reduction) const;  // vardiff.rs:554
    VarDiff kldiv(const Var& target, Reduction reduction) const;            // vardiff.rs:583
    VarDiff nll(const Var& target, Reduction reduction) const;              // vardiff.rs:725
    VarDiff mv(const Var& rhs) const;
    VarDiff mv(const VarDiff& rhs) const;
    VarDiff vm(const Var& rhs) const;
    VarDiff vm(const VarDiff& rhs) const;
    VarDiff vv(const Var& rhs) const;
    VarDiff vv(const VarDiff& rhs) const;
    VarDiff pad(const std::vector<int>& padding, float value = 0.f) const;
    VarDiff pad(const std::vector<int>& padding, PaddingMode mode) const;
    VarDiff mm(const Var& rhs) const;
    VarDiff mm(const VarDiff& rhs) const;
    VarDiff mm_t(const Var& rhs) const;
    VarDiff mm_t(const VarDiff& rhs) const;
    VarDiff convolution(const Var& input, const std::vector<int>& stride, const std::vector<int>& dilation,
                        int groups) const;
    VarDiff convolution(const VarDiff& input, const std::vector<int>& stride, const std::vector<int>& dilation,
                        int groups) const;
    VarDiff split_heads(int B, int S, int H, int dh) const;
    VarDiff merge_heads(int B, int S, int H, int dh) const;
    VarDiff bmm(const VarDiff& rhs) const;
    VarDiff bmm_t(const VarDiff& rhs) const;
    VarDiff heads_scores(const VarDiff& keys, int B, int S, int H, int dh) const;
    VarDiff heads_context(const VarDiff& values, int B, int S, int H, int dh) const;
    VarDiff attention_probs(float scale, double p, Shared<bool> status, bool store_probs = false) const;
    VarDiff heads_attention(const VarDiff& keys, const VarDiff& values, int B, int S, int H, int dh, float scale, double p,
                            Shared<bool> status) const;
};

// `Add/Sub/Mul/Div` with NumPy broadcasting, all four differentiability combinations
// (var.rs:859-1026, vardiff.rs:866-1069) and the f32 scalar forms (var.rs:746-838).
#define NK_DECLARE_BINARY(OP)                         \
    Var operator OP(const Var& l, const Var& r);      \
    VarDiff operator OP(const Var& l, const VarDiff& r);   \
    VarDiff operator OP(const VarDiff& l, const Var& r);   \
    VarDiff operator OP(const VarDiff& l, const VarDiff& r); \
    Var operator OP(const Var& l, float r);           \
    VarDiff operator OP(const VarDiff& l, float r);
NK_DECLARE_BINARY(+)
NK_DECLARE_BINARY(-)
NK_DECLARE_BINARY(*)
NK_DECLARE_BINARY(/)
#undef NK_DECLARE_BINARY

// leaf constructors (lib.rs:51-143)
Var from_host(DevicePtr dev, const Shape& shape, const float* host);  // `from_ndarray`
// lib.rs:160-240: `eye`, `linspace` (end inclusive), `logspace` (base^linspace, negative base -> negative values),
// `geomspace` (panics where the reference returns None: zero endpoint or endpoints of different sign), `range` (half open)
Var eye(DevicePtr dev, int n);
Var linspace(DevicePtr dev, float start, float end, int n);
Var logspace(DevicePtr dev, float base, float start, float end, int n);
Var geomspace(DevicePtr dev, float start, float end, int n);
Var range(DevicePtr dev, float start, float end, float step);
Var zeros(DevicePtr dev, const Shape& shape);
Var ones(DevicePtr dev, const Shape& shape);
Var full(DevicePtr dev, const Shape& shape, float value);
Var rand(DevicePtr dev, const Shape& shape, uint64_t seed);  // U[0,1) (host RNG, uploaded)

// ---------------------------------------------------------------------------------------------
// neuronika-nn
// ---------------------------------------------------------------------------------------------
// Restart the per-thread sequence of Philox keys handed to random nodes: the next Dropout / attention-probabilities
// node built on this thread uses key `seed`, the one after it seed + 0x632BE59BD9B4E019, ...
void manual_seed(uint64_t seed);

namespace nn {

// neuronika-nn/src/init.rs: parameter initialisers.  Values are produced on the host and uploaded (one-off work at
// model construction).  The random ones take an explicit seed (the reference draws from `thread_rng`).
namespace init {
float calculate_gain(const std::string& non_linearity);                 // init.rs:25-33
std::pair<float, float> calculate_fan_in_fan_out(const VarDiff& param);  // init.rs:45-64 (trailing extents are SUMMED there)
void constant(const VarDiff& param, float value);                        // :74
void zeros(const VarDiff& param);                                        // :83
void ones(const VarDiff& param);                                         // :92
void eye(const VarDiff& param);                                          // :104
void dirac(const VarDiff& param, int groups);                            // :131-170
void uniform(const VarDiff& param, float low, float high, uint64_t seed);            // :177
void normal(const VarDiff& param, float mean, float std, uint64_t seed);             // :195
void xavier_uniform(const VarDiff& param, float gain, uint64_t seed);                // :213
void xavier_normal(const VarDiff& param, float gain, uint64_t seed);                 // :236
}  // namespace init

// `Linear` neuronika-nn/src/lib.rs:406-448: weight (out,in), bias (out), U(-k,k), k = 1/sqrt(in)
struct Linear {
    VarDiff weight, bias;
    bool fused = true;  // bias added in the GEMM epilogue (one node); false: the reference's two nodes mm_t, +
    Linear(DevicePtr dev, int in_features, int out_features, uint64_t seed);
    Linear(VarDiff weight, VarDiff bias) : weight(std::move(weight)), bias(std::move(bias)) {}
    VarDiff forward(const Var& input) const;      // input.mm_t(W) + b
    VarDiff forward(const VarDiff& input) const;
    // `forward(input).relu()` as ONE node (ours; the reference composes the two): max(input.W^T + b, 0) from the GEMM
    // epilogue, the pre-activation never stored; in the backward pass the ReLU mask is applied by whichever GEMM produces
    // this node's output gradient (a following Linear's input-gradient GEMM), or in place when another kind of node does.
    // Same values and gradients as the two nodes, bit for bit; the one observable difference: in a pass where every writer
    // of the OUTPUT's gradient applied the mask while storing (a following Linear), the output's `grad()` holds
    // (y > 0) * dL/dy afterwards - the entries where y = 0 read 0.  (A root's gradient, and any gradient written by other
    // kinds of nodes, keeps dL/dy: the mask then goes into a scratch copy.)
    VarDiff forward_relu(const Var& input) const;
    VarDiff forward_relu(const VarDiff& input) const;
};
// What `VarDiff::relu()` needs to rebuild a fused Linear as Linear+ReLU (see `VarDiff::linear_origin`)
struct LinearOrigin {
    VarDiff weight, bias;
    Var input;
    bool differentiable_input = false;       // the input was a VarDiff: its gradient and tape follow
    Shared<Gradient> input_grad;
    History<BackwardEntry> input_history;
};
// Graph-build peephole `forward(x).relu()` -> the Linear+ReLU node (on by default, per thread; off: the ReLU node over the
// Linear's output, as the tests that pit the two graphs against each other bit for bit need it).  Returns the old setting.
bool set_relu_peephole(bool on);

// `LSTMCell` neuronika-nn/src/lib.rs:453-541.  Weights (4H,in)/(4H,H), biases (4H), U(-k,k), k = 1/sqrt(H).
// `forward` keeps the reference's exact composition: state = (cell_state, hidden); gate chunks 0..3 get
// sigmoid, tanh, sigmoid, sigmoid (:528-533); returns (new_cell_state, new_hidden) (:534-537).
struct LSTMCell {
    VarDiff weight_ih, weight_hh, bias_ih, bias_hh;
    LSTMCell(DevicePtr dev, int input_size, int hidden_size, uint64_t seed);
    std::pair<VarDiff, VarDiff> forward(const std::pair<VarDiff, VarDiff>& state, const Var& input) const;
    std::pair<VarDiff, VarDiff> forward(const std::pair<VarDiff, VarDiff>& state, const VarDiff& input) const;
};

// `GRUCell` neuronika-nn/src/lib.rs:543-625.  Weights (3H,in)/(3H,H), biases (3H); forward :602-624.
struct GRUCell {
    VarDiff weight_ih, weight_hh, bias_ih, bias_hh;
    GRUCell(DevicePtr dev, int input_size, int hidden_size, uint64_t seed);
    VarDiff forward(const VarDiff& hidden, const Var& input) const;
    VarDiff forward(const VarDiff& hidden, const VarDiff& input) const;
};

// `Conv1d` / `Conv2d` / `Conv3d` struct/new neuronika-nn/src/lib.rs:630-722, 724-814, 816-916: weight
// (Cout, Cin/groups, k...), bias (Cout, 1...), U(-k,k) with k = sqrt(1/(Cin*prod(kernel))).  Their forward
// is `todo!()` in the reference snapshot (:716-721, :809-814, :908-914) and is defined here as
// pad(padding, padding_mode) -> convolution(stride, dilation, groups) -> + bias.
struct ConvNd {
    VarDiff weight, bias;
    std::vector<int> padding, stride, dilation;
    PaddingMode padding_mode;
    int groups = 1;
    bool fused = true;  // bias added in the convolution epilogue (one node); false: convolution node + Addition node
    // Zero padding folded into the forward and the kernel gradient where the library's kernels can read the unpadded input
    // (nk_conv_padding_folds: the Winograd geometries at sizes the rules give to those kernels): no Pad node, no padded copy
    // (C3: 110 MB and a 40 us kernel per step less); false: the padded copy as a forward-only node (what rounds 2 - 4 built)
    bool fold_padding = true;
    ConvNd(int nd, DevicePtr dev, int in_channels, int out_channels, std::vector<int> kernel, std::vector<int> padding,
           PaddingMode mode, std::vector<int> stride, std::vector<int> dilation, int groups, uint64_t seed);
    VarDiff forward(const Var& input) const;
    VarDiff forward(const VarDiff& input) const;
};
// Constructor argument order = the reference's `new` (neuronika-nn/src/lib.rs:671-679, 762-770, 857-865):
// (in_channels, out_channels, kernel_size, padding, padding_mode, stride, dilation); `seed` (ours, last) fixes the
// U(-k, k) initialisation.
struct Conv1d : ConvNd {
    Conv1d(DevicePtr dev, int in_channels, int out_channels, int kernel, int padding, PaddingMode mode, int stride,
           int dilation, uint64_t seed = 0)
        : ConvNd(1, std::move(dev), in_channels, out_channels, {kernel}, {padding}, mode, {stride}, {dilation}, 1, seed) {}
};
struct Conv2d : ConvNd {
    Conv2d(DevicePtr dev, int in_channels, int out_channels, std::vector<int> kernel, std::vector<int> padding,
           PaddingMode mode, std::vector<int> stride, std::vector<int> dilation, uint64_t seed = 0)
        : ConvNd(2, std::move(dev), in_channels, out_channels, std::move(kernel), std::move(padding), mode, std::move(stride),
                 std::move(dilation), 1, seed) {}
};
struct Conv3d : ConvNd {
    Conv3d(DevicePtr dev, int in_channels, int out_channels, std::vector<int> kernel, std::vector<int> padding,
           PaddingMode mode, std::vector<int> stride, std::vector<int> dilation, uint64_t seed = 0)
        : ConvNd(3, std::move(dev), in_channels, out_channels, std::move(kernel), std::move(padding), mode, std::move(stride),
                 std::move(dilation), 1, seed) {}
};
// `nn::GroupedConv1d / 2d / 3d`: named in the reference's module index (src/lib.rs:783-797,
// neuronika-nn/src/lib.rs:369-387) and in BASELINE.json's north_star; the structs themselves are absent from the
// reference snapshot (its grouped convolution lives at node level: `Convolution::convolution_with_groups`,
// node/convolution/mod.rs:125-144,256-294).  Same fields and argument order as ConvNd plus `groups` after `dilation`;
// weight (Cout, Cin / groups, k...), k = sqrt(1 / (Cin / groups * prod(kernel))).
struct GroupedConv1d : ConvNd {
    GroupedConv1d(DevicePtr dev, int in_channels, int out_channels, int kernel, int padding, PaddingMode mode, int stride,
                  int dilation, int groups, uint64_t seed = 0)
        : ConvNd(1, std::move(dev), in_channels, out_channels, {kernel}, {padding}, mode, {stride}, {dilation}, groups, seed) {}
};
struct GroupedConv2d : ConvNd {
    GroupedConv2d(DevicePtr dev, int in_channels, int out_channels, std::vector<int> kernel, std::vector<int> padding,
                  PaddingMode mode, std::vector<int> stride, std::vector<int> dilation, int groups, uint64_t seed = 0)
        : ConvNd(2, std::move(dev), in_channels, out_channels, std::move(kernel), std::move(padding), mode, std::move(stride),
                 std::move(dilation), groups, seed) {}
};
struct GroupedConv3d : ConvNd {
    GroupedConv3d(DevicePtr dev, int in_channels, int out_channels, std::vector<int> kernel, std::vector<int> padding,
                  PaddingMode mode, std::vector<int> stride, std::vector<int> dilation, int groups, uint64_t seed = 0)
        : ConvNd(3, std::move(dev), in_channels, out_channels, std::move(kernel), std::move(padding), mode, std::move(stride),
                 std::move(dilation), groups, seed) {}
};

// `ModelStatus`-style train/eval switch shared with the Dropout nodes (node/dropout/mod.rs:27).
struct Dropout {
    double p;
    Shared<bool> status;
    explicit Dropout(double p) : p(p), status(std::make_shared<bool>(true)) {}
    void train() const { *status = true; }
    void eval() const { *status = false; }
    VarDiff forward(const VarDiff& x) const { return x.dropout(p, status); }
};

// Multi-head attention composed from reference ops (the module does not exist in the reference;
// SURVEY.md 8a note): Q,K,V = x.mm_t(W)+b; per (b,h): P = dropout(softmax(Q.mm_t(K)*dh^-1/2, 1));
// O = P.mm(V); out = cat(O).mm_t(Wo)+bo.
struct MultiheadAttention {
    Linear q, k, v, o;
    int d_model, heads;
    Dropout drop;
    bool fused = true;  // scale + softmax + dropout as one node (false: three reference nodes)
    bool strided_heads = true;  // attention GEMMs read Q/K/V and write O in the projection layout (false: split/merge copies)
    bool fused_core = true;     // scores -> probabilities -> context as one node on the fused attention kernels (dh in {32, 64, 128}, any S)
    // The three projection weights (and biases, and their gradients) are views of ONE (3*d_model, d_model) allocation, rows
    // [Wq; Wk; Wv]: with `packed_qkv` the projections run as one GEMM with N = 3*d_model, their input gradient as one GEMM
    // with K = 3*d_model, the weight gradients as one GEMM with M = 3*d_model, and the fused attention kernels read Q, K, V
    // as column blocks of the packed output (`nk_attention_qkv_*`).  q / k / v stay ordinary `Linear`s over those views
    // (optimizers, serde and the data-parallel exchange see three parameters as before).  false: three Linear nodes.
    bool packed_qkv = true;
    MultiheadAttention(DevicePtr dev, int d_model, int heads, double p, uint64_t seed);
    // four Linear layers built elsewhere (e.g. deserialised): their weights are NOT packed, `packed_qkv` is off
    MultiheadAttention(Linear q, Linear k, Linear v, Linear o, int heads, double p);
    VarDiff forward(const VarDiff& x, int batch) const;  // x: (batch*seq, d_model)

   private:
    Shared<HipArray> wqkv_, bqkv_, gwqkv_, gbqkv_;  // packed storage (null when the layers were handed in)
};

}  // namespace nn

// ---------------------------------------------------------------------------------------------
// neuronika-variable/src/serde.rs:10-58 — (de)serialisation of leaves in ndarray's serde wire format
//   {"v":1,"dim":[d0,d1,...],"data":[row-major f32 values]}
// `Var`/`VarDiff` serialise their data (D2H copy); deserialising creates a leaf (`VarDiff`: a leaf with
// `requires_grad()`).  Modules serialise as objects of their parameters, like `#[derive(Serialize)]` on
// the reference structs (neuronika-nn/src/lib.rs:397-404: {"weight":..., "bias":...}).
// ---------------------------------------------------------------------------------------------
namespace serde {

struct Json {  // minimal JSON document model (object keys keep their order)
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    std::string text;  // number literal or string value
    std::vector<Json> items;
    std::vector<std::pair<std::string, Json>> members;
    const Json& at(const std::string& key) const;
    bool has(const std::string& key) const;
};
Json parse(const std::string& text);

std::string to_json(const Var& v);
std::string to_json(const VarDiff& v);
Var var_from_json(DevicePtr dev, const Json& j);
Var var_from_json(DevicePtr dev, const std::string& text);
VarDiff vardiff_from_json(DevicePtr dev, const Json& j);
VarDiff vardiff_from_json(DevicePtr dev, const std::string& text);
std::string to_json(const nn::Linear& l);
nn::Linear linear_from_json(DevicePtr dev, const Json& j);
nn::Linear linear_from_json(DevicePtr dev, const std::string& text);

}  // namespace serde

// ---------------------------------------------------------------------------------------------
// neuronika-optim: Optimizer / StochasticGD (optimizer.rs:33-95, sgd/mod.rs:186-236)
// ---------------------------------------------------------------------------------------------
namespace optim {

struct Penalty {  // penalty.rs:2-79
    float l1 = 0.f, l2 = 0.f;
};

// `Optimizer<T>` (optimizer.rs:33-95): register / step / zero_grad / get_lr / set_lr.  State
// buffers (momentum, moments) live on the device, zero-initialised like the reference's.
class Optimizer {
   public:
    virtual ~Optimizer() = default;
    void register_param(const VarDiff& p);  // `register`
    virtual void step();
    void zero_grad() const;
    float get_lr() const { return lr_; }
    void set_lr(float lr) { lr_ = lr; }
    const std::vector<VarDiff>& params() const { return params_; }

   protected:
    Optimizer(float lr, Penalty penalty, int nstate) : lr_(lr), penalty_(penalty), nstate_(nstate) {}
    virtual void optimize(const VarDiff& p, std::vector<Shared<HipArray>>& state, int step) = 0;
    float lr_;
    Penalty penalty_;

   protected:
    std::vector<VarDiff> params_;
    std::vector<std::vector<Shared<HipArray>>> state_;
    std::vector<int> steps_;

   private:
    int nstate_;
};

class SGD : public Optimizer {  // sgd/mod.rs
   public:
    SGD(float lr, Penalty penalty = {}, float momentum = 0.f, float dampening = 0.f, bool nesterov = false);
    void step() override;  // every registered parameter of a device in ONE launch (nk_sgd_step_multi)

   private:
    void optimize(const VarDiff& p, std::vector<Shared<HipArray>>& state, int step) override;
    float momentum_, dampening_;
    bool nesterov_;
};

class Adam : public Optimizer {  // adam/mod.rs ; amsgrad/mod.rs when amsgrad = true
   public:
    Adam(float lr, float beta1 = 0.9f, float beta2 = 0.999f, float eps = 1e-8f, Penalty penalty = {}, bool amsgrad = false);

   private:
    void optimize(const VarDiff& p, std::vector<Shared<HipArray>>& state, int step) override;
    float beta1_, beta2_, eps_;
    bool amsgrad_;
};

class Adagrad : public Optimizer {  // adagrad/mod.rs
   public:
    Adagrad(float lr, float lr_decay = 0.f, float eps = 1e-10f, Penalty penalty = {});

   private:
    void optimize(const VarDiff& p, std::vector<Shared<HipArray>>& state, int step) override;
    float lr_decay_, eps_;
};

class RMSProp : public Optimizer {  // rmsprop/mod.rs
   public:
    RMSProp(float lr, float alpha = 0.99f, float eps = 1e-8f, float momentum = 0.f, bool centered = false, Penalty penalty = {});

   private:
    void optimize(const VarDiff& p, std::vector<Shared<HipArray>>& state, int step) override;
    float alpha_, eps_, momentum_;
    bool centered_;
};

// neuronika-optim/src/lr_scheduler/{mod,step_lr,multi_step_lr,exponential_lr,lambda_lr,multiplicative_lr}.rs:
// `step()` = prepare_step (last_lr <- current_lr, epoch += 1; mod.rs:51-59) then the policy, then `optimizer.set_lr`.
namespace lr_scheduler {

class LRScheduler {
   public:
    virtual ~LRScheduler() = default;
    void step();
    float get_last_lr() const { return last_lr_; }
    float get_current_lr() const { return current_lr_; }
    size_t get_current_epoch() const { return epoch_; }
    void set_current_epoch(size_t e) { epoch_ = e; }

   protected:
    explicit LRScheduler(Optimizer& opt) : opt_(opt), initial_lr_(opt.get_lr()), last_lr_(opt.get_lr()), current_lr_(opt.get_lr()) {}
    virtual bool update(float& lr) = 0;  // new lr for epoch_ (already incremented); false = unchanged
    Optimizer& opt_;
    float initial_lr_, last_lr_, current_lr_;
    size_t epoch_ = 0;
};
class StepLR : public LRScheduler {  // every `step_size` epochs: lr *= gamma  (step_lr/mod.rs:59-65)
   public:
    StepLR(Optimizer& opt, size_t step_size, float gamma) : LRScheduler(opt), step_size_(step_size), gamma_(gamma) {}
    void set_gamma(float g) { gamma_ = g; }

   private:
    bool update(float& lr) override;
    size_t step_size_;
    float gamma_;
};
class MultiStepLR : public LRScheduler {  // at each milestone: lr *= gamma  (multi_step_lr/mod.rs:56-67)
   public:
    MultiStepLR(Optimizer& opt, std::vector<size_t> milestones, float gamma) : LRScheduler(opt), milestones_(std::move(milestones)), gamma_(gamma) {}
    void set_milestones(std::vector<size_t> m) { milestones_ = std::move(m); }

   private:
    bool update(float& lr) override;
    std::vector<size_t> milestones_;
    float gamma_;
};
class ExponentialLR : public LRScheduler {  // every epoch: lr *= gamma  (exponential_lr/mod.rs:54-58)
   public:
    ExponentialLR(Optimizer& opt, float gamma) : LRScheduler(opt), gamma_(gamma) {}
    void set_gamma(float g) { gamma_ = g; }

   private:
    bool update(float& lr) override { lr = last_lr_ * gamma_; return true; }
    float gamma_;
};
class LambdaLR : public LRScheduler {  // lr = initial_lr * f(epoch)  (lambda_lr/mod.rs:56-61)
   public:
    LambdaLR(Optimizer& opt, std::function<float(size_t)> f) : LRScheduler(opt), f_(std::move(f)) {}

   private:
    bool update(float& lr) override { lr = initial_lr_ * f_(epoch_); return true; }
    std::function<float(size_t)> f_;
};
class MultiplicativeLR : public LRScheduler {  // lr = last_lr * f(epoch)  (multiplicative_lr/mod.rs:54-59)
   public:
    MultiplicativeLR(Optimizer& opt, std::function<float(size_t)> f) : LRScheduler(opt), f_(std::move(f)) {}

   private:
    bool update(float& lr) override { lr = last_lr_ * f_(epoch_); return true; }
    std::function<float(size_t)> f_;
};

}  // namespace lr_scheduler

}  // namespace optim

// ---------------------------------------------------------------------------------------------
// data-parallel gradient exchange (net-new): RCCL sum all-reduce of the registered parameters'
// gradient buffers on the side stream, bucketed in reverse registration order so a bucket can
// start as soon as backward has produced it.
// ---------------------------------------------------------------------------------------------
namespace dp {

class Communicator {
   public:
    static std::string unique_id();  // 128 raw bytes, create on rank 0
    Communicator(DevicePtr dev, int nranks, int rank, const std::string& id);
    // `nranks` virtual ranks holding this rank's values (nk_comm_init_replicas): sum all-reduce = multiply by nranks
    // (channels, gbps: the paced stand-in of benchmarks/overlap_projection.py; 0, 0: an unthrottled streaming pass)
    static std::shared_ptr<Communicator> replicas(DevicePtr dev, int nranks, int channels = 0, double gbps = 0.0);
    ~Communicator();
    int rank() const { return rank_; }
    int size() const { return size_; }
    nk_comm* raw() const { return h_; }
    DevicePtr device() const { return dev_; }

   private:
    Communicator(DevicePtr dev, int nranks, int channels, double gbps);
    DevicePtr dev_;
    nk_comm* h_ = nullptr;
    int rank_, size_;
};

// Overlapped exchange: pass to `loss.backward(1/world, &sync)`; each registered parameter's
// gradient is all-reduced (sum) on the side stream as soon as the last backward node writing it
// has been issued (reverse layer order), while the remaining backward GEMMs keep the compute
// stream busy.  Gradients below `small_elems` (biases: latency-bound) are collected and sent as
// ONE RCCL group as soon as the last of them is final.  `join()` makes the compute stream wait
// for the exchange (no host sync).  Every rank must run the same tape: the order of the
// collectives is the order in which backward finalises the gradients.
class GradientSync : public BackwardHook {
   public:
    GradientSync(std::shared_ptr<Communicator> comm, const std::vector<VarDiff>& params, size_t small_elems = 65536);
    ~GradientSync() override;
    void grad_ready(const Gradient* g) override;
    bool wants_parts(const Gradient* g) const override;
    void grad_part_ready(const Gradient* g, size_t offset, size_t count) override;
    void join();
    size_t bytes_per_step() const { return bytes_; }
    // run the exchange even with a single rank (RCCL then copies in place): lets one GPU exercise every code path
    void set_force_exchange(bool on) { force_ = on; }
    size_t exchanges_issued() const { return issued_; }  // collective launches so far (a group counts once)
    size_t elements_exchanged() const { return elems_; }
    // Which weight gradients are handed over in row blocks (two launches of the weight-gradient GEMM instead of one):
    //   All       every large gradient (rounds 1 - 2);
    //   LastOnly  only the large gradient that became final LAST in the previous backward pass - the one whose exchange is
    //             exposed behind the pass; the others have the rest of backward to hide behind and keep the single GEMM
    //             launch (a split weight-gradient GEMM pays its prologue / drain twice: profiles/r03_overlap_projection.md);
    //   None      never.
    // Default LastOnly; `set_parts` overrides (measurement aid).  The first pass of LastOnly splits nothing.
    enum class Parts { All, LastOnly, None };
    void set_parts(Parts p) { parts_ = p; }
    // How many of the GPU's resident-block slots the exchange occupies while it runs: RCCL's channel workgroups (one slot
    // each; cap them with NCCL_MAX_NCHANNELS and pass the same number).  From the first large gradient handed over until
    // `join()` the device handle is told (nk_device_set_busy_slots), and the backward GEMMs issued in between plan their last
    // round of tiles around the missing slots (sgemm_tail_kernel) instead of leaving a ragged one - 10 - 15 % per 4096^3 GEMM
    // otherwise (profiles/r04_gemm_under_load.md, profiles/r05_gemm_under_load.md).  0 (default): the GEMMs assume an idle chip.
    void set_busy_slots(int n) { busy_slots_ = n < 0 ? 0 : n; }
    int busy_slots() const { return busy_slots_; }

   private:
    void flush_small();
    std::shared_ptr<Communicator> comm_;
    std::unordered_map<const Gradient*, Shared<Gradient>> params_;
    std::unordered_map<const Gradient*, size_t> parts_done_;  // elements already handed over piecewise this pass
    std::vector<const Gradient*> small_pending_;               // small gradients that are final, not yet sent
    std::vector<nk_event*> events_;
    size_t next_event_ = 0;
    size_t bytes_ = 0;
    size_t issued_ = 0;
    size_t elems_ = 0;
    size_t small_elems_;
    size_t n_small_ = 0;
    bool force_ = false;
    int busy_slots_ = 0;
    bool busy_told_ = false;                      // the device handle currently carries busy_slots_
    Parts parts_ = Parts::LastOnly;
    const Gradient* last_large_ = nullptr;       // large gradient handed over last in the current pass
    const Gradient* split_next_ = nullptr;       // ... in the previous pass: the one LastOnly splits
    bool active() const { return comm_->size() > 1 || force_; }
};

// Non-overlapped form: all-reduce every gradient after backward has been issued.
void all_reduce_gradients(const Communicator& comm, const std::vector<VarDiff>& params);

}  // namespace dp

}  // namespace neuronika
