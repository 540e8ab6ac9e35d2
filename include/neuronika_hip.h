/*
 * neuronika_hip.h — C ABI of the MI355X (gfx950) dense-tensor backend for neuronika's
 * Var/VarDiff op graph.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own:
 * the seam is the pair of crate-private traits
 *     trait Forward  { fn forward(&self);  }   neuronika-variable/src/autograd.rs:7-12
 *     trait Backward { fn backward(&self); }   neuronika-variable/src/autograd.rs:20-25
 * implemented by node structs that own handles to their operand/output buffers.  Each entry
 * point below replaces the BODY of one such `forward()` / `backward()` (cited per function);
 * the reference's own accelerator template (`neuronika-variable/src/cuda/`) shows the shape a
 * backend takes: a `Device` handle (cuda/device.rs:11-58), a device array replacing
 * `ndarray::Array` (cuda/cuarray.rs:10-19) and nodes that call the library in `forward()`
 * (cuda/cunode/binary_op/mod.rs:55-82).  INTEGRATION.md shows the Rust binding.
 *
 * Conventions (identical to the reference's ndarray path):
 *   - every tensor is dense f32, C-contiguous (row-major), described by (pointer, shape[]);
 *   - every *_fwd OVERWRITES its output (GEMM beta = 0); every *_bwd ACCUMULATES (`+=`)
 *     into the operand gradient (GEMM beta = 1);
 *   - the host owns every device buffer; the library never keeps a data pointer past a call;
 *   - calls are asynchronous on the device's compute stream, in call (= tape) order; the host
 *     synchronises only in nk_download / nk_device_sync / nk_event_* queries;
 *   - one host thread per nk_device (the reference graph is Rc<RefCell<..>>, i.e. !Send);
 *     different devices may be driven from different threads / processes concurrently;
 *   - every function returns NK_OK (0) or an error code; nk_last_error() gives the message
 *     of the calling thread's last failure.  The reference convention is panic
 *     (`.unwrap()`, cuda/device.rs:36-45; `assert!`, utils.rs:438-496): a host binding turns
 *     a non-zero status into a panic.  Shape validation that the reference does on the host
 *     (`check_conv_args`, `cobroadcast`) is repeated here and reported as NK_ERR_INVALID.
 *
 * No torch / C++ types cross this boundary: plain pointers, sizes and scalars only.
 */
#ifndef NEURONIKA_HIP_H
#define NEURONIKA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nk_device nk_device; /* cuda/device.rs:11-16  `Device`            */
typedef struct nk_event nk_event;   /* hipEvent on a device stream (timing/ordering) */
typedef struct nk_comm nk_comm;     /* one rank of a RCCL communicator (net-new)  */

enum nk_status {
    NK_OK = 0,
    NK_ERR_INVALID = 1,     /* bad argument / shape (reference: assert!/panic!)      */
    NK_ERR_HIP = 2,         /* HIP runtime failure                                   */
    NK_ERR_RCCL = 3,        /* RCCL failure                                          */
    NK_ERR_OOM = 4,         /* device allocation failed                              */
    NK_ERR_UNSUPPORTED = 5  /* valid in the reference, not implemented by this build */
};

enum nk_binary_op { NK_ADD = 0, NK_SUB = 1, NK_MUL = 2, NK_DIV = 3 };
enum nk_reduction { NK_REDUCTION_SUM = 0, NK_REDUCTION_MEAN = 1 }; /* lib.rs:29-36 */

#define NK_MAX_DIMS 8

/* ------------------------------------------------------------------ lifecycle ---------- */
/* `Device::new(idx)` cuda/device.rs:34-58.  Creates the compute stream and the side
 * (communication) stream of GPU `idx`. */
int nk_device_count(int* out);
int nk_device_create(int idx, nk_device** out);
int nk_device_destroy(nk_device* dev);
int nk_device_sync(nk_device* dev);
int nk_device_index(const nk_device* dev);
void* nk_stream_compute(nk_device* dev); /* hipStream_t */
void* nk_stream_comm(nk_device* dev);    /* hipStream_t */
const char* nk_last_error(void);
const char* nk_version(void);
/* Development overrides of the kernels' launch heuristics, per device handle (the library reads NO environment variable):
 *   NK_TUNE_GEMM_FORCE     values = ti, tj, splits[, tiles per block[, tile-order group height[, look-ahead threshold]]]
 *                          (ti, tj in {1, 2}: 64- or 128-wide tile sides); n = 0 returns to the rules
 *   NK_TUNE_GEMM_KPAIR     values[0] = -1 rule / 0 never / 1 k-pair blocks, lock-step groups / 2 skewed groups
 *   NK_TUNE_ATTENTION_OCC  values[0] = 0 rule / 2: forward register budget sized for two blocks per CU
 *   NK_TUNE_GEMM_PAIR      values[0] = -1 rule / 0 nk_sgemm_pair always launches twice / 1 one launch whenever eligible
 *   NK_TUNE_CONV_NARROW    values[0] = 0 the conv kernel gradient's uniform launch / 1..100 its mixed launch (the last, half-empty
 *                          column tile through 64-wide blocks), a narrow block's k-tile priced at that percentage of a wide one's;
 *                          n = 0: the measured rules (65 % with 128-row tiles, 80 % with 64-row tiles)
 *   NK_TUNE_CONV_WINOGRAD  values[0] = -1 rule / 0 the 3x3 stride-1 forward and input gradient never take the Winograd F(2x2, 3x3)
 *                          kernels (implicit GEMM as in rounds 1 - 4) / 1 whenever the shape allows (also below the block-count rule);
 *                          values[1] (optional) = -1 rule / 0 no staggered start of its persistent blocks / > 0 the stagger unit in
 *                          shader clocks (blocks one tile block short of the longest walk start 1 - 3 units late);
 *                          values[2] (optional) = -1 rule / 0 narrow blocks (two waves, 64 output channels, chunks of 16 reduction
 *                          channels) / 1 wide blocks (four waves, 128 channels, chunks of 32) where the channel counts allow both;
 *                          values[3] (optional) = -1 rule / 0 the kernel gradient never takes its Winograd F(3x3, 2x2) form / 1 whenever
 *                          the shape allows (64 | both channel counts)
 *   NK_TUNE_GEMM_CHAIN     values[0] = -1 rule / 0 an unsplit GEMM sums K as ONE f32 chain whatever its length / L (a multiple of 64):
 *                          unsplit plain-epilogue GEMMs (MatMul, MatMulT, weight gradients) with K > L run as consecutive launches
 *                          over equal pieces of K, each on top of the last (beta = 1): chains of at most L.  Rule: L = 2048 - what
 *                          keeps 4096- and 8192-long contractions inside 1e-6 K |a| |b| of the f64 result (SURVEY.md 8c ii; the
 *                          reference's matrixmultiply sums K in cache blocks too, matrix_matrix_mul/mod.rs:33-39)
 *   NK_TUNE_CONV_S2DX      values[0] = -1 rule / 0 the 3x3 stride-2 input gradient never takes its fused-phase kernel (the four stride
 *                          phases of a tile in one block walk; per-phase implicit GEMMs as in rounds 1 - 5) and the 3x3 stride-2 forward
 *                          never its tap-plane kernel (by rule only with 128 | output channels and eight or more blocks per CU) / 1 whenever the shape
 *                          allows (one group, even input extents, padding 0 or 1 alike on both axes, 64 | input channels, 16 | output
 *                          channels) / 2, 3: the same with narrow (two waves, 64 channels) / wide (four waves, 128 channels) blocks forced
 * For schedule sweeps (benchmarks/ab_*.py) and the tests that pit one schedule against another bit for bit; results never
 * depend on them beyond summation order (split-K, chain length, algorithm). */
enum { NK_TUNE_GEMM_FORCE = 0, NK_TUNE_GEMM_KPAIR = 1, NK_TUNE_ATTENTION_OCC = 2, NK_TUNE_GEMM_PAIR = 3, NK_TUNE_CONV_NARROW = 4,
       NK_TUNE_CONV_WINOGRAD = 5, NK_TUNE_GEMM_CHAIN = 6, NK_TUNE_CONV_S2DX = 7 };
int nk_dev_tune(nk_device* dev, int knob, const int* values, int n);
/* How many convolution launches on this handle took the Winograd F(2x2, 3x3) kernels so far (forward + input gradient; the rule
 * of NK_TUNE_CONV_WINOGRAD decides per launch).  For harnesses that must say which algorithm produced a time: bench.py quotes the
 * C3 roofline on the DIRECT algorithmic flops of node/convolution/mod.rs:85-123,146-189 and states beside it what was executed. */
int nk_conv_winograd_launches(nk_device* dev, uint64_t* count);
/* Tell the device handle that `n` of the GPU's resident-block slots (two 128x128 GEMM blocks per CU) are held by work on another
 * stream until further notice - the channel workgroups of an all-reduce in flight beside the backward pass
 * (vardiff.rs:125-141 -> optimizer.rs:81-86 is where the exchange sits; dp::GradientSync sets it when it hands the first
 * gradient over and clears it in join()).  GEMM launches whose tile count no longer divides the free slots then run whole
 * rounds of one tile per block and cut the left-over tiles along K (sgemm_tail_kernel, nk_gemm.hip): deterministic, bits a
 * function of (shape, n); every tile outside the left-over rectangle is the plain launch's.  n = 0 (the default): the chip is
 * ours, plain launches.  0 <= n <= CUs. */
int nk_device_set_busy_slots(nk_device* dev, int n);

/* ------------------------------------------------------------------ memory ------------- */
/* `CuArray::zeroed` cuda/cuarray.rs:35-42; outputs and gradients are allocated zeroed at
 * graph-build time (var.rs:224,1041; gradient.rs:47-54). */
int nk_alloc_zeroed(nk_device* dev, size_t n_f32, float** out);
int nk_free(nk_device* dev, float* ptr);
/* `CuArray::from_slice / from_ndarray` cuda/cuarray.rs:62-72,114-117 (H2D) */
int nk_upload(nk_device* dev, float* dst, const float* host_src, size_t n);
/* `CuArray::as_ndarray` cuda/cuarray.rs:101-106 (D2H, synchronises the compute stream) */
int nk_download(nk_device* dev, float* host_dst, const float* src, size_t n);
/* root-gradient seeding `grad_mut().fill(seed)` vardiff.rs:133; `zero_grad` vardiff.rs:100-102;
 * `Gradient::with_grad` re-zero gradient.rs:71-78 */
int nk_fill(nk_device* dev, float* ptr, size_t n, float value);
int nk_copy(nk_device* dev, float* dst, const float* src, size_t n);

/* H2D input pipeline = the device half of `for batch in dataset.batch(n)` (neuronika-data/src/lib.rs:81,570):
 * page-locked host staging + a third (copy) stream, so that the upload of batch k+1 overlaps the compute of
 * batch k.  nk_upload_async returns immediately; order it against the compute stream with events recorded /
 * waited on stream 2. */
int nk_host_alloc(size_t bytes, void** out);
int nk_host_free(void* ptr);
int nk_upload_async(nk_device* dev, float* dst, const float* pinned_src, size_t n);

/* ------------------------------------------------------------------ events ------------- */
int nk_event_create(nk_device* dev, nk_event** out);
int nk_event_destroy(nk_event* ev);
int nk_event_record(nk_event* ev, int on_comm_stream); /* 0: compute stream, 1: comm stream, 2: copy stream */
int nk_event_sync(nk_event* ev);
int nk_event_elapsed_ms(nk_event* start, nk_event* stop, float* ms);
int nk_stream_wait_event(nk_device* dev, int on_comm_stream, nk_event* ev);

/* ------------------------------------------------------------------ graph capture ------ */
/* Small graphs (the reference's quickstart MLP: 64x3 inputs) are launch-bound: tens of kernels of a few
 * microseconds each.  Everything a tape step enqueues on the compute stream between nk_graph_begin and nk_graph_end
 * is recorded into a hipGraph instead of executed; nk_graph_launch replays it with one submission.  The captured
 * region must not synchronise with the host (no nk_download / nk_device_sync / item()) nor grow an allocation, and
 * it replays the SAME launches: scalars baked into kernel arguments (the learning rate) stay what they were at capture
 * time.  Calls whose kernel arguments must change from call to call REFUSE to be captured (NK_ERR_INVALID) instead of
 * freezing them: optimizer steps that depend on the step count (nk_adam_step: 1 - beta^step; nk_adagrad_step with
 * lr_decay != 0), and the forwards that draw a dropout mask (nk_dropout_fwd / nk_scale_softmax_dropout_fwd /
 * nk_attention_fwd with train != 0 and 0 < p < 1: the Philox offset - every replay would drop the same elements).
 * SGD / RMSProp steps, evaluation-mode and p = 0 dropout capture fine.  A workspace the
 * device outgrows later stays allocated while any nk_graph of that device exists (captured kernels keep its address);
 * destroy a device's graphs before the device. */
typedef struct nk_graph nk_graph;
int nk_graph_begin(nk_device* dev);
int nk_graph_end(nk_device* dev, nk_graph** out);
int nk_graph_launch(nk_graph* graph);
int nk_graph_destroy(nk_graph* graph);

/* ------------------------------------------------------------------ kernel timing ------ */
/* Bench instrumentation: between nk_profile_begin and nk_profile_end every launch of the
 * MFMA kernels is bracketed by a HIP event pair on the compute stream.  nk_profile_end
 * synchronises and returns, for one kernel class, the number of launches, the sum of their
 * durations and the algorithmic flop they performed (2*M*N*K per GEMM; 2*N*Cout*L*K per conv
 * pass). */
enum nk_kernel_class { NK_KERNEL_SGEMM = 0, NK_KERNEL_CONV = 1, NK_KERNEL_ATTENTION = 2 };
int nk_profile_begin(nk_device* dev);
/* Suspends (paused != 0) / resumes the bracketing inside a window without dropping its records: an event pair costs the stream
 * ~10 us per launch, so a harness may instrument every n-th step of its timed region instead of all of them. */
int nk_profile_pause(nk_device* dev, int paused);
int nk_profile_end(nk_device* dev, int kernel_class, int* launches, double* total_ms, double* total_flop);

/* ------------------------------------------------------------------ GEMM (MFMA) -------- */
/* Row-major C(MxN) = alpha * op(A)(MxK) * op(B)(KxN) + beta * C; op(X) = X or X^T
 * (trans != 0: the stored matrix is the transpose, i.e. A is stored KxM with leading
 * dimension lda).  Replaces `ndarray::linalg::general_mat_mul` at its six call sites:
 * node/matrix_matrix_mul/mod.rs:33,65,97 and node/matrix_matrix_mul_t/mod.rs:33,65,97. */
int nk_sgemm(nk_device* dev, int transA, int transB, int M, int N, int K, float alpha,
             const float* A, int lda, const float* B, int ldb, float beta, float* C, int ldc);
/* Batched over a two-level batch index b = bo * batch_inner + bi; operand offset =
 * bo * stride_outer + bi * stride_inner (elements).  Used by the composed multi-head
 * attention: bo = sample, bi = head, so Q_bh is a strided view of the (B*S) x d projection. */
int nk_sgemm_batched(nk_device* dev, int transA, int transB, int M, int N, int K, float alpha,
                     const float* A, int lda, long long sAo, long long sAi,
                     const float* B, int ldb, long long sBo, long long sBi, float beta,
                     float* C, int ldc, long long sCo, long long sCi,
                     int batch_outer, int batch_inner);

/* Two independent products in ONE launch: C0 = op(A0).op(B0) + beta0*C0 and C1 = op(A1).op(B1) + beta1*C1 (alpha = 1).  When
 * neither product fills the chip by itself (1024^3: 256 blocks, one per CU) the two grids run side by side - every CU gets the
 * second resident block a large launch has, and launch boundary, dispatch ramp and tail are paid once.  Taken by rule for
 * aligned, unsplit (NN | NT | TN) + TN pairs of equal tile shape whose blocks together fit the chip's resident slots; two
 * ordinary launches otherwise (and whenever an output overlaps the other product's output or operands).  Every output is the fma chain nk_sgemm gives it without k-pair blocks (NK_TUNE_GEMM_KPAIR = 0):
 * bit-identical to two calls under that setting. */
int nk_sgemm_pair(nk_device* dev,
                  int transA0, int transB0, int M0, int N0, int K0, const float* A0, int lda0, const float* B0, int ldb0,
                  float beta0, float* C0, int ldc0,
                  int transA1, int transB1, int M1, int N1, int K1, const float* A1, int lda1, const float* B1, int ldb1,
                  float beta1, float* C1, int ldc1);
/* ... over a two-level batch (nk_sgemm_batched's strides, per operand): the dK / dV products of the attention backward (one
 * launch at small batch x heads; C5's 2 x 4096 blocks measure 0.5 - 1 % faster as two launches and stay two) */
int nk_sgemm_pair_batched(nk_device* dev, int batch_outer, int batch_inner,
                          int transA0, int transB0, int M0, int N0, int K0, const float* A0, int lda0, long long sA0o, long long sA0i,
                          const float* B0, int ldb0, long long sB0o, long long sB0i, float beta0, float* C0, int ldc0,
                          long long sC0o, long long sC0i,
                          int transA1, int transB1, int M1, int N1, int K1, const float* A1, int lda1, long long sA1o, long long sA1i,
                          const float* B1, int ldb1, long long sB1o, long long sB1i, float beta1, float* C1, int ldc1,
                          long long sC1o, long long sC1i);

/* Node-level wrappers, one per reference forward()/backward() body. */
/* MatrixMatrixMul::forward  node/matrix_matrix_mul/mod.rs:31-41   C(n,o) = A(n,m).B(m,o) */
int nk_mm_fwd(nk_device* dev, const float* A, const float* B, float* C, int n, int m, int o);
/* MatrixMatrixMulBackwardLeft::backward  :63-73    dA(n,m) += G(n,o).B(m,o)^T */
int nk_mm_bwd_left(nk_device* dev, float* dA, const float* G, const float* B, int n, int m, int o);
/* MatrixMatrixMulBackwardRight::backward :95-105   dB(m,o) += A(n,m)^T.G(n,o) */
int nk_mm_bwd_right(nk_device* dev, float* dB, const float* A, const float* G, int n, int m, int o);
/* MatrixMatrixMulBackward::backward :121-126 (both operands differentiable: `self.left.backward(); self.right.backward()`)
 * as one call = nk_sgemm_pair of the two products above; assign_x != 0: that gradient is freshly zeroed, written unread. */
int nk_mm_bwd(nk_device* dev, float* dA, float* dB, const float* G, const float* A, const float* B, int n, int m, int o,
              int assign_a, int assign_b);
/* MatrixMatrixMulT::forward  node/matrix_matrix_mul_t/mod.rs:31-41  C(n,o) = A(n,m).B(o,m)^T */
int nk_mm_t_fwd(nk_device* dev, const float* A, const float* B, float* C, int n, int m, int o);
/* MatrixMatrixMulTBackwardLeft::backward  :63-73   dA(n,m) += G(n,o).B(o,m) */
int nk_mm_t_bwd_left(nk_device* dev, float* dA, const float* G, const float* B, int n, int m, int o);
/* MatrixMatrixMulTBackwardRight::backward :95-105  dB(o,m) += G(n,o)^T.A(n,m) */
int nk_mm_t_bwd_right(nk_device* dev, float* dB, const float* G, const float* A, int n, int m, int o);
/* MatrixMatrixMulTBackward::backward :121-126, both products as one call (see nk_mm_bwd) */
int nk_mm_t_bwd(nk_device* dev, float* dA, float* dB, const float* G, const float* A, const float* B, int n, int m, int o,
                int assign_a, int assign_b);

/* `Linear::forward` neuronika-nn/src/lib.rs:425-447  Y(n,o) = X(n,m).W(o,m)^T + b(o): the MatrixMatrixMulT node
 * (matrix_matrix_mul_t/mod.rs:31-41) and the broadcast Addition node (addition/mod.rs:39-50) as ONE kernel - the
 * bias is added to the f32 accumulator in the GEMM epilogue, bit-identical to the two-node result.  Its backward is
 * nk_mm_t_bwd_left (dX += G.W), nk_mm_t_bwd_right (dW += G^T.X) and nk_unbroadcast_add (db += column sums of G). */
int nk_linear_fwd(nk_device* dev, const float* X, const float* W, const float* bias, float* Y, int n, int m, int o);
/* `Linear::forward` followed by `ReLU::forward` (node/relu/mod.rs:29-38) as ONE kernel: Y = max(X.W^T + b, 0), the ReLU
 * applied to the f32 value the Linear epilogue would have stored (`o.max(0.)`: a NaN gives 0) - bit-identical to the two
 * launches; the pre-activation is never written.  ReLU's backward needs only `x > 0`, and max(x, 0) > 0 <=> x > 0, so Y
 * itself is the mask (nk_linear_bwd_input_relu, nk_relu_mask_inplace). */
int nk_linear_relu_fwd(nk_device* dev, const float* X, const float* W, const float* bias, float* Y, int n, int m, int o);
/* MatrixMatrixMulTBackwardLeft::backward (matrix_matrix_mul_t/mod.rs:63-73) followed by ReLUBackward::backward
 * (relu/mod.rs:67-79) of the node that produced this Linear's input X(n,m) = max(Z, 0):
 *   dZ(n,m) (+)= mask * (G(n,o).W(o,m)),  mask = ((X > 0.) as usize as f32)   (0 * inf = NaN, as the reference's product)
 * applied when the GEMM tile is stored: the gradient w.r.t. X is never written.  assign != 0: dZ is a freshly zeroed
 * gradient, written without being read. */
int nk_linear_bwd_input_relu(nk_device* dev, float* dZ, const float* G, const float* W, const float* X, int n, int m, int o,
                             int assign);
/* ReLUBackward::backward in place: g = ((y > 0.) as f32) * g  (the fused Linear+ReLU node's fallback when a consumer other
 * than nk_linear_bwd_input_relu wrote into its gradient; idempotent, so contributions that arrived masked stay as they are) */
int nk_relu_mask_inplace(nk_device* dev, float* g, const float* y, size_t n);

/* ------------------------------------------------------------------ convolution -------- */
/* N-d (nd = 1,2,3) cross-correlation without internal padding, NC[D]HW layout.
 *   x: [N, Cin, in...]   w: [Cout, Cin/groups, k...]   y: [N, Cout, out...]
 *   out_i = (in_i - dilation_i*(k_i-1) - 1)/stride_i + 1          utils.rs:207-237
 * x_shape has 2+nd entries, w_shape 2+nd entries.
 * Convolution::forward            node/convolution/mod.rs:331-355 (-> :85-144)  y  = conv(x,w)
 * ConvolutionBackwardInput        :427-449 (-> :146-189, 256-274)               dx += ...
 * ConvolutionBackwardKernel       :488-510 (-> :191-226, 276-294)               dw += ...
 * Algorithms, chosen inside each call by geometry and size (rules in csrc/nk_conv.hip, overridable through nk_dev_tune): Winograd
 * F(2x2, 3x3) (forward, input gradient) and F(3x3, 2x2) (kernel gradient) on the f32 MFMA core for 3 x 3 / stride 1 / dilation 1 / one
 * group with 64 | channel counts and any output extents from 2 x 2 on (odd ones through instantiations with masked border tiles); implicit GEMM on the same core for channel counts that are multiples of 32 and
 * a generic form for the rest; direct kernels for <= 16 channels per group.  Every form sums in a fixed order (run-to-run identical);
 * the forms differ from each other in that order only (equal on integer-valued data, to contraction tolerance otherwise). */
int nk_conv_fwd(nk_device* dev, int nd, const float* x, const int* x_shape, const float* w,
                const int* w_shape, float* y, const int* stride, const int* dilation, int groups);
int nk_conv_bwd_input(nk_device* dev, int nd, float* dx, const int* x_shape, const float* g,
                      const float* w, const int* w_shape, const int* stride, const int* dilation,
                      int groups);
int nk_conv_bwd_kernel(nk_device* dev, int nd, float* dw, const int* w_shape, const float* g,
                       const float* x, const int* x_shape, const int* stride, const int* dilation,
                       int groups);
/* `Conv{1,2,3}d` module forward = convolution node + broadcast Addition of the (Cout,1,..) bias (neuronika-nn/src/
 * lib.rs:630-916; addition/mod.rs:39-50) as ONE kernel: bias[co] is added to the f32 accumulator in the epilogue,
 * bit-identical to the two-node result.  Backward = nk_conv_bwd_input, nk_conv_bwd_kernel, nk_unbroadcast_add. */
int nk_conv_bias_fwd(nk_device* dev, int nd, const float* x, const int* x_shape, const float* w,
                     const int* w_shape, const float* bias, float* y, const int* stride,
                     const int* dilation, int groups);
/* The same module's backward towards its kernel AND its bias in one pass: ConvolutionBackwardKernel (convolution/mod.rs:
 * 191-226) plus AdditionBackwardRight of the (Cout,1,..) bias (addition/mod.rs:109-135), db[co] (+)= sum of g over samples
 * and positions.  The implicit-GEMM pass stages g as its A operand anyway and sums it on the way (no separate 200 MB
 * reduction at C3); geometries that take other kernels run the reduction behind the scenes.  `assign_*` != 0: first write. */
int nk_conv_bwd_kernel_bias(nk_device* dev, int nd, float* dw, float* db, const int* w_shape, const float* g,
                            const float* x, const int* x_shape, const int* stride, const int* dilation, int groups,
                            int assign_dw, int assign_db);
/* The `Conv2d` module (lib.rs:724-812: pad -> convolution -> + bias) with Zero padding FOLDED INTO the forward and the kernel-gradient
 * pass: `x` / `x_shape` are the UNPADDED input, `padding[i]` the symmetric zero padding of spatial axis i; the padded copy (Pad::forward,
 * pad/zero/mod.rs:5-31 - 110 MB and a 40 us kernel at C3) is never made.  Only the Winograd kernels read their operands through
 * out-of-range-is-zero buffer loads, so only their geometries fold: 3 x 3, stride 1, dilation 1, one group, padding 0 or 1 per axis (not
 * all zero), 64 | both channel counts.  nk_conv_padding_folds answers, for a geometry and the rules in force on the
 * handle, whether BOTH passes would run their Winograd kernels anyway (*folds = 1: build the module node without the Pad node and
 * call the two `_padded` entries; 0: pad, then nk_conv_bias_fwd / nk_conv_bwd_kernel_bias).  The `_padded` entries themselves fold for
 * every geometry the kernels can (whatever the block-count rules say); for the others - and when the rules in force at CALL time decline,
 * e.g. a nk_dev_tune change after the graph was built - they are the two nodes they stand for: Pad::forward into a scratch region of the
 * device handle, then the convolution entry on the copy (any nd, stride, dilation, groups; never NK_ERR_UNSUPPORTED).  Same values as the
 * two-node form, bit for bit, either way (zeros are read instead of stored).  bias / db may be NULL.  The input gradient's padded form is
 * below. */
int nk_conv_padding_folds(nk_device* dev, int nd, const int* x_shape, const int* padding, const int* w_shape, const int* stride,
                          const int* dilation, int groups, int* folds);
int nk_conv_bias_fwd_padded(nk_device* dev, int nd, const float* x, const int* x_shape, const int* padding, const float* w,
                            const int* w_shape, const float* bias, float* y, const int* stride, const int* dilation, int groups);
int nk_conv_bwd_kernel_bias_padded(nk_device* dev, int nd, float* dw, float* db, const int* w_shape, const float* g, const float* x,
                                   const int* x_shape, const int* padding, const int* stride, const int* dilation, int groups,
                                   int assign_dw, int assign_db);
/* `Conv{1,2,3}d` module backward towards its input when the module's padding mode is Zero (lib.rs:630-916: pad ->
 * convolution): ConvolutionBackwardInput (convolution/mod.rs:146-189) followed by PadBackward (pad/mod.rs:131-181, the
 * centre block of the padded gradient is accumulated into dx) as ONE kernel.  x_shape is the UNPADDED input
 * [N, Cin, in...], `padding[i]` the symmetric zero padding of spatial axis i the forward convolution saw; only the
 * columns dx needs are computed and the padded gradient is never stored.  Same values as the two-node form.  `_assign`:
 * see the first-write variants below. */
int nk_conv_bwd_input_padded(nk_device* dev, int nd, float* dx, const int* x_shape, const int* padding,
                             const float* g, const float* w, const int* w_shape, const int* stride,
                             const int* dilation, int groups);
int nk_conv_bwd_input_padded_assign(nk_device* dev, int nd, float* dx, const int* x_shape,
                                    const int* padding, const float* g, const float* w,
                                    const int* w_shape, const int* stride, const int* dilation,
                                    int groups);
/* Pad<Constant|Zero>::forward  node/pad/mod.rs:97-129 + pad/constant/mod.rs:14-39;
 * symmetric `padding[i]` on both sides of spatial axis i.  x_shape = [N, C, in...]. */
int nk_pad_const_fwd(nk_device* dev, int nd, const float* x, const int* x_shape, float* y,
                     const int* padding, float value);
/* Pad<Reflective>::forward  pad/reflective/mod.rs:9-136 (border i<pad reads index pad-i, i>=len+pad reads
 * 2(len-1)-(i-pad); requires padding[i] < in[i]);  Pad<Replicative>::forward  pad/replicative/mod.rs:9-134
 * (borders repeat the edge element).  Same shapes as nk_pad_const_fwd. */
int nk_pad_reflective_fwd(nk_device* dev, int nd, const float* x, const int* x_shape, float* y,
                          const int* padding);
int nk_pad_replicative_fwd(nk_device* dev, int nd, const float* x, const int* x_shape, float* y,
                           const int* padding);
/* PadBackward::backward  node/pad/mod.rs:157-181   dx += centre(g)  (every padding mode: the reference
 * does not fold the border gradients back for Reflective/Replicative, neither do we) */
int nk_pad_bwd(nk_device* dev, int nd, float* dx, const int* x_shape, const float* g,
               const int* padding);

/* ------------------------------------------------------------------ broadcast binaries - */
/* Addition|Subtraction|Multiplication|Division::forward node/<op>/mod.rs:39-50.
 * out_shape must equal cobroadcast(l_shape, r_shape) (utils.rs:97-125). */
int nk_binary_fwd(nk_device* dev, int op, float* out, const int* out_shape, int out_nd,
                  const float* l, const int* l_shape, int l_nd,
                  const float* r, const int* r_shape, int r_nd);
/* <Op>BackwardLeft::backward : d_left += unbroadcast(local),
 *   add/sub: local = g; mul: g*r; div: g/r            (addition/mod.rs:86-91, subtraction
 *   :87-92, multiplication :91-103, division :90-99).  `l` may be NULL for add/sub/mul/div
 *   (unused); `r` is needed for mul/div. */
int nk_binary_bwd_left(nk_device* dev, int op, float* d_left, const int* l_shape, int l_nd,
                       const float* g, const int* g_shape, int g_nd,
                       const float* r, const int* r_shape, int r_nd);
/* <Op>BackwardRight::backward : d_right += unbroadcast(local),
 *   add: g; sub: -g; mul: g*l; div: -g*l/r^2          (addition/mod.rs:129-134, subtraction
 *   :130-136, multiplication :138-149, division :139-149). */
int nk_binary_bwd_right(nk_device* dev, int op, float* d_right, const int* r_shape, int r_nd,
                        const float* g, const int* g_shape, int g_nd,
                        const float* l, const int* l_shape, int l_nd, const float* r);
/* `utils::accumulate` (utils.rs:152-192) with the INTENDED semantics: dst += src summed
 * over every axis dst lacks or has with extent 1 (the reference's lane-axis choice is
 * defective for non-square shapes; not replicated). */
int nk_unbroadcast_add(nk_device* dev, float* dst, const int* dst_shape, int dst_nd,
                       const float* src, const int* src_shape, int src_nd);
/* ReLU::forward node/relu/mod.rs:29-38; ReLUBackward::backward :67-79 (dx += (x>0)*g) */
int nk_relu_fwd(nk_device* dev, const float* x, float* y, size_t n);
int nk_relu_bwd(nk_device* dev, float* dx, const float* g, const float* x, size_t n);

/* Pointwise unary nodes ("next" row f-2): forward y = f(x) overwrites; backward dx += f'(.)*g with
 * `ref` = the buffer the reference node keeps (its INPUT for ln, softplus, leaky_relu, pow; its
 * OUTPUT for exp, sqrt, sigmoid, tanh; unused for neg).  node/<op>/mod.rs:35 and :73-86.
 *   NEG   y = -x                       dx -= g                       negation
 *   EXP   y = exp(x)                   dx += g*y                     exp
 *   LN    y = ln(x)                    dx += g/x                     logn
 *   SQRT  y = sqrt(x)                  dx += g/(y*2)                 sqrt
 *   SIGMOID y = 1/(1+exp(-x))          dx += g*y*(1-y)               sigmoid
 *   TANH  y = tanh(x)                  dx += g*(1-y^2)               tanh
 *   SOFTPLUS y = ln(1+exp(x))          dx += g/(1+exp(-x))           softplus
 *   LEAKY_RELU y = x>0 ? x : 0.01x     dx += (x>0)*g + (x<=0)*0.01   leaky_relu (sic: the reference
 *                                       adds 0.01, not 0.01*g — leaky_relu/mod.rs:77-80; replicated)
 *   POW   y = x^e (integer e = iparam) dx += g * x^(e-1) * e         power */
enum nk_unary_op { NK_NEG = 0, NK_EXP = 1, NK_LN = 2, NK_SQRT = 3, NK_SIGMOID = 4, NK_TANH = 5,
                   NK_SOFTPLUS = 6, NK_LEAKY_RELU = 7, NK_POW = 8 };
int nk_unary_fwd(nk_device* dev, int op, const float* x, float* y, size_t n, int iparam);
int nk_unary_bwd(nk_device* dev, int op, float* dx, const float* g, const float* ref, size_t n, int iparam);

/* ------------------------------------------------------------------ reductions --------- */
/* Sum::forward node/sum/mod.rs:28-35 ; SumBackward :60-67 (dx += g, g a device scalar) */
int nk_sum_fwd(nk_device* dev, const float* x, size_t n, float* out);
int nk_sum_bwd(nk_device* dev, float* dx, size_t n, const float* g);
/* Mean::forward node/mean/mod.rs:28-35 ; MeanBackward :60-72 (dx += g/len) */
int nk_mean_fwd(nk_device* dev, const float* x, size_t n, float* out);
int nk_mean_bwd(nk_device* dev, float* dx, size_t n, const float* g);
/* SquaredError::forward node/squared_error/mod.rs:42-59 ; backward :94-123 */
int nk_mse_fwd(nk_device* dev, const float* x, const float* target, size_t n, int reduction,
               float* out);
int nk_mse_bwd(nk_device* dev, float* dx, const float* g, const float* x, const float* target,
               size_t n, int reduction);

/* ------------------------------------------------------------------ first-write variants */
/* The reference allocates every gradient zeroed (gradient.rs:47-54) and every backward node `+=`s into it.
 * For the FIRST node writing into a gradient of the current pass, `0 + v` needs neither the memset nor the
 * read of the destination: the `_assign` variants compute exactly what their `+=` twin computes on an all-zero
 * destination, writing without reading (the GEMM-shaped nodes get the same through nk_sgemm's beta = 0).  Only nodes
 * whose backward covers the WHOLE destination have a twin (not Chunk's tile update or NLL's scatter).
 * The tape (host `Gradient`) keeps the zero fill pending and hands it to the first writer. */
int nk_conv_bwd_input_assign(nk_device* dev, int nd, float* dx, const int* x_shape, const float* g,
                             const float* w, const int* w_shape, const int* stride,
                             const int* dilation, int groups);
int nk_conv_bwd_kernel_assign(nk_device* dev, int nd, float* dw, const int* w_shape, const float* g,
                              const float* x, const int* x_shape, const int* stride,
                              const int* dilation, int groups);
int nk_binary_bwd_left_assign(nk_device* dev, int op, float* d_left, const int* l_shape, int l_nd,
                              const float* g, const int* g_shape, int g_nd, const float* r,
                              const int* r_shape, int r_nd);
int nk_binary_bwd_right_assign(nk_device* dev, int op, float* d_right, const int* r_shape, int r_nd,
                               const float* g, const int* g_shape, int g_nd, const float* l,
                               const int* l_shape, int l_nd, const float* r);
int nk_unbroadcast_assign(nk_device* dev, float* dst, const int* dst_shape, int dst_nd,
                          const float* src, const int* src_shape, int src_nd);
int nk_unary_bwd_assign(nk_device* dev, int op, float* dx, const float* g, const float* ref, size_t n,
                        int iparam);
int nk_softmax_bwd_assign(nk_device* dev, float* dx, const float* g, const float* y, const int* shape,
                          int nd, int axis);
int nk_log_softmax_bwd_assign(nk_device* dev, float* dx, const float* g, const float* y,
                              const int* shape, int nd, int axis);
int nk_dropout_bwd_assign(nk_device* dev, float* dx, const float* g, const float* noise, size_t n,
                          double p, int train);
int nk_concat_bwd_part_assign(nk_device* dev, float* d_operand, const float* g, const int* g_shape,
                              int nd, int axis, int offset, int op_len);
int nk_transpose_bwd_assign(nk_device* dev, float* dx, const float* g, const int* x_shape, int nd);
int nk_sum_bwd_assign(nk_device* dev, float* dx, size_t n, const float* g);
int nk_mean_bwd_assign(nk_device* dev, float* dx, size_t n, const float* g);
int nk_relu_bwd_assign(nk_device* dev, float* dx, const float* g, const float* x, size_t n);
int nk_mse_bwd_assign(nk_device* dev, float* dx, const float* g, const float* x, const float* target,
                      size_t n, int reduction);
int nk_pad_bwd_assign(nk_device* dev, int nd, float* dx, const int* x_shape, const float* g,
                      const int* padding);
int nk_split_heads_bwd_assign(nk_device* dev, float* dx, const float* g, int B, int S, int H, int dh);
int nk_merge_heads_bwd_assign(nk_device* dev, float* dx, const float* g, int B, int S, int H, int dh);
int nk_scale_softmax_dropout_bwd_assign(nk_device* dev, float* d_scores, const float* g_out,
                                        const float* probs, const float* noise, long long rows, int L,
                                        float scale, double p, int train, uint64_t seed, uint64_t offset);

/* ------------------------------------------------------------------ loss criteria ------- */
/* Element-pair criteria reducing to a scalar; x and target share `shape`.
 *   NK_LOSS_MAE             AbsoluteError        node/absolute_error/mod.rs:42-58, :93-123
 *   NK_LOSS_BCE             BinaryCrossEntropy   node/bce/mod.rs:42-62, :97-127   (ln clamped at -100, bwd denominator >= EPSILON)
 *   NK_LOSS_BCE_WITH_LOGITS BCEWithLogits        node/bce_with_logits/mod.rs:42-66, :101-131
 *   NK_LOSS_KLDIV           KLDiv                node/kldiv/mod.rs:42-59, :92-113  (x = log-probabilities; terms with
 *                           target <= 0 contribute 0 - the masked form the reference's vectors require; Mean divides
 *                           by shape[0], every other criterion by the element count)
 * fwd writes out[0]; bwd: dx += d(loss)/dx * g[0]  (g = device scalar). */
enum nk_loss { NK_LOSS_MAE = 0, NK_LOSS_BCE = 1, NK_LOSS_BCE_WITH_LOGITS = 2, NK_LOSS_KLDIV = 3 };
int nk_loss_fwd(nk_device* dev, int loss, const float* x, const float* target, const int* shape, int nd,
                int reduction, float* out);
int nk_loss_bwd(nk_device* dev, int loss, float* dx, const float* g, const float* x, const float* target,
                const int* shape, int nd, int reduction);
/* NegativeLogLikelihood node/nll/mod.rs:43-69, :104-137 on the documented layout (var.rs:645-661):
 * x (minibatch, C, d1..dk) log-probabilities, target (minibatch, d1..dk) class indices stored as f32 and read
 * with Rust's saturating `as usize` (NaN / negative -> 0, fraction dropped; index >= C selects nothing).
 * fwd: out = -sum x[n, target[n,r], r]  (Mean: / shape[0], :64);  bwd: dx[n, target, r] -= g (Mean: / target.len(), :114) */
int nk_nll_fwd(nk_device* dev, const float* x, const float* target, const int* shape, int nd,
               int reduction, float* out);
int nk_nll_bwd(nk_device* dev, float* dx, const float* g, const float* target, const int* shape, int nd,
               int reduction);

/* ------------------------------------------------------------------ GEMV / dot ---------- */
/* MatrixVectorMul node/matrix_vector_mul/mod.rs:31-41  y(n) = A(n,m).x(m);  BackwardLeft :63-69  dA += g (x) x;
 * BackwardRight :92-102  dx += A^T.g */
int nk_mv_fwd(nk_device* dev, const float* A, const float* x, float* y, int n, int m);
int nk_mv_bwd_left(nk_device* dev, float* dA, const float* g, const float* x, int n, int m);
int nk_mv_bwd_right(nk_device* dev, float* dx, const float* A, const float* g, int n, int m);
/* VectorMatrixMul node/vector_matrix_mul/mod.rs:31-41  y(o) = v(m).B(m,o);  BackwardLeft :63-73  dv += B.g;
 * BackwardRight :95-101  dB += v (x) g */
int nk_vm_fwd(nk_device* dev, const float* v, const float* B, float* y, int m, int o);
int nk_vm_bwd_left(nk_device* dev, float* dv, const float* B, const float* g, int m, int o);
int nk_vm_bwd_right(nk_device* dev, float* dB, const float* v, const float* g, int m, int o);
/* VectorVectorMul node/vector_vector_mul/mod.rs:31-34  out = l.r;  backward :57-63  d_operand += other * g[0] */
int nk_vv_fwd(nk_device* dev, const float* l, const float* r, size_t n, float* out);
int nk_vv_bwd(nk_device* dev, float* d_operand, const float* other, const float* g, size_t n);

/* ------------------------------------------------------------------ softmax ------------ */
/* Softmax::forward node/softmax/mod.rs:37-53 ; SoftmaxBackward :84-104
 * LogSoftmax::forward node/logsoftmax/mod.rs:37-53 ; LogSoftmaxBackward :84-102
 * `axis` is any axis of `shape`. */
int nk_softmax_fwd(nk_device* dev, const float* x, float* y, const int* shape, int nd, int axis);
int nk_softmax_bwd(nk_device* dev, float* dx, const float* g, const float* y, const int* shape,
                   int nd, int axis);
int nk_log_softmax_fwd(nk_device* dev, const float* x, float* y, const int* shape, int nd, int axis);
int nk_log_softmax_bwd(nk_device* dev, float* dx, const float* g, const float* y,
                       const int* shape, int nd, int axis);

/* Fused attention probabilities (the Multiplication-by-scalar, Softmax(last axis) and Dropout
 * nodes of the composed multi-head attention in ONE pass over the rows x L score tensor; the
 * module does not exist in the reference — SURVEY.md 8a — its oracle is the composition of
 * node/multiplication, node/softmax and node/dropout, and this produces the same values):
 *   probs = softmax(scores * scale) ; out = dropout(probs)   (mask = Philox(seed, offset), the
 *   same stream nk_dropout_fwd draws; `noise` may be NULL: the mask is then regenerated in the
 *   backward pass instead of being stored).
 * backward: g_p = g_out * mask (no 1/(1-p): reference quirk) ; d_scaled = probs*(g_p - sum(g_p*probs)) ;
 *   d_scores += d_scaled * scale. */
int nk_scale_softmax_dropout_fwd(nk_device* dev, const float* scores, float* probs, float* out, float* noise,
                                 long long rows, int L, float scale, double p, int train, uint64_t seed,
                                 uint64_t offset);
int nk_scale_softmax_dropout_bwd(nk_device* dev, float* d_scores, const float* g_out, const float* probs,
                                 const float* noise, long long rows, int L, float scale, double p, int train,
                                 uint64_t seed, uint64_t offset);
/* Same backward with the probabilities RECOMPUTED from the scores (the forward kernel's exact operation sequence, so
 * bit-identical values): pass `probs = NULL` to nk_scale_softmax_dropout_fwd and the 4-byte/element store plus the
 * buffer disappear.  `assign` != 0: first-write form (see the _assign variants). */
int nk_scale_softmax_dropout_bwd_from_scores(nk_device* dev, float* d_scores, const float* g_out,
                                             const float* scores, const float* noise, long long rows,
                                             int L, float scale, double p, int train, uint64_t seed,
                                             uint64_t offset, int assign);

/* ------------------------------------------------------------------ fused attention core ---
 * The composed multi-head attention's per-(sample, head) chain in one kernel per direction (SURVEY.md 8a note; the
 * composition is MatrixMatrixMulT node/matrix_matrix_mul_t/mod.rs:31-41, Multiplication node/multiplication/mod.rs:39-50,
 * Softmax node/softmax/mod.rs:37-53, Dropout node/dropout/mod.rs:53-79, MatrixMatrixMul node/matrix_matrix_mul/mod.rs:31-41):
 *   S_bh = Q_bh.K_bh^T ; P = softmax(S*scale, axis 1) ; Pd = dropout(P) ; O_bh = Pd.V_bh
 * Q, K, V, O, dO, dQ are the (B*S) x (H*dh) projection layout (head h = columns h*dh .. h*dh+dh-1, sample b = rows
 * b*S ..); scores / dS / dropped are (B*H, SP, SP) with SP = S rounded up to a multiple of 32 (row stride SP; the entries of
 * rows / columns >= S are scratch: padded keys hold a score of -inf and dS = Pd = 0); stats is (B*H, SP, 2) =
 * (m2, 1 / sum_k exp2(S*c1 - m2)) per row with
 * c1 = scale*log2(e) and the shift m2 in [max_k S*c1 - 6, max_k S*c1]: P = exp2(S*c1 - m2) * stats[..,1].  scale > 0.
 * The score tile stays on chip between the two products (online softmax forward, recomputed probabilities backward).
 * Dropout mask: score (bh, r, k) takes draw (bh*SP + r)*SP + k of the layout documented at nk_dropout_fwd - for S % 32 == 0 the
 * Philox stream of nk_scale_softmax_dropout_fwd on the (B*H, S, S) tensor (same seed / offset -> same mask); one forward
 * consumes ceil(B*H*SP*SP / 8) calls.
 * nk_attention_supported: dh in {32, 64, 128}, S >= 1, not (train and p == 1); callers fall back to the node-by-node path. */
int nk_attention_supported(int S, int dh, double p, int train);
/* forward: writes the raw scores (for the backward pass), the row statistics, the dropout draws (1 bit per score:
 * B*H*SP*SP/32 words laid out [b*H + h][SP/32 query tiles][SP/32 key tiles][32 queries of the tile], bit 16 j + e of a word =
 * key 32 kt + 16 j + e kept - opaque to callers, who only hand the buffer from the forward to the backward; may be NULL
 * when dropout is inactive) and O.  `scores` = `stats` = NULL: inference, nothing is
 * kept for a backward pass (O only: no (B*H, SP, SP) tensor exists at all). */
int nk_attention_fwd(nk_device* dev, const float* Q, const float* K, const float* V, float* scores, float* stats,
                     uint32_t* mask_bits, float* O, int B, int S, int H, int dh, float scale, double p, int train,
                     uint64_t seed, uint64_t offset);
/* backward: dQ_bh (+)= dS_bh.K_bh, dK_bh (+)= dS_bh^T.Q_bh, dV_bh (+)= Pd_bh^T.dO_bh (`assign_*` != 0: first write).  dS and
 * Pd ((B*H, SP, SP) each) are scratch the caller owns; they are WRITTEN by the fused kernel and read by the two batched
 * products this call issues after it.  The mask is the forward's (`mask_bits`), as the reference's backward node reads the
 * forward's noise buffer.  DropoutBackward multiplies by the 0/1 mask only (node/dropout/mod.rs:113-128), SoftmaxBackward
 * node/softmax/mod.rs:84-104, MatrixMatrixMul(T)Backward node/matrix_matrix_mul{,_t}/mod.rs:63-105. */
int nk_attention_bwd(nk_device* dev, float* dQ, float* dK, float* dV, float* dS, float* dropped, const float* dO,
                     const float* O, const float* scores, const float* stats, const uint32_t* mask_bits, const float* Q,
                     const float* K, const float* V, int B, int S, int H, int dh, float scale, double p, int train,
                     int assign_dq, int assign_dk, int assign_dv);
/* The same two entry points for Q, K, V (and dQ, dK, dV) that are the three column blocks of ONE (B*S, 3*H*dh) matrix - the
 * output of a single Linear over the row-stacked projection weights [Wq; Wk; Wv] (`nn::MultiheadAttention`'s packed
 * projections: one GEMM with N = 3*H*dh forward, one with K = 3*H*dh for the input gradient, instead of three each).  Row
 * stride 3*H*dh, column offsets 0, H*dh, 2*H*dh; everything else as above.  `assign`: dQKV is a fresh gradient. */
int nk_attention_qkv_fwd(nk_device* dev, const float* QKV, float* scores, float* stats, uint32_t* mask_bits, float* O, int B, int S,
                         int H, int dh, float scale, double p, int train, uint64_t seed, uint64_t offset);
int nk_attention_qkv_bwd(nk_device* dev, float* dQKV, float* dS, float* dropped, const float* dO, const float* O, const float* scores,
                         const float* stats, const uint32_t* mask_bits, const float* QKV, int B, int S, int H, int dh, float scale,
                         double p, int train, int assign);
/* ------------------------------------------------------------------ dropout ------------ */
/* Dropout::forward node/dropout/mod.rs:53-79.  train && 0<p<1: noise ~ Bernoulli(1-p) in
 * {0,1} is (re)drawn from Philox4x32-10(seed, offset) and written to `noise` (f32, like the
 * reference's shared noise array); y = x*noise/(1-p).  Draw layout (shared by every masked entry point): call
 * `offset + i/8` serves elements 8(i/8) .. +7; element i takes word (i%8)/2 of it, rotated by 16 bits for odd i, and
 * is kept iff that 32-bit value < floor((1-p) * 2^32) - rand 0.8's Bernoulli construction on 32 bits.  One forward
 * over n elements consumes ceil(n/8) calls: the host advances `offset` by that much per forward.  !train or p==0: y = x.  p==1: y = 0
 * and `noise` is left untouched.  p outside [0,1] -> NK_ERR_INVALID (reference panics,
 * dropout/mod.rs:38-40). */
int nk_dropout_fwd(nk_device* dev, const float* x, float* y, float* noise, size_t n, double p,
                   int train, uint64_t seed, uint64_t offset);
/* DropoutBackward::backward :113-128.  !train or p==0: dx += g; else dx += g*noise
 * (NOT divided by 1-p: reference behaviour, kept). */
int nk_dropout_bwd(nk_device* dev, float* dx, const float* g, const float* noise, size_t n,
                   double p, int train);

/* ------------------------------------------------------------------ layout glue -------- */
/* Chunk::forward node/chunk/mod.rs:48-64 ; ChunkBackward :99-113.  `chunk_no` indexes
 * ndarray's exact_chunks(chunk_shape) iteration order (row-major over the chunk grid). */
int nk_chunk_fwd(nk_device* dev, const float* x, const int* x_shape, float* y,
                 const int* chunk_shape, int nd, int chunk_no);
int nk_chunk_bwd(nk_device* dev, float* dx, const int* x_shape, const float* g,
                 const int* chunk_shape, int nd, int chunk_no);
/* MultiConcatenate::forward node/multi_concatenate/mod.rs:37-50 ; backward :81-97.
 * One call per operand: copies/accumulates the slice [offset, offset+op_len) of `axis`. */
int nk_concat_fwd_part(nk_device* dev, const float* operand, float* out, const int* out_shape,
                       int nd, int axis, int offset, int op_len);
int nk_concat_bwd_part(nk_device* dev, float* d_operand, const float* g, const int* g_shape,
                       int nd, int axis, int offset, int op_len);
/* Transpose::forward node/transpose/mod.rs:28-37 (reversed axes) ; backward :62-69 */
int nk_transpose_fwd(nk_device* dev, const float* x, float* y, const int* x_shape, int nd);
int nk_transpose_bwd(nk_device* dev, float* dx, const float* g, const int* x_shape, int nd);
/* Head split / merge for the composed attention = Chunk((S,dh)) for every (b,h) tile followed
 * by the per-tile consumers, collapsed into one strided copy:  x[(B*S), H*dh] <-> y[B*H, S, dh].
 * fwd overwrites, bwd accumulates — the same data movement as B*H Chunk nodes
 * (var.rs:401-417) resp. cat(axis 1) per sample then cat(axis 0) (var.rs:564-584). */
int nk_split_heads_fwd(nk_device* dev, const float* x, float* y, int B, int S, int H, int dh);
int nk_split_heads_bwd(nk_device* dev, float* dx, const float* g, int B, int S, int H, int dh);
int nk_merge_heads_fwd(nk_device* dev, const float* x, float* y, int B, int S, int H, int dh);
int nk_merge_heads_bwd(nk_device* dev, float* dx, const float* g, int B, int S, int H, int dh);

/* ------------------------------------------------------------------ optimizer (next row)  */
/* Every optimizer first adds the penalty to the gradient IN PLACE, as the reference does
 * (`grad += penalize(w)`; Penalty penalty.rs:63-79: L1 -> l1*signum(w) with Rust's signum
 * (+-0 -> +-1), L2 -> 2*l2*w, ElasticNet -> both; l1 = l2 = 0: no penalty), then updates w.
 * Optimizer state buffers are owned by the host and start zeroed.
 *
 * SGDParam::optimize  neuronika-optim/src/sgd/mod.rs:186-236: velocity == NULL (momentum <=
 * f32::EPSILON): w -= grad*lr.  Otherwise buffer = buffer*momentum + grad*(1-dampening);
 * nesterov: w -= (grad + buffer*momentum)*lr, else w -= buffer*lr. */
int nk_sgd_step(nk_device* dev, float* w, float* grad, float* velocity, size_t n, float lr,
                float momentum, float dampening, int nesterov, float l1, float l2);
/* The same update for `count` parameters in ONE launch (`Optimizer::step`, optimizer.rs:81-86, walks the registered
 * parameters; their updates are independent): w[i], grad[i], n[i] as above, velocity == NULL or velocity[i] == NULL without
 * momentum.  Element for element the arithmetic of nk_sgd_step. */
int nk_sgd_step_multi(nk_device* dev, int count, float* const* w, float* const* grad, float* const* velocity, const size_t* n,
                      float lr, float momentum, float dampening, int nesterov, float l1, float l2);
/* AdamParam::optimize adam/mod.rs:131-169 ; AMSGradParam::optimize amsgrad/mod.rs:163-205 when
 * max_exp_avg_sq != NULL.  `step` is the 1-based step count (bias corrections 1 - beta^step). */
int nk_adam_step(nk_device* dev, float* w, float* grad, float* exp_avg, float* exp_avg_sq,
                 float* max_exp_avg_sq, size_t n, float lr, float beta1, float beta2, float eps, int step,
                 float l1, float l2);
/* AdagradParam::optimize adagrad/mod.rs:113-140: clr = lr / (1 + (step-1)*lr_decay). */
int nk_adagrad_step(nk_device* dev, float* w, float* grad, float* grad_sq, size_t n, float lr,
                    float lr_decay, float eps, int step, float l1, float l2);
/* RMSPropParam::optimize rmsprop/mod.rs:193-296: grad_avg != NULL selects the centered variant,
 * buffer != NULL the momentum variant (all four combinations). */
int nk_rmsprop_step(nk_device* dev, float* w, float* grad, float* square_avg, float* grad_avg,
                    float* buffer, size_t n, float lr, float alpha, float eps, float momentum, float l1,
                    float l2);

/* ------------------------------------------------------------------ data parallel ------ */
/* Net-new (the reference has no communication backend).  One nk_comm per process/GPU; the
 * 128-byte unique id is created on rank 0 and distributed by the host (any side channel). */
#define NK_COMM_ID_BYTES 128
int nk_comm_unique_id(char id[NK_COMM_ID_BYTES]);
int nk_comm_init_rank(nk_device* dev, int nranks, int rank, const char id[NK_COMM_ID_BYTES],
                      nk_comm** out);
/* Single process, one host thread per GPU: the communicators of `ndev` device handles of THIS process at
 * once (ncclCommInitAll); out[i] belongs to devs[i] and is rank i of ndev.  Each thread then drives its own communicator
 * with the calls below; no unique id has to travel. */
int nk_comm_init_all(int ndev, nk_device* const* devs, nk_comm** out);
/* A communicator of `nranks` virtual ranks that all hold THIS rank's values (no RCCL, no peers):
 * its sum all-reduce multiplies the buffer by nranks on the side stream, with the same stream
 * ordering as the real one.  Lets a single GPU check that an exchange schedule covers every
 * element of every gradient exactly once (a sum over ONE real rank is the identity and would
 * hide a wrong offset or count) and price the schedule without fabric traffic. */
int nk_comm_init_replicas(nk_device* dev, int nranks, int channels, double gbps, nk_comm** out);
int nk_comm_destroy(nk_comm* comm);
/* In-place sum all-reduce of buf[0..n) on the device's SIDE stream.  The side stream first
 * waits for `after` (an event recorded on the compute stream once the bucket's gradients are
 * final; NULL: waits for everything enqueued on the compute stream so far). */
int nk_allreduce_sum_async(nk_comm* comm, float* buf, size_t n, nk_event* after);
/* The same for a list of buffers as ONE RCCL group (one fused launch): for the small,
 * latency-bound gradients (biases) of a step. */
int nk_allreduce_sum_group_async(nk_comm* comm, float* const* bufs, const size_t* counts, int nbufs,
                                 nk_event* after);
/* Make the compute stream wait for all all-reduces issued so far (no host sync). */
int nk_comm_join(nk_comm* comm);
int nk_comm_rank(const nk_comm* comm);
int nk_comm_size(const nk_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* NEURONIKA_HIP_H */
