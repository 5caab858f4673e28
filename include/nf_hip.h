/*
 * nf_hip.h -- C ABI of libnf_hip.so: coupling-flow log_prob hot path on MI355X (gfx950).
 *
 * The reference (kaijennissen/Normalizing_Flows) is pure Python and has no FFI
 * for this path; the seam it does have is the Python-level distrax protocol
 * (SURVEY.md section 8b).  Every entry point below names the reference
 * function(s) it replaces as file:line relative to the reference root.  The
 * Python side that binds these with ctypes is normalizing_flows_amd/_lib.py;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 (NF_OK) or a negative nf_status; nothing throws
 *     across the ABI; nf_last_error() returns a thread-local message.
 *   - the caller owns all host buffers; the library owns device buffers behind
 *     opaque handles.  A handle has one HIP stream; a handle is not thread-safe,
 *     distinct handles may be used from distinct threads.
 *   - sample matrices are row-major float32 [rows, event_dim]; per-sample
 *     vectors are float32 [rows].
 *   - `mem` says where x / y / out pointers live: NF_MEM_HOST (the library
 *     stages through its own device buffers) or NF_MEM_DEVICE (device pointers,
 *     e.g. from nf_malloc; no copies, work is only enqueued on the handle's
 *     stream and the caller synchronises with nf_flow_synchronize).
 */
#ifndef NF_HIP_H
#define NF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NF_ABI_VERSION 1
#define NF_MAX_HIDDEN 8

typedef enum nf_status {
  NF_OK = 0,
  NF_ERR_INVALID = -1,     /* bad argument / shape (the reference raises ValueError) */
  NF_ERR_HIP = -2,         /* a HIP runtime call failed */
  NF_ERR_UNSUPPORTED = -3, /* valid request this build has no kernel for */
  NF_ERR_NO_DEVICE = -4,   /* no gfx950 device visible */
  NF_ERR_COMM = -5,        /* RCCL missing or a collective failed */
  NF_ERR_STATE = -6        /* call order (e.g. log_prob before set_weights) */
} nf_status;

typedef enum nf_kind {
  /* src/real_nvp.py:71-171  split halves + flip, MLP conditioner with leaky_relu,
     (shift, log_scale) = split(out, 2) of width event_dim/2 each */
  NF_KIND_AFFINE_SPLIT = 0,
  /* src/real_nvp_toy_densities.py:74-110  distrax.MaskedCoupling(ScalarAffine):
     conditioner emits 2 numbers per sample, broadcast over every event dim */
  NF_KIND_AFFINE_MASKED = 1,
  /* src/maf_toy_densities.py:74-111, src/maf_mnist.py:52-89
     distrax.MaskedCoupling(RationalQuadraticSpline): conditioner emits
     event_dim * (3*bins+1) numbers */
  NF_KIND_RQS_MASKED = 2
} nf_kind;

typedef enum nf_activation { NF_ACT_RELU = 0, NF_ACT_LEAKY_RELU = 1 /* slope 0.01 */ } nf_activation;
typedef enum nf_base {
  NF_BASE_STD_NORMAL = 0, /* Independent(Normal(0,1)) / MultivariateNormalDiag(0,I) */
  NF_BASE_UNIFORM01 = 1   /* Independent(Uniform(0,1)): 0 inside [0,1]^D, -inf outside */
} nf_base;
typedef enum nf_mem { NF_MEM_HOST = 0, NF_MEM_DEVICE = 1 } nf_mem;

/* Which way a chain is walked and which per-layer map is applied. */
typedef enum nf_direction {
  /* distrax.Chain.forward_and_log_det: LAST layer first, each layer's forward map
     y_t = x_t*exp(ls)+shift (affine) or spline forward.  This is what
     Transformed(base, Inverse(Chain)).log_prob runs
     (src/real_nvp_toy_densities.py:104,123). */
  NF_DIR_CHAIN_FORWARD = 0,
  /* distrax.Chain.inverse_and_log_det: layer 0 first, each layer's inverse map.
     This is what Transformed.sample runs (src/real_nvp_toy_densities.py:136). */
  NF_DIR_CHAIN_INVERSE = 1,
  /* src/real_nvp.py:228-248 sample_nvp_chain: layer 0 first, forward_nvp. */
  NF_DIR_FIRST_TO_LAST_FORWARD = 2,
  /* src/real_nvp.py:251-283 log_prob_nvp_chains: last layer first, inverse_nvp. */
  NF_DIR_LAST_TO_FIRST_INVERSE = 3
} nf_direction;

typedef struct nf_flow_desc {
  int32_t abi_version;             /* NF_ABI_VERSION */
  int32_t kind;                    /* nf_kind */
  int32_t event_dim;               /* D: flattened event size (2 toy, 784 MNIST) */
  int32_t num_layers;              /* L coupling layers */
  int32_t num_hidden;              /* hidden Linear layers of the conditioner MLP */
  int32_t hidden[NF_MAX_HIDDEN];   /* their widths */
  int32_t activation;              /* nf_activation */
  /* masked kinds: layer l passes through (and shows the conditioner) the dims d
     with (d % 2 == first_mask_parity) XOR (l odd); reference value 1
     (src/real_nvp_toy_densities.py:80-82,102).
     split kind: layer l flips iff (l odd) XOR first_mask_parity; reference value 0
     (src/real_nvp.py:209-218). */
  int32_t first_mask_parity;
  int32_t base;                    /* nf_base */
  int32_t rqs_bins;                /* K (RQS kind only) */
  float rqs_range_min, rqs_range_max;
  float rqs_min_bin_size, rqs_min_knot_slope; /* distrax defaults 1e-4, 1e-4 */
} nf_flow_desc;

typedef struct nf_flow nf_flow;
typedef struct nf_comm nf_comm;

/* ---- library / device ------------------------------------------------- */
int nf_abi_version(void);
const char* nf_last_error(void);
int nf_device_count(int* out_count);
int nf_set_device(int device);
int nf_device_name(int device, char* buf, size_t buf_len);
int nf_device_synchronize(void);

/* plain device memory for callers that keep batches resident in HBM */
int nf_malloc(void** out_ptr, size_t bytes);
int nf_free(void* ptr);
int nf_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int nf_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);

/* ---- flow handle ------------------------------------------------------- */
/* Number of float32 values nf_flow_set_weights expects for `desc`
   (sum over layers and Linear modules of in*out + out). */
int nf_flow_weight_count(const nf_flow_desc* desc, size_t* out_count);

/* Replaces make_flow_model (src/real_nvp_toy_densities.py:74-110,
   src/maf_toy_densities.py:74-111, src/maf_mnist.py:52-89) and init_nvp_chain's
   structure (src/real_nvp.py:198-220): builds the chain on the current device.
   Kernel choice is by shape: 2-D masked flows with two relu hidden layers of at most 64 units run on the
   fused 64/64 kernels (every run of the reference uses 64/64: its command-line flags are parsed but never
   passed to main, src/real_nvp_toy_densities.py:236-253 vs :284-291, so the module constants :34-36 decide)
   -- narrower layers are completed with zero-weight units inside the handle, every
   weight / gradient vector that crosses this ABI keeps the layout `desc` describes (environment
   NF_NO_PAD=1: keep such flows on the general path); other shapes take the general path. */
int nf_flow_create(const nf_flow_desc* desc, nf_flow** out_flow);
void nf_flow_destroy(nf_flow* flow);

/* Host->device upload of all conditioner parameters.  `packed` is layer-major;
   inside a layer the Linear modules in creation order, each as W [in,out]
   row-major (y = x @ W + b, the haiku / src/real_nvp.py:190-194 orientation)
   followed by b [out].  For masked kinds the first Linear has in = event_dim and
   the last has out = 2 (affine) or event_dim*(3K+1) (RQS); for the split kind
   in = event_dim/2, out = event_dim. */
int nf_flow_set_weights(nf_flow* flow, const float* packed, size_t count);

/* Name of the kernel family the handle dispatches to ("affine2d_h64_mfma",
   "split2d_small", "generic_gemm"); for diagnostics and tests. */
int nf_flow_kernel_name(const nf_flow* flow, char* buf, size_t buf_len);
int nf_flow_synchronize(nf_flow* flow);

/* Per-handle options (nothing in the reference corresponds: jax picks its matmul
   precision globally).  Known names:
     "hidden_matmul" = "f32" (default): the hidden 64x64 Linear on the exact-float32 MFMA;
                     = "bf16x3": the same Linear as six bf16 MFMAs on a three-way bf16
                       split of both operands (dropped terms <= 2^-24 relative, the size
                       of float32's own product rounding).  About 1.8x the throughput.
                       Handles on the affine2d_h64_mfma kernel (its 64x64 Linear), the
                       fused spline kernel (both of its Linears) and the general path
                       (every Linear with K >= 32 and N >= 64); NF_ERR_UNSUPPORTED on the
                       split2d kernel, which has no MFMA.
   Applies to every later chain walk of the handle (log_prob, nll, sample, chain_apply).  The
   forward half of nf_nll_grad / nf_train_step on the fused kernels keeps the float32 kernels
   (the bf16x3 weight image is packed on the host; refreshing it after every optimiser step
   would cost a synchronise and two copies per step); the fused affine backward follows the
   mode for the two products that feed other products.
     "skip_nonfinite_updates" = "0" (default) | "1"
       "1": optax.apply_if_finite around nf_train_step's optimiser -- a step whose gradient
            holds a NaN or an infinity (one far outlier is enough for an unconstrained
            exp(log_scale) flow in float32) leaves weights and moments as they are, and Adam's
            step count (its bias correction) advances only on applied steps, as the wrapped
            optimiser's own count does; the loss returned is still the non-finite one.  "0" is what the reference's unwrapped
            optax.adam / sgd_update do: every weight becomes NaN. */
int nf_flow_set_option(nf_flow* flow, const char* name, const char* value);
int nf_flow_get_option(const nf_flow* flow, const char* name, char* buf, size_t buf_len);
/* The environment variable NF_HIDDEN_MATMUL=bf16x3 makes that value the default of every
   handle created afterwards on a kernel that has the mode. */

/* ---- the hot path -------------------------------------------------------- */
/* Replaces log_prob.apply -> distrax.Transformed.log_prob
   (src/real_nvp_toy_densities.py:113-123, src/maf_toy_densities.py:112-125,
   src/maf_mnist.py:109-118) and log_prob_nvp_chains with log_prob_N01
   (src/real_nvp.py:57-68,268-283).  out_log_prob[rows]. */
int nf_log_prob(nf_flow* flow, const float* x, int64_t rows, float* out_log_prob, int mem);

/* Same with the chain walked in an explicit nf_direction before the base
   log-density is added: Transformed(base, bijector).log_prob for a bijector that
   is the chain itself rather than Inverse(chain). */
int nf_chain_log_prob(nf_flow* flow, int direction, const float* x, int64_t rows,
                      float* out_log_prob, int mem);

/* Base log-density alone: Independent(Normal(0,1)).log_prob / log_prob_N01
   (src/real_nvp.py:57-68, src/real_nvp_toy_densities.py:105-108) or
   Independent(Uniform(0,1)).log_prob (src/maf_mnist.py:84-88).  No handle needed. */
int nf_base_log_prob(int base, int event_dim, const float* x, int64_t rows, float* out, int mem);

/* Base draw alone: Philox4x32-10 (counter = index of each group of four consecutive
   elements of the row-major [rows, event_dim] matrix, key = seed), Box-Muller normals or
   24-bit uniforms.  Replaces the base_distribution.sample / sample_N01 draw
   (src/real_nvp_toy_densities.py:105-108,136; src/real_nvp.py:42-54); the reference's JAX
   threefry stream itself is not reproduced. */
int nf_base_sample(int base, int event_dim, uint64_t seed, int64_t rows, float* out, int mem);

/* Replaces sample.apply -> distrax.Transformed.sample (src/real_nvp_toy_densities.py:126-136,
   197-203; src/maf_toy_densities.py:128-139; src/maf_mnist.py:121-131) and sample_nvp_chain
   (src/real_nvp.py:228-248): the base draw of nf_base_sample followed by the generative walk
   of the chain, all on the device.  out_x[rows, event_dim]. */
int nf_sample(nf_flow* flow, uint64_t seed, int64_t rows, float* out_x, int mem);

/* Synthetic 2-D training/evaluation rows with the distributions of the reference's host-side
   generators: make_dataset_banana (src/densities/banana.py:31-41), make_dataset_energy_1..4
   (src/densities/energies.py:47-127), make_dataset_gaussian_blobs
   (src/densities/gaussian_blobs.py:34-59).  out[rows, 2]; Philox4x32-10 keyed by seed. */
typedef enum nf_density {
  NF_DENSITY_BANANA = 0, NF_DENSITY_ENERGY1 = 1, NF_DENSITY_ENERGY2 = 2, NF_DENSITY_ENERGY3 = 3,
  NF_DENSITY_ENERGY4 = 4, NF_DENSITY_BLOBS = 5
} nf_density;
int nf_density_sample(int density, uint64_t seed, int64_t rows, float* out, int mem);

/* Replaces Chain / MaskedCoupling forward_and_log_det, inverse_and_log_det and
   forward_nvp / inverse_nvp chains (see nf_direction).  Walks layers
   [layer_begin, layer_end) of the chain; pass 0, num_layers for the whole chain,
   l, l+1 for the single MaskedCoupling l.  y[rows, D], logdet[rows]; either
   output may be NULL. */
int nf_chain_apply(nf_flow* flow, int direction, int layer_begin, int layer_end,
                   const float* x, int64_t rows, float* y, float* logdet, int mem);

/* Replaces the stand-alone evaluation of the INNER (element-wise, event_ndims_in = 0) bijectors the coupling layers
   are built from -- `forward_and_log_det` / `inverse_and_log_det` of the distrax.Bijector protocol
   (README.md:32-57) for distrax.ScalarAffine(shift, log_scale | scale) (src/real_nvp_toy_densities.py:86) and
   distrax.RationalQuadraticSpline(params, range_min, range_max) (src/maf_toy_densities.py:87, src/maf_mnist.py:66)
   -- on explicit parameter arrays: element e of x[n] is mapped with ITS OWN parameters,
     NF_ELEM_AFFINE_LOG_SCALE  params[n][2] = (shift, log_scale):  y = exp(log_scale) x + shift, logdet = log_scale
     NF_ELEM_AFFINE_SCALE      params[n][2] = (shift, scale):      y = scale x + shift,          logdet = log|scale|
     NF_ELEM_RQS               params[n][3K+1] (widths K | heights K | slopes K+1): the spline of nf_kind
                               NF_KIND_RQS_MASKED (same arithmetic, same kernels), linear tails outside the range,
   inverse != 0: the inverse map and ITS log-determinant (= minus the forward one at the preimage).
   out_y[n], out_logdet[n]; either may be NULL.  No handle: runs on the default stream and has completed on return.
   (Broadcasting shift / log_scale / params against x, as distrax does, is the caller's job:
   normalizing_flows_amd/bijectors.py materialises the broadcast arrays.) */
typedef enum nf_elementwise_kind { NF_ELEM_AFFINE_LOG_SCALE = 0, NF_ELEM_AFFINE_SCALE = 1, NF_ELEM_RQS = 2 } nf_elementwise_kind;
typedef struct nf_elementwise_desc {
  int32_t abi_version;             /* NF_ABI_VERSION */
  int32_t kind;                    /* nf_elementwise_kind */
  int32_t rqs_bins;                /* K (NF_ELEM_RQS only) */
  float rqs_range_min, rqs_range_max;
  float rqs_min_bin_size, rqs_min_knot_slope; /* distrax defaults 1e-4, 1e-4 */
} nf_elementwise_desc;
int nf_elementwise_apply(const nf_elementwise_desc* desc, int inverse, const float* x, const float* params,
                         int64_t n, float* out_y, float* out_logdet, int mem);

/* Replaces loss_fn / eval_fn: -mean(log_prob) (src/real_nvp_toy_densities.py:139-151,
   src/real_nvp.py:292-293).  Sums log_prob over the local rows in float64 in a
   fixed order; when a communicator is attached the local sum is all-reduced
   (RCCL, one float64) before dividing by rows_global.  out_log_prob may be NULL
   or receive the per-sample values as nf_log_prob does.  Blocking: *out_nll is
   valid on return. */
int nf_nll(nf_flow* flow, const float* x, int64_t rows_local, int64_t rows_global,
           float* out_log_prob, double* out_nll, int mem);

/* Same work, enqueue only (device pointers only): the scalar is written to
   slot `slot` (0..NF_NLL_SLOTS-1) of a pinned host ring the handle owns; read it
   with nf_nll_result after nf_flow_synchronize.  With a communicator attached the
   slots' local sums wait on the device and ONE all-reduce serves all waiting slots
   when nf_flow_synchronize / nf_nll_result next run -- those two calls are then
   collective: every rank must make them at the same points of its program. */
#define NF_NLL_SLOTS 64
int nf_nll_enqueue(nf_flow* flow, const float* x_dev, int64_t rows_local, int64_t rows_global,
                   float* out_log_prob_dev, int slot);
int nf_nll_result(nf_flow* flow, int slot, double* out_nll);

/* ---- training step (every masked coupling flow, and the split flow of
   src/real_nvp.py) ------------------------------------------------------------------------- */
typedef enum nf_optimizer_kind { NF_OPT_SGD = 0, NF_OPT_ADAM = 1 } nf_optimizer_kind;
typedef struct nf_optimizer {
  int32_t kind;          /* nf_optimizer_kind */
  float learning_rate;
  float beta1, beta2, eps; /* optax.adam defaults 0.9, 0.999, 1e-8 */
} nf_optimizer;

/* Replaces jax.value_and_grad(loss) (src/real_nvp.py:297) / jax.grad(loss_fn)
   (src/real_nvp_toy_densities.py:170): NLL of the local rows' share and d NLL / d weights in
   the layout of nf_flow_set_weights (all-reduced over the communicator when one is attached).
   out_grad[nf_flow_weight_count] lives where `mem` says. */
int nf_nll_grad(nf_flow* flow, const float* x, int64_t rows_local, int64_t rows_global, double* out_nll,
                float* out_grad, int mem);

/* Replaces `update` (src/real_nvp_toy_densities.py:164-173: grad, optax.adam update, apply_updates)
   and `sgd_update` (src/real_nvp.py:295-306): one optimisation step on the handle's weights,
   entirely on the device.  *out_nll is the loss BEFORE the step. */
int nf_train_step(nf_flow* flow, const float* x, int64_t rows_local, int64_t rows_global,
                  const nf_optimizer* opt, double* out_nll, int mem);

/* Current weights of the handle, layout of nf_flow_set_weights (host buffer). */
int nf_flow_get_weights(nf_flow* flow, float* packed, size_t count);

/* ---- stream timing (bench.py: HIP events on the stream the kernels run on) */
int nf_timer_start(nf_flow* flow);
int nf_timer_stop(nf_flow* flow, float* out_ms); /* synchronises on the stop event */

/* Diagnostic, not part of the path: streams a device [rows, 2] float32 matrix with the
   hot path's access shape so that rocprofv3's FETCH_SIZE can be calibrated on a known byte
   count (profiles/README.md). */
int nf_debug_read_f2(const float* x_dev, int64_t rows, float* out_dev);

/* Diagnostic, not part of the path: one Linear of the general path, C[m, n] = act(A[m, k] W[k, n] + bias)
   (act 0 none, 1 relu, 2 leaky_relu 0.01; the dispatch and kernels hk.nets.MLP / hk.Linear of
   src/maf_mnist.py:34-49 run on inside a handle), on device buffers the CALLER placed: the tests put C and W
   at the end of an allocation with sentinels behind them to pin what the kernels' buffer-descriptor range
   checks drop (tests/test_parity_gpu.py::test_gemm_range_check_canary).  Synchronises. */
int nf_debug_gemm(const float* a_dev, int lda, const float* w_dev, const float* bias_dev, float* c_dev, int ldc,
                  int m, int n, int k, int act);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ---------------------- */
/* librccl.so.1 is dlopen'ed on first use; a build box without it still loads
   libnf_hip.so.  id is an ncclUniqueId (128 bytes) made on rank 0 and handed to
   the other ranks by the host (bench.py uses torch.distributed's store). */
#define NF_COMM_ID_BYTES 128
/* What every rank checks BEFORE any rank enters nf_comm_init_rank: a usable HIP device is current and
   librccl loads with the symbols used here.  ncclCommInitRank blocks until every rank has entered it, so a
   rank that cannot get that far must be found first: the host exchanges the results of this call over its
   own (CPU) channel and all ranks give up together (normalizing_flows_amd/sharding.py: agree_all). */
int nf_comm_preflight(void);
int nf_comm_unique_id(unsigned char id[NF_COMM_ID_BYTES]);
int nf_comm_init_rank(nf_comm** out_comm, int world_size, int rank,
                      const unsigned char id[NF_COMM_ID_BYTES]);
int nf_comm_destroy(nf_comm* comm);
int nf_flow_attach_comm(nf_flow* flow, nf_comm* comm); /* NULL detaches */
/* Number of collectives (NLL-sum and gradient all-reduces) this handle has issued so far.  With a
   communicator attached nf_nll, nf_nll_result, nf_flow_synchronize, nf_nll_grad and nf_train_step are
   collective calls: one rank skipping one of them leaves the others waiting inside RCCL.  Ranks that
   exchange this count over their host channel at a point where it must agree (bench.py does after the
   timed region) turn such a program error into a message instead of a hang at the NEXT collective.
   The mean these collectives implement: src/real_nvp_toy_densities.py:143,150. */
int nf_flow_collective_count(const nf_flow* flow, uint64_t* out_count);

#ifdef __cplusplus
}
#endif
#endif /* NF_HIP_H */
