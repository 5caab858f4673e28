// Internal declarations shared by the translation units of libnf_hip.so.
// Not part of the ABI (that is include/nf_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "nf_hip.h"

namespace nf {

// ---- error plumbing -------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define NF_HIP(call)                                                        \
  do {                                                                      \
    hipError_t _e = (call);                                                 \
    if (_e != hipSuccess) return ::nf::hip_fail(_e, #call, __FILE__, __LINE__); \
  } while (0)

#define NF_TRY(call)        \
  do {                      \
    int _s = (call);        \
    if (_s != NF_OK) return _s; \
  } while (0)

// ---- kernels that ask for more than the default 64 KiB of dynamic LDS ----------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a (device, kernel) pair: set once per pair, whichever thread
// and whichever device gets there first.  A thread leaves this call only after the attribute IS set on its current
// device (a plain `static bool` let a second thread launch while the first was still inside hipFuncSetAttribute, and
// never set the attribute on a second device of the same process: VERDICT r03 weak 10).
int ensure_dynamic_lds(const void* kernel, int bytes);
// NF_DYN_LDS(bytes, kernel): the kernel last, so that a template-id's commas need no parentheses
#define NF_DYN_LDS(bytes, ...) NF_TRY(::nf::ensure_dynamic_lds(reinterpret_cast<const void*>(__VA_ARGS__), (bytes)))

// ---- environment switches ------------------------------------------------------------
// Read ONCE when a handle is created and kept in it; an ABI call installs its handle's snapshot for the calling
// thread (UseSwitches), launch code asks switches().  No getenv on a launch path.
struct Switches {
  bool force_generic = false, no_pad = false, no_zero_copy = false, no_host_pipeline = false;
  bool gemm_no_full = false, gemm_simple = false, no_splitk = false, no_k5 = false, rqs_simple = false;
  bool no_mlp_small = false, no_mlp_small_bwd = false, affine2d_no_msplit = false;
  bool no_rqs2d_bwd = false;             // NF_NO_RQS2D_BWD: the fused 2-D spline flow trains through the general plan
  bool no_small_m = false;               // NF_NO_SMALL_M: the general path's small-batch (M <= 256) kernels off
  bool no_compact_l1 = false;            // NF_NO_COMPACT_L1: the first Linear of wide masked flows over its whole (half-zero) k range
  int affine2d_tune[2] = {0, 0}, affine2d_x3_tune[2] = {0, 0}, rqs2d_tune[2] = {0, 0}, rqs2d_x3_tune[2] = {0, 0};  // T, WAVES; 0 = unset
  int split2d_samples = 0, split2d_grid = 0;  // 0 = unset
  int splitk_s = 0;                           // NF_SPLITK_S (development): number of k splits of the few-tile GEMMs
  bool no_tn_few_rows = false;                // NF_NO_TN_FEW_ROWS: 32 x 32 tiles for every weight gradient
  bool no_bwd_overlap = false;                // NF_NO_BWD_OVERLAP: weight gradients inside the chain, one stream
  bool no_small_m_rqs = false;                // NF_NO_SMALL_M_RQS: split-K slabs + rqs_rows_kernel at small batches
  int small_m_nt = 0;                         // NF_SMALL_M_NT = 1 | 2 | 4: column blocks per workgroup of gemm_small_m_kernel (any N); 8: any N, automatic form
  long long inline_tail_rows = 0;             // NF_INLINE_TAIL_LG (clamped to [0, 40]); 0 = unset
  static Switches from_env();
};
const Switches& switches();  // the snapshot of the handle whose call this thread is in (process default outside one)
struct UseSwitches {
  const Switches* prev;
  explicit UseSwitches(const Switches* s);
  ~UseSwitches();
};

// ---- what every chain kernel is told ------------------------------------------
constexpr int kBlockSumCap = 16384;  // per-block float64 partial sums one NLL slot holds: every kernel's max_grid() stays below
struct ChainIO {
  const float* x;          // [rows, D]
  const float* logdet_in;  // [rows] or null (0)
  float* y;                // [rows, D] or null
  float* logdet_out;       // [rows] or null
  float* logprob_out;      // [rows] or null: base log-density + logdet
  double* block_sums;      // [gridDim.x <= kBlockSumCap] or null: per-block sum of logprob (float64)
  long long rows;
  int layer_begin, layer_end;  // walk layers [begin, end)
  int descending;              // 1: end-1 .. begin, 0: begin .. end-1
  int inverse_op;              // 1: x = (y - shift) * exp(-ls), logdet -= ls
  int add_base;                // 1: logprob includes the base log-density
  int base;                    // nf_base
  float* tape;                 // [num_layers][rows][D] or null: the input of every walked layer, at that layer's index
                               // (the training forward of the fused spline flow: rqs2d_bwd reads it)
};

// ---- path 1: masked affine coupling, D = 2, conditioner 2 -> 64 -> 64 -> 2 ------
namespace affine2d {
// LDS / global image of ONE coupling layer, in floats.  Built on the host by
// pack_layer() so that every MFMA operand is a plain contiguous LDS read.
constexpr int kHidden = 64;
constexpr int kA2 = 0;                  // [32 kstep][64 lane][2 mtile]  W2 as MFMA A operand
constexpr int kW1B1 = 4096;             // [64 k][2]                      (w1[k], b1[k])
constexpr int kB2 = kW1B1 + 128;        // [2 half][2 mtile][16 reg]      b2 in accumulator order
constexpr int kW3 = kB2 + 64;           // [2 half][2 mtile][16 reg][2]   W3 rows in accumulator order
constexpr int kB3 = kW3 + 128;          // [2] + 2 pad
constexpr int kLayerFloats = kB3 + 4;   // 4420 floats = 17,680 B
constexpr int kMaxResidentLayers = 9;   // 9 * 17,680 + 64 <= 160 KiB of LDS
constexpr int kThreads = 512;           // 8 waves: two per SIMD

// cond_dim: which of the two event dims feeds the conditioner in this layer.
void pack_layer(const float* W1 /*[2][64]*/, const float* b1, const float* W2 /*[64][64]*/,
                const float* b2, const float* W3 /*[64][2]*/, const float* b3, int cond_dim,
                float* image /*[kLayerFloats]*/);

// wimg holds the images of ALL layers of the flow, layer-major.
int launch(const float* wimg, int first_mask_parity, const ChainIO& io, int grid_cap,
           hipStream_t stream, int* out_grid);
int max_grid();  // blocks the launch uses at most (= size of block_sums needed)
}  // namespace affine2d

// ---- path 1, optional "bf16x3" hidden matmul (nf_flow_set_option) ------------------------------
namespace affine2d_x3 {
// image of ONE layer (floats); streamed through an LDS double buffer
constexpr int kA = 0;                   // [3 part][4 kstep16][2 mtile][64 lane][8] bf16: W2 * 2^64 split hi/mid/lo
constexpr int kW1B1 = 3 * 4 * 2 * 64 * 8 / 2;  // 6144: [64 k][2] (w1, b1) * 2^-64
constexpr int kB2 = kW1B1 + 128;        // accumulator order, as affine2d
constexpr int kW3 = kB2 + 64;
constexpr int kB3 = kW3 + 128;
constexpr int kLayerFloats = kB3 + 4;   // 6468 floats = 25,872 B
void pack_layer(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                const float* b3, int cond_dim, float* image);
int launch(const float* wimg, int first_mask_parity, const ChainIO& io, hipStream_t stream, int* out_grid);
constexpr int kMaxGrid = 512;
}  // namespace affine2d_x3

// ---- backward of path 1 (training step) ---------------------------------
namespace affine2d_bwd {
// per-layer canonical weights: W1[2][64], b1[64], W2[64][64], b2[64], W3[64][2], b3[2]
constexpr int kCanonFloats = 4482;
// LDS image of one layer for the backward kernel (floats)
constexpr int oA2 = 0;              // [32 kstep][64 lane][2 mtile]    W2 * 2^64 (forward recompute)
constexpr int oA2T = 4096;          // [32 kstep'][64 lane][2 mtile']  W2 rows, k in accumulator order
constexpr int oW1B1 = 8192;         // [64 k][2]                        (w1, b1) * 2^-64
constexpr int oW1acc = oW1B1 + 128; // [2 half][2 mtile][16 reg]        w1 in accumulator order
constexpr int oB1acc = oW1acc + 64;
constexpr int oB2 = oB1acc + 64;
constexpr int oW3 = oB2 + 64;       // [2 half][2 mtile][16 reg][2]
constexpr int oB3 = oW3 + 128;
constexpr int kImageFloats = oB3 + 8;  // 8648
// one workgroup's partial gradient of one layer (floats)
constexpr int pW2 = 0;              // [2 mk][2 mj][16 reg][64 lane]  (still carries 2^-64)
constexpr int pB2 = 4096;           // [64 units]: the small tensors leave the kernel indexed by unit
constexpr int pW3 = pB2 + 64;       // [64 units][2]
constexpr int pB1 = pW3 + 128;
constexpr int pW1 = pB1 + 64;
constexpr int pB3 = pW1 + 64;
constexpr int kPartialFloats = pB3 + 4;  // 4420
void pack_layer(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                const float* b3, int cond_dim, float* image);
void build_gather_map(int cond_dim, int* map /*[kCanonFloats]*/, float* scale);
int grid_blocks();
// One launch for the whole chain (layer 0 first) + one gather for every layer's gradient.  `images`: the layers'
// images back to back -- float32 (kImageFloats each) or, x3, bf16x3 (x3_image_floats() each; launch_build_x3_images
// makes them from the float32 ones): the two products that feed other products then run as bf16x3.
// partials: [num_layers][grid_blocks()][kPartialFloats].
int launch_chain(const float* images, int num_layers, int first_mask_parity, float* v, float* G, float* partials,
                 long long rows, hipStream_t stream, int* out_parts, bool x3 = false);
int x3_image_floats();
int launch_build_x3_images(const float* img32, float* imgx3, int num_layers, hipStream_t stream);
int launch_gather(const float* partials, int nparts, int num_layers, int first_mask_parity, const int* map0,
                  const int* map1, const float* scale0, const float* scale1, float* grad, long long rows_global,
                  hipStream_t stream);
}  // namespace affine2d_bwd
int launch_adam(float* w, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                long long step, hipStream_t stream, const int* skip = nullptr,  // *skip != 0: leave everything as it is
                float* dev_bias = nullptr,  // [3] with skip: device-side step count and bias corrections (applied steps only)
                const unsigned* live = nullptr);  // one bit per entry or null: entries whose bit is 0 are left alone (plan layout)
int launch_sgd(float* w, const float* g, size_t n, float lr, hipStream_t stream, const int* skip = nullptr,
               const unsigned* live = nullptr);
int launch_nonfinite_flag(const float* g, size_t n, int* flag, hipStream_t stream);  // *flag = any NaN / inf in g
int launch_repack(const float* canon, const int* src, const signed char* shift, float* img, size_t n,
                  hipStream_t stream);
int launch_repack2(const float* canon, const int* src0, const signed char* shift0, float* img0, size_t n0,
                   const int* src1, const signed char* shift1, float* img1, size_t n1, hipStream_t stream);

// ---- path 1b: masked RQS coupling, D = 2, conditioner 2 -> 64 -> 64 -> 2*(3K+1) ----------
namespace rqs2d {
// image of ONE layer (floats); layers are streamed through an LDS double buffer.
// The last Linear (64 -> 3K+1 parameters of the transformed dim) is split by what the spline needs of it:
//   * the 2K bin-size logits and the two boundary slope logits: ONE 32-row MFMA tile (kA3; half-wave 0's 16
//     accumulator slots = K width logits, slot 14/15 = slope logits 0 / K; half-wave 1 the same with heights);
//   * of the K+1 slope logits only the two of the bin that holds the sample are ever used: their columns
//     are kept in accumulator order (kWS) and each lane gathers the pair of its bin after the bin search
//     (2 x 32 v_fma + one half-wave add instead of a second 32-row MFMA tile: 2048 MFMA cycles -> 280).
constexpr int kSlopePairs = 12;        // K <= 12 bins: K pairs (slope k, slope k+1) per hidden unit
constexpr int kA2 = 0;                 // [32 kstep][64 lane][2 mtile]   W2 (scaled 2^64)
constexpr int kA3 = 4096;              // [8 kgroup][64 lane][4 kstep']  bin-size + boundary-slope rows, k in accumulator order
constexpr int kWS = kA3 + 2048;        // [2 half][32 slot][kSlopePairs][2] slope columns k and k+1 side by side (one aligned
                                       // 8-byte read per hidden unit fetches the pair of bin k), units in accumulator order
constexpr int kW1B1 = kWS + 2 * 32 * kSlopePairs * 2;  // [64 k][2]
constexpr int kB2 = kW1B1 + 128;       // [2 half][2 mtile][16 reg]
constexpr int kB3 = kB2 + 64;          // [2 half][16 slot]               bias of the rows each half owns
constexpr int kBS = kB3 + 32;          // [kSlopePairs][2]                bias of slope logits k and k+1
constexpr int kLayerFloats = kBS + 2 * kSlopePairs;  // 7928 floats = 31,712 B
constexpr int kSlotSlopeFirst = 14, kSlotSlopeLast = 15;
void pack_layer(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                const float* b3, int cond_dim, int K, float* image);
int launch(const float* wimg, const nf_flow_desc& d, const ChainIO& io, hipStream_t stream, int* out_grid);
int max_grid();
bool supported_bins(int K);
}  // namespace rqs2d

// ---- backward of path 1b (training step of the 2-D spline flow) ------------------------------
// The `update` of src/maf_toy_densities.py:168-176 for the flows the fused spline kernel runs: one launch walks the
// whole chain backwards (layer 0 first), recomputing the conditioner per 32-sample tile from the layer's recorded INPUT
// (the forward kernel's `tape`, 8 B per sample and layer: the float32 inverse of a spline is not accurate enough to
// rebuild it) -- no activation, no parameter tensor and no parameter gradient ever reaches HBM.
namespace rqs2d_bwd {
constexpr int kMaxParams = 37;          // 3K+1, K <= 12
// canonical weights of one layer: W1[2][64], b1[64], W2[64][64], b2[64], W3[64][2P], b3[2P]  (P = 3K+1)
inline int canon_floats(int K) { return 128 + 64 + 4096 + 64 + 130 * (3 * K + 1); }
// LDS image of one layer (floats): the forward kernel's operands for the recompute + the transposed ones
constexpr int oA2 = 0;                  // [32 kstep][64 lane][2 mtile]   W2 * 2^64, natural k (recompute of a2)
constexpr int oA2T = 4096;              // [32 kstep'][64 lane][2 mtile'] W2 rows, k in accumulator order (g1 = W2 g2)
constexpr int oA3 = 8192;               // [8 kgroup][64 lane][4 kstep']  bin-size + boundary-slope rows (rqs2d::kA3)
constexpr int oA3T = oA3 + 2048;        // [K kstep][64 lane][2 mtile]    W3 columns (width s | height s) by hidden unit
constexpr int oWS = oA3T + 12 * 128;    // [2 half][32 slot][12][2]       slope column pairs (rqs2d::kWS)
constexpr int oW1B1 = oWS + 2 * 32 * 12 * 2;  // [64 k][2]  (w1, b1) * 2^-64
constexpr int oW1acc = oW1B1 + 128;     // [2 half][2 mtile][16 reg]      w1 in accumulator order
constexpr int oB1acc = oW1acc + 64;
constexpr int oB2 = oB1acc + 64;
constexpr int oB3 = oB2 + 64;           // [2 half][16 slot]
constexpr int oBS = oB3 + 32;           // [12][2]
constexpr int kImageFloats = oBS + 24;  // 13,688 floats = 54,752 B
// one workgroup's partial gradient of one layer (floats)
constexpr int pW2 = 0;                  // [2 mk][2 mj][16 reg][64 lane]  (still carries 2^-64)
constexpr int pB2 = 4096;               // [64 units]
constexpr int pB1 = pB2 + 64;
constexpr int pW1 = pB1 + 64;
constexpr int pW3 = pW1 + 64;           // [64 units][40]: the transformed dim's 3K+1 columns, canonical order
constexpr int pB3 = pW3 + 64 * 40;      // [40]
constexpr int kPartialFloats = pB3 + 40;  // 6,888
void pack_layer(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                int cond_dim, int K, float* image);
void build_gather_map(int cond_dim, int K, int* map /*[canon_floats(K)]*/, float* scale);
int grid_blocks();
bool supported_bins(int K);
struct SplineConst { float range_min, range_max, min_bin, min_slope, slope_offset; };
// tape [num_layers][rows][2]: every layer's input (layer l's slice = what the forward walk fed layer l);
// z [rows][2]: the chain's output; G [rows][2]: scratch (d loss_i / d state, in and out);
// partials [num_layers][grid_blocks()][kPartialFloats]
int launch_chain(const float* images, const nf_flow_desc& d, const float* tape, const float* z, float* G,
                 float* partials, long long rows, hipStream_t stream, int* out_parts);
int launch_gather(const float* partials, int nparts, const nf_flow_desc& d, const int* map0, const int* map1,
                  const float* scale0, const float* scale1, float* grad, long long rows_global, hipStream_t stream);
}  // namespace rqs2d_bwd

// ---- path 1b, optional "bf16x3" matmul: both MFMA passes on split-bf16 operands ----------------
namespace rqs2d_x3 {
constexpr int kA2 = 0;                  // [3 part][4 k16][2 mtile][64 lane][8] bf16   W2 * 2^64
constexpr int kA3 = 6144;               // same shape: last Linear, k in accumulator order
constexpr int kW1B1 = 12288;            // [64 k][2] (w1, b1) * 2^-64
constexpr int kB2 = kW1B1 + 128;
constexpr int kB3 = kB2 + 64;
constexpr int kLayerFloats = kB3 + 64;  // 12,544 floats = 50,176 B
void pack_layer(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                const float* b3, int cond_dim, int K, float* image);
int launch(const float* wimg, const nf_flow_desc& d, const ChainIO& io, hipStream_t stream, int* out_grid);
}  // namespace rqs2d_x3

// ---- path 2: split-halves affine coupling (variant A), D = 2, MLP 1 -> H -> 2 ----
namespace split2d {
constexpr int kMaxHidden = 64;
// per-layer device image: w0[H], b0[H], w1[H][2], b1[2]
__host__ __device__ inline int layer_floats(int H) { return 4 * H + 2; }
int launch(const float* wimg, int H, int activation, int first_flip, const ChainIO& io,
           hipStream_t stream, int* out_grid);
int max_grid();
int max_layers_per_launch(int H);  // the walked layers' weights are resident in LDS
// training step (src/real_nvp.py:295-306): the backward walk of the whole chain + one gradient gather
bool bwd_supported(int H);
int bwd_max_blocks();
int launch_bwd_chain(const float* wimg, int H, int num_layers, int activation, int first_flip, float* v, float* G,
                     float* partials, float* grad, long long rows, long long rows_global, hipStream_t stream);
}  // namespace split2d

// ---- path 3: general path (GEMM chain + transform kernels), any D / widths / kind ---
namespace generic {
struct Linear {
  int K = 0, N = 0;           // y[., N] = x[., K] @ W + b (after host-side pruning)
  size_t w_off = 0, b_off = 0;  // offsets into d_weights, floats
  size_t x3_off = 0;            // offset into d_wx3 (bf16 units); valid when Plan::x3
};
struct RqsConst {
  int K;
  float range_min, range_max, min_bin, min_slope, slope_offset;
};
RqsConst make_rqs_const(const nf_flow_desc& d);
// a_copy / act 5 (C += A W): only where gemm_small_m16_applicable() says so
int launch_gemm(const float* A, int lda, const float* W, const float* bias, float* Cm, int ldc, int M, int N, int K,
                int act, hipStream_t stream, const float* aux = nullptr, float* a_copy = nullptr);
// nf_gemm_small_m.hip: one Linear at M <= 256 rows in one launch (no split-K slabs, no reduction launch)
bool gemm_small_m_applicable(const float* A, int lda, int M, int N, int K);
bool gemm_small_m16_applicable(const float* A, int lda, int M, int N, int K);
int launch_gemm_small_m(const float* A, int lda, const float* W, const float* bias, float* Cm, int ldc, int M, int N,
                        int K, int act, hipStream_t stream, const float* aux, float* a_copy = nullptr);
// nf_small_m_rqs.hip: a masked spline layer's last Linear + the spline at M <= 256 rows in one launch; the log-dets
// arrive as [column group][row] partials (small_m_rqs_groups() of them per layer; 0: not a shape for that kernel)
int small_m_rqs_groups(const float* A, int lda, int M, int N, int Kdim, int Dt, int bins);
size_t small_m_rqs_packed_floats(int N, int Kdim, int bins);   // W re-laid in that kernel's operand order
int launch_small_m_rqs_pack(const float* W, float* Wp, int N, int Kdim, int bins, hipStream_t stream);
int launch_small_m_rqs(const float* A, int lda, const float* Wp, const float* bias, float* y, float* ldpart,
                       float* theta_out, int M, int N, int Kdim, int D, int par, int inverse_op, const RqsConst& rc,
                       hipStream_t stream);
int launch_transform(const nf_flow_desc& d, float* y, float* ld, const float* params, long long n, int par,
                     int inverse_op, const RqsConst& rc, hipStream_t stream);
// stand-alone ScalarAffine on explicit per-element parameters (nf_elementwise_apply); y / ld may be null
int launch_scalar_affine(const float* x, const float* params, float* y, float* ld, long long n, int inverse,
                         int params_are_scale, hipStream_t stream);

struct Plan {
  nf_flow_desc desc;
  std::vector<Linear> linears;  // [layer][linear]
  float* d_weights = nullptr;
  float* d_scratch = nullptr;
  long long scratch_rows = 0;
  int max_width = 0;
  int act_width = 0, param_width = 0;  // widest hidden layer / widest conditioner output
  // optional "bf16x3" matmul (nf_gemm_x3.hip): split weight images, built on first use
  bool x3 = false;
  unsigned short* d_wx3 = nullptr;
  std::vector<float> host_weights;  // what d_weights holds (after masking / pruning)
  std::vector<int> canon_index;     // per d_weights entry: index into the packed (canonical) vector, -1 = zero
  int set_x3(bool on);
  // ---- training (nf_generic_bwd.hip): masked kinds, chain in the log_prob direction ---------------
  struct Backward {
    bool ready = false;
    size_t n_weights = 0;            // floats in d_weights
    float* d_grad = nullptr;         // d(NLL)/d(d_weights), same layout
    float* d_wT = nullptr;           // every Linear's W transposed, [N][K], at wT_off
    std::vector<size_t> wT_off;
    int* d_canon_index = nullptr;
    unsigned* d_live_bits = nullptr;  // bit e: d_canon_index[e] >= 0 (what the optimiser reads: 1/32 of the index map's bytes)
    void* d_jobs = nullptr;          // per-Linear table of the one-launch transpose
    int transpose_blocks = 1;
    float* d_zero = nullptr;         // zero bias for the data-gradient GEMMs
    float* d_partials = nullptr;     // weight-gradient partial sums, one slab per row chunk
    float* d_small_partials = nullptr;  // narrow conditioners (nf_mlp_small_bwd.hip): one slab per workgroup
    size_t partial_floats = 0;
    float* d_scratch = nullptr;
    long long scratch_rows = 0;
    float* d_tape = nullptr;         // [num_layers][tape_rows][D]: every layer's input from the forward walk
    long long tape_rows = 0;
    float* d_acts = nullptr;         // every Linear's output of that walk: [tape_rows][N] at act_off[layer*nlin+j]
    std::vector<size_t> act_off;
    // small batches: the weight-gradient products leave the chain of data gradients and run beside it on a second
    // stream (Plan::backward); every layer then keeps its own gradient buffers so that nothing is overwritten under them
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> ev;      // one per layer + the side stream's last
    float* d_layer_grads = nullptr;  // [num_layers][rows][param_width + num_hidden * gwidth]
    long long layer_grad_rows = 0;
  } bwd;
  // SURVEY's K5 (spline layers whose last Linear is wide enough for the MFMA GEMM): weights of that Linear
  // re-laid so that a 128-column block holds whole transformed dims; the spline runs in the GEMM epilogue
  bool k5 = false;
  float* d_wk5 = nullptr;            // per layer: [Kdim][k5_npad] then the bias [k5_npad]
  size_t k5_layer_floats = 0;
  int k5_npad = 0;
  std::vector<size_t> k5_x3_off;     // per layer: the K5-layout split image inside d_wx3 (bf16x3 mode)
  bool k5_stale = false;             // an optimiser step changed d_weights: re-lay before the next fused walk
  // Masked kinds with a wide event (the 784-D flow): `where(mask, x, 0) @ W1` multiplies half of its k range by zero
  // rows.  Evaluation walks of large batches gather the pass-through dims of x into a dense [rows, w1c_k] matrix first
  // and multiply it with the live rows of W1 only (the same products in the same order: exact zeros leave a sum's
  // bits alone) -- half the first Linear's MFMA work for one streaming pass over x.
  bool w1c = false;
  int w1c_k[2] = {0, 0};             // compact k (pass-through dims, padded to a multiple of 4) for pass parity 0 / 1
  float* d_w1c = nullptr;            // per layer: [w1c_k[parity of the layer]][N]
  std::vector<size_t> w1c_off;
  bool w1c_stale = false;
  // Small batches (M <= 256) of a masked spline flow: the last Linears re-laid in the operand order of
  // linear_rqs_small_m_kernel (nf_small_m_rqs.hip) -- a second copy of those weights, made by the first walk that
  // needs it
  float* d_wsmr = nullptr;
  std::vector<size_t> wsmr_off;      // per layer, floats
  bool wsmr_stale = true;
  int ensure_wsmr(hipStream_t stream);
  size_t xc_off = 0;                 // floats into d_scratch where the gathered matrix starts (16-byte aligned)
  int refresh_w1c(hipStream_t stream);
  int refresh_k5(hipStream_t stream);  // from d_weights (after build; after optimiser steps on the next evaluation walk)
  bool record_tape = false;          // walk() fills bwd.d_tape (set around the training forward)
  int ensure_tape(long long rows);
  int prepare_backward();            // after build(): index map, transposes, buffers
  int refresh_from_canonical(const float* d_canon, hipStream_t stream);  // new canonical weights -> d_weights (+ below)
  // The optimiser works on d_weights itself (the plan's layout: masked rows and pruned columns are not there to be
  // updated -- their gradient is a structural zero in the reference too): after a step only the transposes follow ...
  int refresh_after_step(hipStream_t stream);
  // ... and the canonical vector is brought up to date when somebody asks for the weights
  int scatter_weights_to_canonical(float* d_canon, hipStream_t stream);
  // z [rows, D]: the chain's output of a walk() recorded with record_tape.  g [rows, D]: scratch.
  // The gradient d(NLL)/d(d_weights), NLL = -mean over rows_global, is left in bwd.d_grad (plan layout); grad_canon
  // [weight_count], if not null, also receives it in the packed (canonical) layout.
  int backward(const float* z, float* g, long long rows, long long rows_global, float* grad_canon,
               size_t weight_count, hipStream_t stream);
  int build(const nf_flow_desc& d, const float* packed);
  void release();
  int ensure_scratch(long long rows);
  // y [rows, D] and ld [rows] hold the chain's input state and are updated in place
  int walk(float* y, float* ld, const ChainIO& io, hipStream_t stream);
};
// Which 128 x 128 tile a GEMM workgroup takes (gx column blocks x gy row blocks, one workgroup each).
// Workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MiB L2, so (i) every XCD gets one
// contiguous run of the tile sequence and (ii) the sequence walks bands of kTileBand row blocks column by
// column: the ~64 workgroups resident on an XCD then cover an 8 x 8 patch -- the band's A rows (8 x 256 KiB
// at K = 512) stay in L2 and a column block's weights are fetched once per band, not once per row block.
// (Row-major order streamed the whole 20-30 MB weight image of the MNIST-shape last Linear through every
// XCD for every row block: 11-17 GB of fabric reads per launch against 0.5 GB algorithmic, rocprofv3
// FETCH_SIZE)
#ifndef NF_TILE_BAND
#define NF_TILE_BAND 8
#endif
constexpr int kTileBand = NF_TILE_BAND;
__device__ __forceinline__ void block_to_tile(int id, int gx, int gy, int& mb, int& nb) {
  const int nblk = gx * gy;
  const int base = nblk >> 3, rem = nblk & 7, x = id & 7;
  id = x * base + (x < rem ? x : rem) + (id >> 3);
  const int band = id / (kTileBand * gx);
  const int rows = gy - band * kTileBand < kTileBand ? gy - band * kTileBand : kTileBand;
  const int r = id - band * kTileBand * gx;
  nb = r / rows;
  mb = band * kTileBand + r - nb * rows;
}

// nf_mlp_small.hip: a whole conditioner (every Linear K, N <= 64) in one launch
bool mlp_small_applicable(const Linear* Ls, int nlin);
// tape != null (needs ldx == Ls[0].K): the input rows are also copied there, dense
int launch_mlp_small(const float* x, int ldx, const Linear* Ls, int nlin, const float* weights, int act,
                     float* const* outs, long long rows, hipStream_t stream, float* tape = nullptr);
// nf_mlp_small_bwd.hip: the MLP backward of such a conditioner in one launch (+ the workgroup reduction)
bool mlp_small_bwd_applicable(const Linear* Ls, int nlin, int D);
int mlp_small_bwd_max_blocks();
int launch_mlp_small_bwd(const float* x, int ldx, const Linear* Ls, int nlin, const float* weights,
                         const float* const* wT /*[nlin]: each Linear transposed, [N][K]*/,
                         const float* const* acts, const float* dth, int act, float* G, int D, float* partial,
                         float* grad, long long rows, hipStream_t stream, bool first_chunk);
size_t x3_image_halfwords(int K, int N);
void pack_w_x3(const float* W /*[K][N]*/, int K, int N, unsigned short* image);
bool gemm_x3_applicable(const float* A, int lda, int N, int K);
int launch_gemm_x3_rqs(const float* A, int lda, const unsigned short* wimg, const float* bias, float* y,
                       float* ldpart, int M, int Npad, int K, int D, int par, int inverse, const RqsConst& rc,
                       hipStream_t stream);
int launch_gemm_x3(const float* A, int lda, const unsigned short* wimg, const float* bias, float* Cm, int ldc,
                   int M, int N, int K, int act, hipStream_t stream);
int launch_base_log_prob(const float* z, const float* ld, float* lp, double* block_sums, long long rows,
                         int D, int base, hipStream_t stream, int* out_grid);
int launch_base_sample(float* out, long long rows, int D, int base, unsigned long long seed, hipStream_t stream);
int launch_density_sample(float* out, long long rows, int density, unsigned long long seed, hipStream_t stream);
int launch_debug_read_f2(const float* x, long long rows, float* out, hipStream_t stream);
}  // namespace generic

// ---- float64 fixed-order reduction of per-block sums --------------------------
int launch_sum_blocks(const double* block_sums, int n, double* out_sum, hipStream_t stream,
                      double* host_out = nullptr);  // host_out: device-mapped pinned slot written by the kernel itself

// ---- RCCL (dlopen'ed) ---------------------------------------------------------
int comm_allreduce_sum_f64(nf_comm* comm, double* dev_values, size_t count, hipStream_t stream);
int comm_allreduce_sum_f32(nf_comm* comm, float* dev_values, size_t count, hipStream_t stream);

}  // namespace nf
