// The handle behind the C ABI (struct nf_flow) and the helpers its translation units share: nf_api.hip (entry points,
// evaluation walks, options) and nf_train.hip (the training step: state set-up, forward + backward + gather, optimiser,
// weight read-back).  Not part of the ABI (that is include/nf_hip.h).
#pragma once

#include "nf_internal.h"

namespace nf {
enum Path { PATH_AFFINE2D_H64 = 0, PATH_SPLIT2D_SMALL = 1, PATH_GENERIC = 2, PATH_RQS2D_H64 = 3 };
}  // namespace nf

struct nf_flow {
  nf_flow_desc desc;
  nf::Switches sw;  // the environment switches as they stood when the handle was created
  int device = 0;
  int path = nf::PATH_GENERIC;
  hipStream_t stream = nullptr;
  // the NLL tail (sum of block sums, RCCL all-reduce, 8-byte copy to the host) runs on a side
  // stream so that the next batch's chain kernel does not wait behind a latency-bound collective
  hipStream_t side = nullptr;
  hipEvent_t ev_kernel[NF_NLL_SLOTS] = {}, ev_tail[NF_NLL_SLOTS] = {};
  bool tail_pending[NF_NLL_SLOTS] = {};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool have_weights = false;
  size_t weight_count = 0;
  // 2-D masked flows with two hidden layers narrower than 64 (the reference scripts default to 50) run on the
  // fused 64/64 kernels: hidden units are added with zero weights and biases (relu(0) = 0: they contribute
  // exact zeros to every sum, receive zero gradients and stay zero under SGD / Adam).  `desc` and
  // `weight_count` then describe the padded flow; the caller keeps seeing its own layout through pad_map.
  bool padded = false;
  size_t user_count = 0;       // weights in the caller's layout (== weight_count unless padded)
  std::vector<int> pad_map;    // caller's index -> padded index
  int* d_pad_map = nullptr;
  float* d_pad_tmp = nullptr;  // [user_count]
  // option "skip_nonfinite_updates" (optax.apply_if_finite): a step whose gradient holds a NaN or an infinity
  // leaves weights and optimiser moments as they are
  bool skip_nonfinite = false;
  int* d_skip_flag = nullptr;
  float* d_adam_bias = nullptr;  // [3]: (1 - b1^t, 1 - b2^t, t) with t = optimiser steps actually applied
  float* d_wimg = nullptr;  // path-specific packed weights
  // PATH_AFFINE2D_H64 with nf_flow_set_option("hidden_matmul", "bf16x3"): a second image, rebuilt
  // lazily from the canonical weights whenever they changed
  bool x3 = false, x3_stale = true;
  float* d_wimg_x3 = nullptr;
  // PATH_RQS2D_H64 trains through the general plan (nf_generic_bwd.hip); the fused image is then
  // rebuilt from the canonical weights the next time the fused kernel is asked for
  bool wimg_stale = false;
  nf::generic::Plan plan;       // PATH_GENERIC
  std::vector<float> h_weights;  // canonical weights as last set from the host
  // training state (PATH_AFFINE2D_H64; allocated on first use)
  struct Train {
    float *d_canon = nullptr, *d_grad = nullptr, *d_m = nullptr, *d_v = nullptr;
    float* d_bwd_img = nullptr;     // [L][affine2d_bwd::kImageFloats]
    float* d_bwd_img_x3 = nullptr;  // [L][x3_image_floats()], built from d_bwd_img when the handle's mode is bf16x3
    bool bwd_x3_stale = true;
    float* d_partials = nullptr;    // [layers][workgroups][kPartialFloats] (split flow: its own layout)
    int* d_gmap[2] = {nullptr, nullptr};      // gather maps for cond_dim 0 / 1
    float* d_gscale[2] = {nullptr, nullptr};
    int *d_fsrc = nullptr, *d_bsrc = nullptr;                // image <- canon maps (forward / backward images)
    signed char *d_fshift = nullptr, *d_bshift = nullptr;
    float *d_state_v = nullptr, *d_state_g = nullptr;
    long long state_rows = 0;
    // the fused spline flow (rqs2d_bwd): its forward walk records every layer's input, [L][tape_rows][2]
    bool rqs_fused = false;
    float* d_tape = nullptr;
    long long tape_rows = 0;
    long long step = 0;
    bool ready = false, dirty = false;  // dirty: device weights are newer than h_weights
    // flows that train through the general plan: the optimiser's vectors (d_m, d_v) and its weights (plan.d_weights,
    // gradient plan.bwd.d_grad) are in the PLAN's layout, opt_count floats; d_canon follows when the weights are asked for
    bool plan_layout = false, canon_stale = false;
    size_t opt_count = 0;
  } train;
  // staging for NF_MEM_HOST calls and for chains longer than one launch
  float *d_x = nullptr, *d_y = nullptr, *d_ld = nullptr, *d_lp = nullptr;
  long long stage_rows = 0;
  double* d_block_sums = nullptr;
  double* d_sum = nullptr;
  double* h_ring = nullptr;  // pinned, NF_NLL_SLOTS
  double* h_ring_dev = nullptr;  // the same memory as the device sees it (hipHostGetDevicePointer), or null
  // small host-buffer calls of the fused 2-D kernels (the reference evaluates batches of 128): the kernel reads x from
  // and writes log_prob to pinned, device-mapped host memory -- no staging copies (lazily allocated)
  float* h_zc = nullptr;         // [kZeroCopyRows * 2 | kZeroCopyRows]
  float* h_zc_dev = nullptr;
  long long ring_rows[NF_NLL_SLOTS];
  nf_comm* comm = nullptr;
  // Every collective this handle issues goes through flow_allreduce_*: one communicator serves both streams
  // (NLL sums on `side`, gradients on `stream`), so each collective waits for the previous one's event when
  // that ran on the other stream -- the order on the wire is the order of the calls on every rank -- and the
  // count lets ranks cross-check that they issued the same number (nf_flow_collective_count).
  unsigned long long collectives = 0;
  hipEvent_t ev_coll = nullptr;
  hipEvent_t ev_pipe[4] = {};   // large host-buffer calls in chunks: copies on the side stream beside the kernels
  hipStream_t coll_stream = nullptr;  // stream of the last collective, nullptr before the first
  // slots whose local sum has not been all-reduced yet (bit per slot), and the staging for the one
  // batched all-reduce that serves them
  unsigned long long reduce_pending = 0;
  double* d_reduce = nullptr;   // NF_NLL_SLOTS
  double* h_reduce = nullptr;   // pinned, NF_NLL_SLOTS
};


// nf_nll_enqueue: at most this many rows keep the NLL tail on the handle's main stream (see there)
constexpr long long kInlineTailRows = 1 << 16;
constexpr long long kZeroCopyRows = 1 << 13;   // 8192 rows: 64 KB in, 32 KB out over the mapping
constexpr long long kPipeChunkRows = 1 << 18;  // large host-buffer calls of the fused 2-D kernels: 2 MB in, 1 MB out per chunk

namespace nf {
int ensure_stage(nf_flow* f, long long rows);
void direction_flags(int direction, int* descending, int* inverse_op);
int default_direction(const nf_flow* f);
int launch_gather(float* dst, const float* src, const int* idx, size_t n, hipStream_t stream);
// Enqueue a chain walk on device pointers (nf_api.hip).  allow_x3 = false: the float32 kernels whatever the handle's mode
int run_chain_device(nf_flow* f, const ChainIO& io, int* out_grid, bool allow_x3 = true);
int run_chain_plan(nf_flow* f, const ChainIO& io, int* out_grid);
int flow_allreduce_f64(nf_flow* f, double* dev_values, size_t count, hipStream_t stream);
int flow_allreduce_f32(nf_flow* f, float* dev_values, size_t count, hipStream_t stream);
int check_ready(nf_flow* f, const void* x, long long rows);
int check_device_ptr(const void* p, const char* what);
// the handle's current weights in its own (padded) layout; downloads them after training steps (nf_train.hip)
int get_weights_internal(nf_flow* f, float* packed, size_t count);
}  // namespace nf
