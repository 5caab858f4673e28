// C ABI of libnf_hip.so (include/nf_hip.h): handles, weight packing, dispatch.
#include "nf_handle.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <set>
#include <unordered_set>
#include <utility>

namespace nf {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  (void)hipGetLastError();  // reported here: not again by the next launch's hipGetLastError() in some other call
  set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  return e == hipErrorNoDevice ? NF_ERR_NO_DEVICE : NF_ERR_HIP;
}

int ensure_dynamic_lds(const void* kernel, int bytes) {
  int dev = 0;
  NF_HIP(hipGetDevice(&dev));
  // fast path: what THIS thread has already seen done on this device (no lock; filled only after the global set says so)
  struct Seen { int dev = -1; std::unordered_set<const void*> kernels; };
  static thread_local Seen seen;
  if (seen.dev != dev) { seen.dev = dev; seen.kernels.clear(); }
  if (seen.kernels.count(kernel)) return NF_OK;
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  {
    std::lock_guard<std::mutex> lock(mu);  // held across the HIP call: a second thread waits until the attribute is set
    if (!done.count({dev, kernel})) {
      NF_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      done.insert({dev, kernel});
    }
  }
  seen.kernels.insert(kernel);
  return NF_OK;
}

Switches Switches::from_env() {
  Switches s;
  auto on = [](const char* name) { return getenv(name) != nullptr; };
  s.force_generic = on("NF_FORCE_GENERIC");
  s.no_pad = on("NF_NO_PAD");
  s.no_zero_copy = on("NF_NO_ZERO_COPY");
  s.no_host_pipeline = on("NF_NO_HOST_PIPELINE");
  s.gemm_no_full = on("NF_GEMM_NO_FULL");
  s.gemm_simple = on("NF_GEMM_SIMPLE");
  s.no_splitk = on("NF_NO_SPLITK");
  s.no_k5 = on("NF_NO_K5");
  s.rqs_simple = on("NF_RQS_SIMPLE");
  s.no_mlp_small = on("NF_NO_MLP_SMALL");
  s.no_mlp_small_bwd = on("NF_NO_MLP_SMALL_BWD");
  s.affine2d_no_msplit = on("NF_AFFINE2D_NO_MSPLIT");
  s.no_rqs2d_bwd = on("NF_NO_RQS2D_BWD");
  s.no_small_m = on("NF_NO_SMALL_M");
  s.no_compact_l1 = on("NF_NO_COMPACT_L1");
  auto pair = [](const char* name, int* out) {
    if (const char* e = getenv(name)) {
      int t = 0, w = 0;
      if (sscanf(e, "%d,%d", &t, &w) == 2 && t > 0 && w > 0) { out[0] = t; out[1] = w; }
    }
  };
  pair("NF_AFFINE2D_TUNE", s.affine2d_tune);
  pair("NF_AFFINE2D_X3_TUNE", s.affine2d_x3_tune);
  pair("NF_RQS2D_TUNE", s.rqs2d_tune);
  pair("NF_RQS2D_X3_TUNE", s.rqs2d_x3_tune);
  if (const char* e = getenv("NF_SPLIT2D_SAMPLES")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 8) s.split2d_samples = v; }
  if (const char* e = getenv("NF_SPLIT2D_GRID")) { const int v = atoi(e); if (v >= 1 && v <= kBlockSumCap) s.split2d_grid = v; }
  s.no_small_m_rqs = on("NF_NO_SMALL_M_RQS");
  s.no_bwd_overlap = on("NF_NO_BWD_OVERLAP");
  s.no_tn_few_rows = on("NF_NO_TN_FEW_ROWS");
  if (const char* e = getenv("NF_SMALL_M_NT")) s.small_m_nt = atoi(e);
  if (const char* e = getenv("NF_SPLITK_S")) { const int v = atoi(e); s.splitk_s = v < 0 ? 0 : (v > 64 ? 64 : v); }
  if (const char* e = getenv("NF_INLINE_TAIL_LG")) {   // development; a shift count outside [0, 63] would be undefined
    int lg = atoi(e);
    lg = lg < 0 ? 0 : (lg > 40 ? 40 : lg);
    s.inline_tail_rows = 1LL << lg;
  }
  return s;
}

static thread_local const Switches* g_switches = nullptr;
const Switches& switches() {
  if (g_switches) return *g_switches;
  static const Switches process_default = Switches::from_env();  // handle-less entry points; thread-safe static init
  return process_default;
}
UseSwitches::UseSwitches(const Switches* s) : prev(g_switches) { g_switches = s; }
UseSwitches::~UseSwitches() { g_switches = prev; }


static const char* path_name(int p) {
  switch (p) {
    case PATH_AFFINE2D_H64: return "affine2d_h64_mfma";
    case PATH_SPLIT2D_SMALL: return "split2d_small";
    case PATH_RQS2D_H64: return "rqs2d_h64_mfma";
    default: return "generic_gemm";
  }
}

// sizes of the Linear modules of one coupling layer's conditioner: in[j], out[j]
static void linear_shapes(const nf_flow_desc& d, std::vector<int>& in, std::vector<int>& out) {
  in.clear();
  out.clear();
  int first_in, last_out;
  if (d.kind == NF_KIND_AFFINE_SPLIT) {
    first_in = d.event_dim / 2;
    last_out = d.event_dim;
  } else if (d.kind == NF_KIND_AFFINE_MASKED) {
    first_in = d.event_dim;
    last_out = 2;
  } else {
    first_in = d.event_dim;
    last_out = d.event_dim * (3 * d.rqs_bins + 1);
  }
  int cur = first_in;
  for (int j = 0; j < d.num_hidden; ++j) {
    in.push_back(cur);
    out.push_back(d.hidden[j]);
    cur = d.hidden[j];
  }
  in.push_back(cur);
  out.push_back(last_out);
}

static int validate_desc(const nf_flow_desc* d) {
  if (!d) { set_error("desc is NULL"); return NF_ERR_INVALID; }
  if (d->abi_version != NF_ABI_VERSION) {
    set_error("desc.abi_version %d, library speaks %d", d->abi_version, NF_ABI_VERSION);
    return NF_ERR_INVALID;
  }
  if (d->kind < NF_KIND_AFFINE_SPLIT || d->kind > NF_KIND_RQS_MASKED) {
    set_error("unknown flow kind %d", d->kind); return NF_ERR_INVALID;
  }
  if (d->event_dim < 2) { set_error("event_dim must be >= 2, got %d", d->event_dim); return NF_ERR_INVALID; }
  if (d->kind == NF_KIND_AFFINE_SPLIT && (d->event_dim & 1)) {
    set_error("split coupling needs an even event_dim, got %d", d->event_dim); return NF_ERR_INVALID;
  }
  if (d->num_layers < 1) { set_error("num_layers must be >= 1"); return NF_ERR_INVALID; }
  if (d->num_hidden < 1 || d->num_hidden > NF_MAX_HIDDEN) {
    set_error("num_hidden must be 1..%d, got %d", NF_MAX_HIDDEN, d->num_hidden); return NF_ERR_INVALID;
  }
  for (int j = 0; j < d->num_hidden; ++j)
    if (d->hidden[j] < 1) { set_error("hidden[%d] must be positive", j); return NF_ERR_INVALID; }
  if (d->activation != NF_ACT_RELU && d->activation != NF_ACT_LEAKY_RELU) {
    set_error("unknown activation %d", d->activation); return NF_ERR_INVALID;
  }
  if (d->base != NF_BASE_STD_NORMAL && d->base != NF_BASE_UNIFORM01) {
    set_error("unknown base distribution %d", d->base); return NF_ERR_INVALID;
  }
  if (d->first_mask_parity != 0 && d->first_mask_parity != 1) {
    set_error("first_mask_parity must be 0 or 1"); return NF_ERR_INVALID;
  }
  if (d->kind == NF_KIND_RQS_MASKED) {
    // the checks distrax.RationalQuadraticSpline makes in its constructor
    if (d->rqs_bins < 1) { set_error("rqs_bins must be >= 1"); return NF_ERR_INVALID; }
    if (!(d->rqs_range_min < d->rqs_range_max)) { set_error("rqs range_min must be below range_max"); return NF_ERR_INVALID; }
    if (d->rqs_min_bin_size * d->rqs_bins > d->rqs_range_max - d->rqs_range_min) {
      set_error("rqs_min_bin_size * bins exceeds the range"); return NF_ERR_INVALID;
    }
    if (!(d->rqs_min_knot_slope < 1.f) || !(d->rqs_min_knot_slope > 0.f)) {
      set_error("rqs_min_knot_slope must be in (0,1)"); return NF_ERR_INVALID;
    }
  }
  return NF_OK;
}

}  // namespace nf

using namespace nf;



namespace nf {

int ensure_stage(nf_flow* f, long long rows) {
  if (rows <= f->stage_rows) return NF_OK;
  const int D = f->desc.event_dim;
  if (f->d_x) { hipFree(f->d_x); hipFree(f->d_y); hipFree(f->d_ld); hipFree(f->d_lp); }
  f->d_x = f->d_y = f->d_ld = f->d_lp = nullptr;
  f->stage_rows = 0;
  NF_HIP(hipMalloc(&f->d_x, sizeof(float) * rows * D));
  NF_HIP(hipMalloc(&f->d_y, sizeof(float) * rows * D));
  NF_HIP(hipMalloc(&f->d_ld, sizeof(float) * rows));
  NF_HIP(hipMalloc(&f->d_lp, sizeof(float) * rows));
  f->stage_rows = rows;
  return NF_OK;
}

void direction_flags(int direction, int* descending, int* inverse_op) {
  switch (direction) {
    case NF_DIR_CHAIN_FORWARD: *descending = 1; *inverse_op = 0; break;
    case NF_DIR_CHAIN_INVERSE: *descending = 0; *inverse_op = 1; break;
    case NF_DIR_FIRST_TO_LAST_FORWARD: *descending = 0; *inverse_op = 0; break;
    default: *descending = 1; *inverse_op = 1; break;
  }
}

__global__ __launch_bounds__(256) void gather_f32_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                         const int* __restrict__ idx, size_t n) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) dst[q] = src[idx[q]];
}
int launch_gather(float* dst, const float* src, const int* idx, size_t n, hipStream_t stream) {
  hipLaunchKernelGGL(gather_f32_kernel, dim3((unsigned)std::min<size_t>(1024, (n + 255) / 256)), dim3(256), 0, stream, dst,
                     src, idx, n);
  NF_HIP(hipGetLastError());
  return NF_OK;
}

static int refresh_x3_image(nf_flow* f) {
  const nf_flow_desc& d = f->desc;
  std::vector<float> w(f->weight_count);
  NF_TRY(get_weights_internal(f, w.data(), w.size()));
  const size_t per_layer = f->weight_count / d.num_layers;
  const bool rqs = f->path == PATH_RQS2D_H64;
  const size_t image_floats = rqs ? rqs2d_x3::kLayerFloats : affine2d_x3::kLayerFloats;
  const int P = 3 * d.rqs_bins + 1;
  std::vector<float> img((size_t)d.num_layers * image_floats);
  for (int l = 0; l < d.num_layers; ++l) {
    const float* W1 = w.data() + per_layer * l;
    const float* b1 = W1 + 2 * 64;
    const float* W2 = b1 + 64;
    const float* b2 = W2 + 64 * 64;
    const float* W3 = b2 + 64;
    const float* b3 = W3 + 64 * (rqs ? 2 * P : 2);
    const int cond_dim = (d.first_mask_parity + l) & 1;
    if (rqs) rqs2d_x3::pack_layer(W1, b1, W2, b2, W3, b3, cond_dim, d.rqs_bins, img.data() + l * image_floats);
    else affine2d_x3::pack_layer(W1, b1, W2, b2, W3, b3, cond_dim, img.data() + l * image_floats);
  }
  NF_HIP(hipStreamSynchronize(f->stream));
  if (!f->d_wimg_x3) NF_HIP(hipMalloc(&f->d_wimg_x3, img.size() * sizeof(float)));
  NF_HIP(hipMemcpy(f->d_wimg_x3, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
  f->x3_stale = false;
  return NF_OK;
}

int run_chain_plan(nf_flow* f, const ChainIO& io, int* out_grid);
static int refresh_rqs_image(nf_flow* f);

// Enqueue a chain walk on device pointers.  out_grid = number of block sums written.
// allow_x3 = false: the float32 kernels whatever the handle's matmul mode (the forward half of the training step:
// the bf16x3 image is packed on the host, so refreshing it after every optimiser step would cost a synchronise and
// two copies per step; include/nf_hip.h documents that this walk keeps float32).
int run_chain_device(nf_flow* f, const ChainIO& io_in, int* out_grid, bool allow_x3) {
  ChainIO io = io_in;
  io.base = f->desc.base;
  if (f->path == PATH_RQS2D_H64 && f->wimg_stale) NF_TRY(refresh_rqs_image(f));
  if (allow_x3 && f->x3 && (f->path == PATH_AFFINE2D_H64 || f->path == PATH_RQS2D_H64)) {
    if (f->x3_stale) NF_TRY(refresh_x3_image(f));
    if (f->path == PATH_RQS2D_H64) return rqs2d_x3::launch(f->d_wimg_x3, f->desc, io, f->stream, out_grid);
    return affine2d_x3::launch(f->d_wimg_x3, f->desc.first_mask_parity, io, f->stream, out_grid);
  }
  // Kernels that keep the walked layers' weights in LDS take a bounded number of layers per launch: longer walks
  // are several launches, state carried in device scratch.
  auto segmented = [&](int max_layers, auto&& launch_one) -> int {
    const int nlay = io.layer_end - io.layer_begin;
    if (nlay <= max_layers) return launch_one(io);
    NF_TRY(ensure_stage(f, io.rows));
    const float* cur_x = io.x;
    const float* cur_ld = io.logdet_in;
    int done = 0;
    while (done < nlay) {
      const int n = std::min(max_layers, nlay - done);
      const bool last = done + n == nlay;
      ChainIO seg = io;
      if (io.descending) { seg.layer_end = io.layer_end - done; seg.layer_begin = seg.layer_end - n; }
      else { seg.layer_begin = io.layer_begin + done; seg.layer_end = seg.layer_begin + n; }
      seg.x = cur_x;
      seg.logdet_in = cur_ld;
      if (!last) {
        seg.y = f->d_y; seg.logdet_out = f->d_ld; seg.logprob_out = nullptr;
        seg.block_sums = nullptr; seg.add_base = 0;
      }
      NF_TRY(launch_one(seg));
      cur_x = f->d_y;
      cur_ld = f->d_ld;
      done += n;
    }
    return NF_OK;
  };
  if (f->path == PATH_SPLIT2D_SMALL)
    return segmented(split2d::max_layers_per_launch(f->desc.hidden[0]), [&](const ChainIO& seg) {
      return split2d::launch(f->d_wimg, f->desc.hidden[0], f->desc.activation, f->desc.first_mask_parity, seg,
                             f->stream, out_grid);
    });
  if (f->path == PATH_RQS2D_H64) return rqs2d::launch(f->d_wimg, f->desc, io, f->stream, out_grid);
  if (f->path == PATH_AFFINE2D_H64)
    return segmented(affine2d::kMaxResidentLayers, [&](const ChainIO& seg) {
      return affine2d::launch(f->d_wimg, f->desc.first_mask_parity, seg, 0, f->stream, out_grid);
    });
  return run_chain_plan(f, io, out_grid);
}

// general path: state lives in y / ld and is updated in place, layer by layer
int run_chain_plan(nf_flow* f, const ChainIO& io, int* out_grid) {
  const int D = f->desc.event_dim;
  if (!io.y || !io.logdet_out) NF_TRY(ensure_stage(f, io.rows));
  float* y = io.y ? io.y : f->d_y;
  float* ld = io.logdet_out ? io.logdet_out : f->d_ld;
  if (y != io.x)
    NF_HIP(hipMemcpyAsync(y, io.x, sizeof(float) * io.rows * D, hipMemcpyDeviceToDevice, f->stream));
  if (io.logdet_in) {
    if (ld != io.logdet_in)
      NF_HIP(hipMemcpyAsync(ld, io.logdet_in, sizeof(float) * io.rows, hipMemcpyDeviceToDevice, f->stream));
  } else {
    NF_HIP(hipMemsetAsync(ld, 0, sizeof(float) * io.rows, f->stream));
  }
  NF_TRY(f->plan.walk(y, ld, io, f->stream));
  if (io.logprob_out || io.block_sums)
    NF_TRY(generic::launch_base_log_prob(y, ld, io.logprob_out, io.block_sums, io.rows, D, io.base,
                                         f->stream, out_grid));
  return NF_OK;
}

// The fused spline kernel's weight image from canonical weights (host side).
static void pack_rqs_images(const nf_flow_desc& d, const float* packed, size_t per_layer, std::vector<float>& img) {
  const int P = 3 * d.rqs_bins + 1;
  img.resize((size_t)d.num_layers * rqs2d::kLayerFloats);
  for (int l = 0; l < d.num_layers; ++l) {
    const float* W1 = packed + per_layer * l;
    const float* b1 = W1 + 2 * 64;
    const float* W2 = b1 + 64;
    const float* b2 = W2 + 64 * 64;
    const float* W3 = b2 + 64;
    const float* b3 = W3 + 64 * 2 * P;
    rqs2d::pack_layer(W1, b1, W2, b2, W3, b3, (d.first_mask_parity + l) & 1, d.rqs_bins,
                      img.data() + (size_t)l * rqs2d::kLayerFloats);
  }
}

static int refresh_rqs_image(nf_flow* f) {
  std::vector<float> w(f->weight_count), img;
  NF_TRY(get_weights_internal(f, w.data(), w.size()));  // synchronises, downloads the trained weights
  pack_rqs_images(f->desc, w.data(), f->weight_count / f->desc.num_layers, img);
  NF_HIP(hipMemcpy(f->d_wimg, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
  f->wimg_stale = false;
  return NF_OK;
}

__global__ void select_pending_kernel(const double* __restrict__ sums, unsigned long long mask,
                                      double* __restrict__ out) {
  const int s = threadIdx.x;
  if (s < NF_NLL_SLOTS) out[s] = (mask >> s & 1ull) ? sums[s] : 0.0;
}

static int flow_collective_order(nf_flow* f, hipStream_t stream) {
  if (!f->ev_coll) NF_HIP(hipEventCreateWithFlags(&f->ev_coll, hipEventDisableTiming));
  if (f->coll_stream && f->coll_stream != stream) NF_HIP(hipStreamWaitEvent(stream, f->ev_coll, 0));
  return NF_OK;
}
static int flow_collective_done(nf_flow* f, hipStream_t stream) {
  NF_HIP(hipEventRecord(f->ev_coll, stream));
  f->coll_stream = stream;
  f->collectives += 1;
  return NF_OK;
}
int flow_allreduce_f64(nf_flow* f, double* dev_values, size_t count, hipStream_t stream) {
  NF_TRY(flow_collective_order(f, stream));
  NF_TRY(comm_allreduce_sum_f64(f->comm, dev_values, count, stream));
  return flow_collective_done(f, stream);
}
int flow_allreduce_f32(nf_flow* f, float* dev_values, size_t count, hipStream_t stream) {
  NF_TRY(flow_collective_order(f, stream));
  NF_TRY(comm_allreduce_sum_f32(f->comm, dev_values, count, stream));
  return flow_collective_done(f, stream);
}

// One all-reduce for every slot that waits for it (side stream, after their sums), then the host copy.
static int flush_pending_reductions(nf_flow* f) {
  const unsigned long long mask = f->reduce_pending;
  if (!mask) return NF_OK;
  if (!f->comm) { set_error("pending reductions without a communicator"); return NF_ERR_STATE; }
  hipLaunchKernelGGL(select_pending_kernel, dim3(1), dim3(64), 0, f->side, f->d_sum, mask, f->d_reduce);
  NF_HIP(hipGetLastError());
  NF_TRY(flow_allreduce_f64(f, f->d_reduce, NF_NLL_SLOTS, f->side));
  NF_HIP(hipMemcpyAsync(f->h_reduce, f->d_reduce, sizeof(double) * NF_NLL_SLOTS, hipMemcpyDeviceToHost, f->side));
  NF_HIP(hipStreamSynchronize(f->side));
  for (int s = 0; s < NF_NLL_SLOTS; ++s)
    if (mask >> s & 1ull) f->h_ring[s] = f->h_reduce[s];
  f->reduce_pending = 0;
  return NF_OK;
}

int check_ready(nf_flow* f, const void* x, long long rows) {
  if (!f) { set_error("flow handle is NULL"); return NF_ERR_INVALID; }
  if (!f->have_weights) { set_error("nf_flow_set_weights has not been called"); return NF_ERR_STATE; }
  if (rows < 0) { set_error("rows must be >= 0"); return NF_ERR_INVALID; }
  if (rows > 0 && !x) { set_error("x is NULL"); return NF_ERR_INVALID; }
  return NF_OK;
}

int check_device_ptr(const void* p, const char* name) {
  if (p && (reinterpret_cast<uintptr_t>(p) & 15)) {
    set_error("device pointer %s must be 16-byte aligned", name);
    return NF_ERR_INVALID;
  }
  return NF_OK;
}

}  // namespace nf

extern "C" {

int nf_abi_version(void) { return NF_ABI_VERSION; }

const char* nf_last_error(void) { return g_last_error.c_str(); }

int nf_device_count(int* out_count) {
  if (!out_count) { set_error("out_count is NULL"); return NF_ERR_INVALID; }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
  *out_count = n;
  return NF_OK;
}

int nf_set_device(int device) { NF_HIP(hipSetDevice(device)); return NF_OK; }

int nf_device_name(int device, char* buf, size_t buf_len) {
  if (!buf || buf_len == 0) { set_error("buf is NULL"); return NF_ERR_INVALID; }
  hipDeviceProp_t prop;
  NF_HIP(hipGetDeviceProperties(&prop, device));
  snprintf(buf, buf_len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return NF_OK;
}

int nf_device_synchronize(void) { NF_HIP(hipDeviceSynchronize()); return NF_OK; }

int nf_malloc(void** out_ptr, size_t bytes) {
  if (!out_ptr) { set_error("out_ptr is NULL"); return NF_ERR_INVALID; }
  *out_ptr = nullptr;
  NF_HIP(hipMalloc(out_ptr, bytes ? bytes : 16));
  return NF_OK;
}
int nf_free(void* ptr) { if (ptr) NF_HIP(hipFree(ptr)); return NF_OK; }
// Handles run on non-blocking streams, which the null stream does not order against: a copy (and the
// handle-less device-mode calls below) first waits for everything enqueued on the device, so that a result a
// handle has only enqueued (device in -> device out calls return before the kernel ran) is complete when it is
// read, and an upload cannot overtake a kernel that still reads the destination.
int nf_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  NF_HIP(hipDeviceSynchronize());
  NF_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return NF_OK;
}
int nf_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  NF_HIP(hipDeviceSynchronize());
  NF_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return NF_OK;
}

int nf_flow_weight_count(const nf_flow_desc* desc, size_t* out_count) {
  NF_TRY(validate_desc(desc));
  if (!out_count) { set_error("out_count is NULL"); return NF_ERR_INVALID; }
  std::vector<int> in, out;
  linear_shapes(*desc, in, out);
  size_t per_layer = 0;
  for (size_t j = 0; j < in.size(); ++j) per_layer += (size_t)in[j] * out[j] + out[j];
  *out_count = per_layer * desc->num_layers;
  return NF_OK;
}

int nf_flow_create(const nf_flow_desc* desc, nf_flow** out_flow) {
  if (!out_flow) { set_error("out_flow is NULL"); return NF_ERR_INVALID; }
  *out_flow = nullptr;
  NF_TRY(validate_desc(desc));
  int ndev = 0;
  nf_device_count(&ndev);
  if (ndev == 0) { set_error("no HIP device visible: libnf_hip has no CPU fallback"); return NF_ERR_NO_DEVICE; }
  nf_flow* f = new (std::nothrow) nf_flow();
  if (!f) { set_error("out of host memory"); return NF_ERR_INVALID; }
  f->desc = *desc;
  f->sw = Switches::from_env();
  nf_flow_weight_count(desc, &f->user_count);
  {
    const nf_flow_desc& u = *desc;
    const bool fused_kind = u.kind == NF_KIND_AFFINE_MASKED ||
                            (u.kind == NF_KIND_RQS_MASKED && rqs2d::supported_bins(u.rqs_bins) &&
                             !f->sw.force_generic);
    if (fused_kind && u.event_dim == 2 && u.num_hidden == 2 && u.activation == NF_ACT_RELU && u.hidden[0] <= 64 &&
        u.hidden[1] <= 64 && (u.hidden[0] < 64 || u.hidden[1] < 64) && !f->sw.no_pad) {
      f->padded = true;
      f->desc.hidden[0] = f->desc.hidden[1] = 64;
      std::vector<int> in, out;
      linear_shapes(u, in, out);  // the caller's shapes; the padded ones differ in hidden widths only
      const int pin[3] = {in[0], 64, 64}, pout[3] = {64, 64, out[2]};
      size_t pper = 0;
      for (int j = 0; j < 3; ++j) pper += (size_t)pin[j] * pout[j] + pout[j];
      f->pad_map.reserve(f->user_count);
      for (int l = 0; l < u.num_layers; ++l) {
        size_t base = pper * l;
        for (int j = 0; j < 3; ++j) {
          for (int k = 0; k < in[j]; ++k)
            for (int n = 0; n < out[j]; ++n) f->pad_map.push_back((int)(base + (size_t)k * pout[j] + n));
          base += (size_t)pin[j] * pout[j];
          for (int n = 0; n < out[j]; ++n) f->pad_map.push_back((int)(base + n));
          base += pout[j];
        }
      }
    }
  }
  nf_flow_weight_count(&f->desc, &f->weight_count);
  const nf_flow_desc& d = f->desc;
  if (d.kind == NF_KIND_AFFINE_MASKED && d.event_dim == 2 && d.num_hidden == 2 &&
      d.hidden[0] == 64 && d.hidden[1] == 64 && d.activation == NF_ACT_RELU)
    f->path = PATH_AFFINE2D_H64;
  else if (d.kind == NF_KIND_AFFINE_SPLIT && d.event_dim == 2 && d.num_hidden == 1 &&
           d.hidden[0] <= split2d::kMaxHidden)
    f->path = PATH_SPLIT2D_SMALL;
  else if (d.kind == NF_KIND_RQS_MASKED && d.event_dim == 2 && d.num_hidden == 2 && d.hidden[0] == 64 &&
           d.hidden[1] == 64 && d.activation == NF_ACT_RELU && rqs2d::supported_bins(d.rqs_bins) &&
           !f->sw.force_generic)
    f->path = PATH_RQS2D_H64;
  else
    f->path = PATH_GENERIC;
  // NF_HIDDEN_MATMUL=bf16x3: process-wide default of the "hidden_matmul" option for the kernels
  // that have the mode (lets a whole test suite or script run under it unchanged)
  if (const char* e = getenv("NF_HIDDEN_MATMUL")) {
    if (strcmp(e, "bf16x3") == 0) {
      if (f->path == PATH_AFFINE2D_H64 || f->path == PATH_RQS2D_H64) f->x3 = true;
      else if (f->path == PATH_GENERIC) f->plan.x3 = true;
    } else if (strcmp(e, "f32") != 0 && e[0] != '\0') {
      set_error("NF_HIDDEN_MATMUL='%s': expected f32 or bf16x3", e);
      delete f;
      return NF_ERR_INVALID;
    }
  }
  int s = NF_OK;
  do {
    if (hipGetDevice(&f->device) != hipSuccess) { s = hip_fail(hipGetLastError(), "hipGetDevice", __FILE__, __LINE__); break; }
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&f->ev0)) != hipSuccess || (e = hipEventCreate(&f->ev1)) != hipSuccess ||
        (e = hipMalloc(&f->d_block_sums, sizeof(double) * kBlockSumCap * NF_NLL_SLOTS)) != hipSuccess ||
        (e = hipMalloc(&f->d_sum, sizeof(double) * NF_NLL_SLOTS)) != hipSuccess ||
        (e = hipMalloc(&f->d_reduce, sizeof(double) * NF_NLL_SLOTS)) != hipSuccess ||
        (e = hipHostMalloc(&f->h_reduce, sizeof(double) * NF_NLL_SLOTS, hipHostMallocDefault)) != hipSuccess ||
        (e = hipHostMalloc(&f->h_ring, sizeof(double) * NF_NLL_SLOTS, hipHostMallocDefault)) != hipSuccess) {
      s = hip_fail(e, "flow resources", __FILE__, __LINE__);
      break;
    }
    // the ring as the device addresses it (mapped, coherent host memory); without it the tail copies 8 bytes
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, f->h_ring, 0) == hipSuccess) f->h_ring_dev = static_cast<double*>(dev);
    else (void)hipGetLastError();
  } while (0);
  for (int q = 0; s == NF_OK && q < NF_NLL_SLOTS; ++q) {
    hipError_t e;
    if ((e = hipEventCreateWithFlags(&f->ev_kernel[q], hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&f->ev_tail[q], hipEventDisableTiming)) != hipSuccess)
      s = hip_fail(e, "hipEventCreate", __FILE__, __LINE__);
  }
  if (s != NF_OK) { nf_flow_destroy(f); return s; }
  memset(f->ring_rows, 0, sizeof(f->ring_rows));
  *out_flow = f;
  return NF_OK;
}

void nf_flow_destroy(nf_flow* f) {
  if (!f) return;
  if (f->stream) hipStreamSynchronize(f->stream);
  if (f->side) hipStreamSynchronize(f->side);
  for (int q = 0; q < NF_NLL_SLOTS; ++q) {
    if (f->ev_kernel[q]) hipEventDestroy(f->ev_kernel[q]);
    if (f->ev_tail[q]) hipEventDestroy(f->ev_tail[q]);
  }
  if (f->side) hipStreamDestroy(f->side);
  hipFree(f->d_wimg);
  hipFree(f->d_wimg_x3);
  {
    nf_flow::Train& t = f->train;
    hipFree(t.d_canon); hipFree(t.d_grad); hipFree(t.d_m); hipFree(t.d_v); hipFree(t.d_bwd_img); hipFree(t.d_bwd_img_x3);
    hipFree(t.d_partials); hipFree(t.d_gmap[0]); hipFree(t.d_gmap[1]); hipFree(t.d_gscale[0]);
    hipFree(t.d_gscale[1]); hipFree(t.d_fsrc); hipFree(t.d_bsrc); hipFree(t.d_fshift); hipFree(t.d_bshift);
    hipFree(t.d_state_v); hipFree(t.d_state_g); hipFree(t.d_tape);
  }
  f->plan.release();
  hipFree(f->d_x); hipFree(f->d_y); hipFree(f->d_ld); hipFree(f->d_lp);
  hipFree(f->d_block_sums); hipFree(f->d_sum); hipFree(f->d_reduce);
  if (f->h_zc) hipHostFree(f->h_zc);
  if (f->h_ring) hipHostFree(f->h_ring);
  if (f->h_reduce) hipHostFree(f->h_reduce);
  hipFree(f->d_pad_map);
  hipFree(f->d_pad_tmp);
  hipFree(f->d_skip_flag);
  hipFree(f->d_adam_bias);
  if (f->ev_coll) hipEventDestroy(f->ev_coll);
  for (int q = 0; q < 4; ++q)
    if (f->ev_pipe[q]) hipEventDestroy(f->ev_pipe[q]);
  if (f->ev0) hipEventDestroy(f->ev0);
  if (f->ev1) hipEventDestroy(f->ev1);
  if (f->stream) hipStreamDestroy(f->stream);
  delete f;
}

static int set_weights_internal(nf_flow* f, const float* packed, size_t count);

int nf_flow_set_weights(nf_flow* f, const float* packed, size_t count) {
  UseSwitches use_sw(f ? &f->sw : nullptr);
  if (!f) { set_error("flow handle is NULL"); return NF_ERR_INVALID; }
  if (!packed) { set_error("packed is NULL"); return NF_ERR_INVALID; }
  if (count != f->user_count) {
    set_error("expected %zu float32 weights for this flow, got %zu", f->user_count, count);
    return NF_ERR_INVALID;
  }
  if (!f->padded) return set_weights_internal(f, packed, count);
  std::vector<float> wide(f->weight_count, 0.f);
  for (size_t q = 0; q < count; ++q) wide[f->pad_map[q]] = packed[q];
  return set_weights_internal(f, wide.data(), wide.size());
}

static int set_weights_internal(nf_flow* f, const float* packed, size_t count) {
  const nf_flow_desc& d = f->desc;
  std::vector<int> in, out;
  linear_shapes(d, in, out);
  size_t per_layer = 0;
  for (size_t j = 0; j < in.size(); ++j) per_layer += (size_t)in[j] * out[j] + out[j];
  f->h_weights.assign(packed, packed + count);
  f->train.dirty = false;
  f->x3_stale = true;
  std::vector<float> img;
  if (f->path == PATH_AFFINE2D_H64) {
    img.resize((size_t)d.num_layers * affine2d::kLayerFloats);
    for (int l = 0; l < d.num_layers; ++l) {
      const float* W1 = packed + per_layer * l;
      const float* b1 = W1 + 2 * 64;
      const float* W2 = b1 + 64;
      const float* b2 = W2 + 64 * 64;
      const float* W3 = b2 + 64;
      const float* b3 = W3 + 64 * 2;
      const int cond_dim = (d.first_mask_parity + l) & 1;
      affine2d::pack_layer(W1, b1, W2, b2, W3, b3, cond_dim, img.data() + (size_t)l * affine2d::kLayerFloats);
    }
  } else if (f->path == PATH_RQS2D_H64) {
    pack_rqs_images(d, packed, per_layer, img);
    f->wimg_stale = false;
  } else if (f->path == PATH_SPLIT2D_SMALL) {
    // W0[1][H], b0[H], W1[H][2], b1[2] is already the kernel's order
    img.assign(packed, packed + count);
  } else {
    NF_HIP(hipStreamSynchronize(f->stream));
    NF_TRY(f->plan.build(d, packed));
    f->have_weights = true;
    if (f->train.ready) {  // a training state exists: restart it from the new weights
      NF_HIP(hipMemcpy(f->train.d_canon, packed, count * sizeof(float), hipMemcpyHostToDevice));
      NF_TRY(f->plan.prepare_backward());
      const size_t oc = f->train.plan_layout ? f->train.opt_count : count;
      NF_HIP(hipMemset(f->train.d_m, 0, oc * sizeof(float)));
      NF_HIP(hipMemset(f->train.d_v, 0, oc * sizeof(float)));
      f->train.step = 0;
      if (f->d_adam_bias) { hipFree(f->d_adam_bias); f->d_adam_bias = nullptr; }  // the device-side count restarts too
      f->train.canon_stale = false;
    }
    return NF_OK;
  }
  NF_HIP(hipStreamSynchronize(f->stream));
  if (f->d_wimg) { hipFree(f->d_wimg); f->d_wimg = nullptr; }
  NF_HIP(hipMalloc(&f->d_wimg, img.size() * sizeof(float)));
  NF_HIP(hipMemcpy(f->d_wimg, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
  f->have_weights = true;
  if (f->train.ready) {  // a training state exists: restart it from the new weights
    const size_t oc = f->train.plan_layout ? f->train.opt_count : count;
    NF_HIP(hipMemset(f->train.d_m, 0, oc * sizeof(float)));
    NF_HIP(hipMemset(f->train.d_v, 0, oc * sizeof(float)));
    f->train.step = 0;
    // the device-side count of APPLIED steps (skip_nonfinite_updates) restarts with the moments: left alone, Adam's
    // bias correction would continue from the old t on zeroed moments
    if (f->d_adam_bias) { hipFree(f->d_adam_bias); f->d_adam_bias = nullptr; }
    f->train.canon_stale = false;
  }
  if (f->train.ready && f->path == PATH_RQS2D_H64 && f->train.rqs_fused) {
    NF_HIP(hipMemcpy(f->train.d_canon, packed, count * sizeof(float), hipMemcpyHostToDevice));
    NF_TRY(launch_repack(f->train.d_canon, f->train.d_bsrc, f->train.d_bshift, f->train.d_bwd_img,
                         (size_t)d.num_layers * rqs2d_bwd::kImageFloats, f->stream));
    NF_HIP(hipStreamSynchronize(f->stream));
  } else if (f->train.ready && f->path == PATH_RQS2D_H64) {
    NF_HIP(hipMemcpy(f->train.d_canon, packed, count * sizeof(float), hipMemcpyHostToDevice));
    NF_TRY(f->plan.build(d, packed));
    NF_TRY(f->plan.prepare_backward());
  }
  if (f->train.ready && f->path == PATH_AFFINE2D_H64) {
    NF_HIP(hipMemcpy(f->train.d_canon, packed, count * sizeof(float), hipMemcpyHostToDevice));
    NF_TRY(launch_repack(f->train.d_canon, f->train.d_bsrc, f->train.d_bshift, f->train.d_bwd_img,
                         (size_t)d.num_layers * affine2d_bwd::kImageFloats, f->stream));
    f->train.bwd_x3_stale = true;
    NF_HIP(hipStreamSynchronize(f->stream));
  }
  return NF_OK;
}

int nf_flow_kernel_name(const nf_flow* f, char* buf, size_t buf_len) {
  if (!f || !buf || !buf_len) { set_error("bad argument"); return NF_ERR_INVALID; }
  snprintf(buf, buf_len, "%s", path_name(f->path));
  return NF_OK;
}

int nf_flow_set_option(nf_flow* f, const char* name, const char* value) {
  UseSwitches use_sw(f ? &f->sw : nullptr);
  if (!f || !name || !value) { set_error("bad argument"); return NF_ERR_INVALID; }
  if (strcmp(name, "hidden_matmul") == 0) {
    const bool on = strcmp(value, "bf16x3") == 0;
    if (!on && strcmp(value, "f32") != 0) {
      set_error("hidden_matmul: unknown value '%s' (f32, bf16x3)", value);
      return NF_ERR_INVALID;
    }
    if (f->path == PATH_AFFINE2D_H64 || f->path == PATH_RQS2D_H64) { f->x3 = on; return NF_OK; }
    if (f->path == PATH_GENERIC) {
      NF_HIP(hipStreamSynchronize(f->stream));
      if (on && f->train.ready && f->train.step > 0) {
        // the split images are packed on the host: after optimiser steps on the device the plan is rebuilt from
        // the trained weights (its own host copy still holds the ones it was built with)
        std::vector<float> w(f->weight_count);
        NF_TRY(get_weights_internal(f, w.data(), w.size()));
        f->plan.x3 = true;
        NF_TRY(f->plan.build(f->desc, w.data()));
        return f->plan.prepare_backward();
      }
      return f->plan.set_x3(on);
    }
    if (!on) return NF_OK;
    set_error("hidden_matmul=bf16x3 exists for the MFMA kernels (%s, %s, %s), this flow runs %s",
              path_name(PATH_AFFINE2D_H64), path_name(PATH_RQS2D_H64), path_name(PATH_GENERIC), path_name(f->path));
    return NF_ERR_UNSUPPORTED;
  }
  if (strcmp(name, "skip_nonfinite_updates") == 0) {
    if (strcmp(value, "0") != 0 && strcmp(value, "1") != 0) {
      set_error("skip_nonfinite_updates: unknown value '%s' (0, 1)", value);
      return NF_ERR_INVALID;
    }
    const bool on = value[0] == '1';
    if (on != f->skip_nonfinite && f->d_adam_bias) {
      // Leaving the mode: the count of APPLIED steps lives on the device (it does not advance on a skipped step);
      // it becomes the host's count, which is what the bias correction uses while the option is off, and the
      // device copy is dropped so that switching the option on again seeds it from the host's count afresh
      // (a stale device count would omit the steps applied in between).
      float bias[3];
      NF_HIP(hipStreamSynchronize(f->stream));
      NF_HIP(hipMemcpy(bias, f->d_adam_bias, sizeof(bias), hipMemcpyDeviceToHost));
      f->train.step = (long long)bias[2];
      NF_HIP(hipFree(f->d_adam_bias));
      f->d_adam_bias = nullptr;
    }
    f->skip_nonfinite = on;
    return NF_OK;
  }
  set_error("unknown option '%s'", name);
  return NF_ERR_INVALID;
}

int nf_flow_get_option(const nf_flow* f, const char* name, char* buf, size_t buf_len) {
  if (!f || !name || !buf || !buf_len) { set_error("bad argument"); return NF_ERR_INVALID; }
  if (strcmp(name, "hidden_matmul") == 0) {
    const bool on = f->path == PATH_GENERIC ? f->plan.x3 : f->x3;
    snprintf(buf, buf_len, "%s", on ? "bf16x3" : "f32");
    return NF_OK;
  }
  if (strcmp(name, "skip_nonfinite_updates") == 0) {
    snprintf(buf, buf_len, "%s", f->skip_nonfinite ? "1" : "0");
    return NF_OK;
  }
  set_error("unknown option '%s'", name);
  return NF_ERR_INVALID;
}

int nf_flow_synchronize(nf_flow* f) {
  UseSwitches use_sw(f ? &f->sw : nullptr);
  if (!f) { set_error("flow handle is NULL"); return NF_ERR_INVALID; }
  NF_HIP(hipStreamSynchronize(f->stream));
  if (f->reduce_pending) NF_TRY(flush_pending_reductions(f));  // collective: every rank synchronises alike
  NF_HIP(hipStreamSynchronize(f->side));
  return NF_OK;
}

int nf_chain_apply(nf_flow* f, int direction, int layer_begin, int layer_end, const float* x,
                   int64_t rows, float* y, float* logdet, int mem) {
  UseSwitches use_sw(f ? &f->sw : nullptr);
  NF_TRY(check_ready(f, x, rows));
  if (direction < NF_DIR_CHAIN_FORWARD || direction > NF_DIR_LAST_TO_FIRST_INVERSE) {
    set_error("unknown direction %d", direction); return NF_ERR_INVALID;
  }
  if (layer_begin < 0 || layer_end > f->desc.num_layers || layer_begin >= layer_end) {
    set_error("layer range [%d,%d) outside the chain's %d layers", layer_begin, layer_end, f->desc.num_layers);
    return NF_ERR_INVALID;
  }
  if (rows == 0) return NF_OK;
  const int D = f->desc.event_dim;
  ChainIO io;
  memset(&io, 0, sizeof(io));
  io.rows = rows;
  io.layer_begin = layer_begin;
  io.layer_end = layer_end;
  direction_flags(direction, &io.descending, &io.inverse_op);
  if (mem == NF_MEM_DEVICE) {
    NF_TRY(check_device_ptr(x, "x")); NF_TRY(check_device_ptr(y, "y")); NF_TRY(check_device_ptr(logdet, "logdet"));
    io.x = x; io.y = y; io.logdet_out = logdet;
    return run_chain_device(f, io, nullptr);
  }
  NF_TRY(ensure_stage(f, rows));
  // a long chain uses d_y / d_ld as its own scratch, so stage outputs there too
  NF_HIP(hipMemcpyAsync(f->d_x, x, sizeof(float) * rows * D, hipMemcpyHostToDevice, f->stream));
  io.x = f->d_x; io.y = y ? f->d_y : nullptr; io.logdet_out = logdet ? f->d_ld : nullptr;
  NF_TRY(run_chain_device(f, io, nullptr));
  if (y) NF_HIP(hipMemcpyAsync(y, f->d_y, sizeof(float) * rows * D, hipMemcpyDeviceToHost, f->stream));
  if (logdet) NF_HIP(hipMemcpyAsync(logdet, f->d_ld, sizeof(float) * rows, hipMemcpyDeviceToHost, f->stream));
  NF_HIP(hipStreamSynchronize(f->stream));
  return NF_OK;
}

}  // extern "C"
namespace nf {
int default_direction(const nf_flow* f) {
  return f->desc.kind == NF_KIND_AFFINE_SPLIT ? NF_DIR_LAST_TO_FIRST_INVERSE : NF_DIR_CHAIN_FORWARD;
}
}  // namespace nf
extern "C" {

static int enqueue_log_prob(nf_flow* f, int direction, const float* x_dev, long long rows, float* lp_dev,
                            double* block_sums, int* out_grid) {
  ChainIO io;
  memset(&io, 0, sizeof(io));
  io.rows = rows;
  io.layer_begin = 0;
  io.layer_end = f->desc.num_layers;
  direction_flags(direction, &io.descending, &io.inverse_op);
  io.x = x_dev;
  io.logprob_out = lp_dev;
  io.block_sums = block_sums;
  io.add_base = 1;
  return run_chain_device(f, io, out_grid);
}

static int ensure_pipe_events(nf_flow* f) {
  for (int q = 0; q < 4; ++q)
    if (!f->ev_pipe[q]) NF_HIP(hipEventCreateWithFlags(&f->ev_pipe[q], hipEventDisableTiming));
  return NF_OK;
}

// log_prob of a large HOST batch on the fused 2-D kernels, in chunks: chunk c's kernel runs on the handle's stream while
// chunk c+1 arrives and chunk c-1 leaves on the side stream (pageable copies block the HOST, so they are issued in the
// order that keeps a kernel enqueued behind them).  A row's bits do not depend on the batch it is evaluated in, so the
// result is the unchunked call's.  0.835 -> see DESIGN 5 (profiles/host_path.py).
// On ANY failure inside a pipelined section both streams are drained before the error is returned: copies into the
// caller's pageable buffer may still be in flight, and the caller may free that buffer as soon as the call has failed
// (ADVICE r03).
struct DrainOnError {
  nf_flow* f;
  bool armed = true;
  explicit DrainOnError(nf_flow* flow) : f(flow) {}
  ~DrainOnError() {
    if (!armed) return;
    (void)hipStreamSynchronize(f->side);
    (void)hipStreamSynchronize(f->stream);
    (void)hipGetLastError();
  }
};

static int log_prob_host_pipelined(nf_flow* f, int direction, const float* x, long long rows, float* out) {
  DrainOnError drain(f);
  NF_TRY(ensure_stage(f, rows));
  NF_TRY(ensure_pipe_events(f));
  const long long C = kPipeChunkRows;
  const long long n = (rows + C - 1) / C;
  auto rows_of = [&](long long c) { return std::min(C, rows - c * C); };
  NF_HIP(hipStreamSynchronize(f->stream));   // the staging buffers are free
  // ev_pipe[c & 1]: chunk c has arrived (side stream); ev_pipe[2 + (c & 1)]: chunk c is computed (main stream)
  auto arrive = [&](long long c) -> int {
    NF_HIP(hipMemcpyAsync(f->d_x + c * C * 2, x + c * C * 2, sizeof(float) * rows_of(c) * 2, hipMemcpyHostToDevice, f->side));
    NF_HIP(hipEventRecord(f->ev_pipe[c & 1], f->side));
    return NF_OK;
  };
  auto compute = [&](long long c) -> int {
    NF_HIP(hipStreamWaitEvent(f->stream, f->ev_pipe[c & 1], 0));
    NF_TRY(enqueue_log_prob(f, direction, f->d_x + c * C * 2, rows_of(c), f->d_lp + c * C, nullptr, nullptr));
    NF_HIP(hipEventRecord(f->ev_pipe[2 + (c & 1)], f->stream));
    return NF_OK;
  };
  NF_TRY(arrive(0));
  NF_TRY(compute(0));
  for (long long c = 0; c < n; ++c) {
    if (c + 1 < n) {   // chunk c+1 arrives while c is computed, and its kernel is enqueued BEFORE the host blocks in c's copy out
      NF_TRY(arrive(c + 1));
      NF_TRY(compute(c + 1));
    }
    NF_HIP(hipStreamWaitEvent(f->side, f->ev_pipe[2 + (c & 1)], 0));
    NF_HIP(hipMemcpyAsync(out + c * C, f->d_lp + c * C, sizeof(float) * rows_of(c), hipMemcpyDeviceToHost, f->side));
  }
  NF_HIP(hipStreamSynchronize(f->side));
  NF_HIP(hipStreamSynchronize(f->stream));
  drain.armed = false;
  return NF_OK;
}

// the handle's pinned, device-mapped staging block (false: the mapping is not to be had, callers copy instead)
static bool ensure_zero_copy(nf_flow* f) {
  if (!f->h_zc) {
    void* dev = nullptr;
    if (hipHostMalloc(&f->h_zc, sizeof(float) * kZeroCopyRows * 3, hipHostMallocMapped) == hipSuccess &&
        hipHostGetDevicePointer(&dev, f->h_zc, 0) == hipSuccess) {
      f->h_zc_dev = static_cast<float*>(dev);
    } else {
      if (f->h_zc) hipHostFree(f->h_zc);
      f->h_zc = nullptr;
      f->h_zc_dev = nullptr;
      (void)hipGetLastError();
    }
  }
  return f->h_zc_dev != nullptr;
}

int nf_log_prob(nf_flow* f, const float* x, int64_t rows, float* out_log_prob, int mem) {
  UseSwitches use_sw(f ? &f->sw : nullptr);
  if (!f) { set_error("flow handle is NULL"); return NF_ERR_INVALID; }
  return nf_chain_log_prob(f, default_direction(f), x, rows, out_log_prob, mem);
}

int nf_chain_log_prob(nf_flow* f, int direction, const float* x, int64_t rows, float* out_log_prob, int mem) {
  UseSwitches use_sw(f ? &f->sw : nullptr);
  NF_TRY(check_ready(f, x, rows));
  if (direction < NF_DIR_CHAIN_FORWARD || direction > NF_DIR_LAST_TO_FIRST_INVERSE) {
    set_error("unknown direction %d", direction); return NF_ERR_INVALID;
  }
  if (!out_log_prob && rows > 0) { set_error("out_log_prob is NULL"); return NF_ERR_INVALID; }
  if (rows == 0) return NF_OK;
  if (mem == NF_MEM_DEVICE) {
    NF_TRY(check_device_ptr(x, "x")); NF_TRY(check_device_ptr(out_log_prob, "out_log_prob"));
    return enqueue_log_prob(f, direction, x, rows, out_log_prob, nullptr, nullptr);
  }
  const int D = f->desc.event_dim;
  const bool fused2d = f->path == PATH_AFFINE2D_H64 || f->path == PATH_RQS2D_H64 || f->path == PATH_SPLIT2D_SMALL;
  if (fused2d && D == 2 && rows <= kZeroCopyRows && !f->sw.no_zero_copy) {
    // Two pageable copies cost the host ~10 us each beside a 25 us kernel (76 us per call at batch 128): the fused
    // kernels read x once and write log_prob once, so both go over the mapping of one pinned buffer instead.
    if (ensure_zero_copy(f)) {
      memcpy(f->h_zc, x, sizeof(float) * rows * 2);
      NF_TRY(enqueue_log_prob(f, direction, f->h_zc_dev, rows, f->h_zc_dev + kZeroCopyRows * 2, nullptr, nullptr));
      NF_HIP(hipStreamSynchronize(f->stream));
      memcpy(out_log_prob, f->h_zc + kZeroCopyRows * 2, sizeof(float) * rows);
      return NF_OK;
    }
  }
  if (fused2d && D == 2 && rows >= 2 * kPipeChunkRows && f->desc.num_layers <= affine2d::kMaxResidentLayers &&
      !f->sw.no_host_pipeline)
    return log_prob_host_pipelined(f, direction, x, rows, out_log_prob);
  NF_TRY(ensure_stage(f, rows));
  NF_HIP(hipMemcpyAsync(f->d_x, x, sizeof(float) * rows * D, hipMemcpyHostToDevice, f->stream));
  NF_TRY(enqueue_log_prob(f, direction, f->d_x, rows, f->d_lp, nullptr, nullptr));
  NF_HIP(hipMemcpyAsync(out_log_prob, f->d_lp, sizeof(float) * rows, hipMemcpyDeviceToHost, f->stream));
  NF_HIP(hipStreamSynchronize(f->stream));
  return NF_OK;
}

int nf_nll_enqueue(nf_flow* f, const float* x_dev, int64_t rows_local, int64_t rows_global,
                   float* out_log_prob_dev, int slot) {
  UseSwitches use_sw(f ? &f->sw : nullptr);
  NF_TRY(check_ready(f, x_dev, rows_local));
  if (slot < 0 || slot >= NF_NLL_SLOTS) { set_error("slot must be 0..%d", NF_NLL_SLOTS - 1); return NF_ERR_INVALID; }
  if (rows_global < rows_local || rows_global < 1) { set_error("rows_global must be >= rows_local and >= 1"); return NF_ERR_INVALID; }
  NF_TRY(check_device_ptr(x_dev, "x")); NF_TRY(check_device_ptr(out_log_prob_dev, "out_log_prob"));
  int grid = 0;
  double* sums = f->d_block_sums + (size_t)slot * kBlockSumCap;
  // the slot's previous tail must have read its block sums before this kernel rewrites them
  if (f->tail_pending[slot]) NF_HIP(hipStreamWaitEvent(f->stream, f->ev_tail[slot], 0));
  // Small batches (the reference evaluates 128 ... 4096 rows per call): the chain kernel takes ~15 us, less than the
  // two event records and two stream waits that move the tail to the side stream cost the host -- the step was
  // API-bound at 36 us.  There the sum kernel follows on the same stream and writes the scalar into the pinned ring:
  // two launches, nothing else.
  // (the split flow of src/real_nvp.py has no matrix work: its kernel is 8 us at 2^20 rows, so the same holds up to
  // 2^21 -- 13.2 against 15.8 us per step at 2^20; from 2^22 on the side stream wins again:
  // profiles/scratch/nll_step_split2d.py)
  long long inline_rows = f->path == PATH_SPLIT2D_SMALL ? (1LL << 21) : kInlineTailRows;
  if (f->sw.inline_tail_rows) inline_rows = f->sw.inline_tail_rows;   // NF_INLINE_TAIL_LG, development
  if (f->comm == nullptr && f->h_ring_dev != nullptr && rows_local > 0 && rows_local <= inline_rows) {
    NF_TRY(enqueue_log_prob(f, default_direction(f), x_dev, rows_local, out_log_prob_dev, sums, &grid));
    NF_TRY(launch_sum_blocks(sums, grid, f->d_sum + slot, f->stream, f->h_ring_dev + slot));
    f->tail_pending[slot] = false;
    f->ring_rows[slot] = rows_global;
    return NF_OK;
  }
  if (rows_local > 0)
    NF_TRY(enqueue_log_prob(f, default_direction(f), x_dev, rows_local, out_log_prob_dev, sums, &grid));
  NF_HIP(hipEventRecord(f->ev_kernel[slot], f->stream));
  NF_HIP(hipStreamWaitEvent(f->side, f->ev_kernel[slot], 0));
  // without a communicator the sum kernel writes the scalar into the pinned ring itself (h_ring_dev)
  const bool direct = f->comm == nullptr && f->h_ring_dev != nullptr && rows_local > 0;
  if (rows_local > 0) {
    NF_TRY(launch_sum_blocks(sums, grid, f->d_sum + slot, f->side, direct ? f->h_ring_dev + slot : nullptr));
  } else {
    NF_HIP(hipMemsetAsync(f->d_sum + slot, 0, sizeof(double), f->side));
  }
  if (f->comm) {
    // sharded: the slot's local sum waits on the device; ONE all-reduce over all waiting slots runs
    // when a result is asked for (nf_nll_result / nf_flow_synchronize), not one per enqueued batch --
    // a tiny collective kernel per step costs the chain kernel beside it ~10 % (measured)
    f->reduce_pending |= 1ull << slot;
  } else if (!direct) {
    NF_HIP(hipMemcpyAsync(f->h_ring + slot, f->d_sum + slot, sizeof(double), hipMemcpyDeviceToHost, f->side));
  }
  NF_HIP(hipEventRecord(f->ev_tail[slot], f->side));
  f->tail_pending[slot] = true;
  f->ring_rows[slot] = rows_global;
  return NF_OK;
}

int nf_nll_result(nf_flow* f, int slot, double* out_nll) {
  UseSwitches use_sw(f ? &f->sw : nullptr);
  if (!f || !out_nll) { set_error("bad argument"); return NF_ERR_INVALID; }
  if (slot < 0 || slot >= NF_NLL_SLOTS || f->ring_rows[slot] == 0) {
    set_error("slot %d holds no result", slot); return NF_ERR_STATE;
  }
  if (f->reduce_pending >> slot & 1ull) NF_TRY(flush_pending_reductions(f));
  *out_nll = -f->h_ring[slot] / (double)f->ring_rows[slot];
  return NF_OK;
}

int nf_nll(nf_flow* f, const float* x, int64_t rows_local, int64_t rows_global, float* out_log_prob,
           double* out_nll, int mem) {
  UseSwitches use_sw(f ? &f->sw : nullptr);
  NF_TRY(check_ready(f, x, rows_local));
  if (!out_nll) { set_error("out_nll is NULL"); return NF_ERR_INVALID; }
  if (mem == NF_MEM_DEVICE) {
    NF_TRY(nf_nll_enqueue(f, x, rows_local, rows_global, out_log_prob, 0));
    NF_TRY(nf_flow_synchronize(f));
    return nf_nll_result(f, 0, out_nll);
  }
  const int D = f->desc.event_dim;
  const bool fused2d = f->path == PATH_AFFINE2D_H64 || f->path == PATH_RQS2D_H64 || f->path == PATH_SPLIT2D_SMALL;
  if (fused2d && D == 2 && rows_local > 0 && rows_local <= kZeroCopyRows && !f->sw.no_zero_copy &&
      ensure_zero_copy(f)) {   // as nf_chain_log_prob: x and log_prob over the mapping of one pinned block
    memcpy(f->h_zc, x, sizeof(float) * rows_local * 2);
    float* lp = f->h_zc_dev + kZeroCopyRows * 2;
    NF_TRY(nf_nll_enqueue(f, f->h_zc_dev, rows_local, rows_global, lp, 0));
    NF_TRY(nf_flow_synchronize(f));
    if (out_log_prob) memcpy(out_log_prob, f->h_zc + kZeroCopyRows * 2, sizeof(float) * rows_local);
    return nf_nll_result(f, 0, out_nll);
  }
  if (rows_local > 0) {
    NF_TRY(ensure_stage(f, rows_local));
    NF_HIP(hipMemcpyAsync(f->d_x, x, sizeof(float) * rows_local * D, hipMemcpyHostToDevice, f->stream));
  }
  NF_TRY(nf_nll_enqueue(f, f->d_x, rows_local, rows_global, out_log_prob ? f->d_lp : nullptr, 0));
  if (out_log_prob && rows_local > 0)
    NF_HIP(hipMemcpyAsync(out_log_prob, f->d_lp, sizeof(float) * rows_local, hipMemcpyDeviceToHost, f->stream));
  NF_TRY(nf_flow_synchronize(f));
  return nf_nll_result(f, 0, out_nll);
}

int nf_base_log_prob(int base, int event_dim, const float* x, int64_t rows, float* out, int mem) {
  if (base != NF_BASE_STD_NORMAL && base != NF_BASE_UNIFORM01) { set_error("unknown base %d", base); return NF_ERR_INVALID; }
  if (event_dim < 1 || rows < 0) { set_error("bad shape"); return NF_ERR_INVALID; }
  if (rows == 0) return NF_OK;
  if (!x || !out) { set_error("x or out is NULL"); return NF_ERR_INVALID; }
  if (mem == NF_MEM_DEVICE) {
    NF_HIP(hipDeviceSynchronize());  // x may have been produced on a handle's (non-blocking) stream
    NF_TRY(generic::launch_base_log_prob(x, nullptr, out, nullptr, rows, event_dim, base, nullptr, nullptr));
    NF_HIP(hipStreamSynchronize(nullptr));  // complete on return: the next consumer may be any stream
    return NF_OK;
  }
  float *dx = nullptr, *dl = nullptr;
  NF_HIP(hipMalloc(&dx, sizeof(float) * rows * event_dim));
  hipError_t e = hipMalloc(&dl, sizeof(float) * rows);
  if (e != hipSuccess) { hipFree(dx); return hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
  int s = NF_OK;
  if ((e = hipMemcpy(dx, x, sizeof(float) * rows * event_dim, hipMemcpyHostToDevice)) != hipSuccess)
    s = hip_fail(e, "hipMemcpy", __FILE__, __LINE__);
  if (s == NF_OK) s = generic::launch_base_log_prob(dx, nullptr, dl, nullptr, rows, event_dim, base, nullptr, nullptr);
  if (s == NF_OK && (e = hipMemcpy(out, dl, sizeof(float) * rows, hipMemcpyDeviceToHost)) != hipSuccess)
    s = hip_fail(e, "hipMemcpy", __FILE__, __LINE__);
  hipFree(dx);
  hipFree(dl);
  return s;
}

int nf_elementwise_apply(const nf_elementwise_desc* desc, int inverse, const float* x, const float* params,
                         int64_t n, float* out_y, float* out_logdet, int mem) {
  if (!desc) { set_error("desc is NULL"); return NF_ERR_INVALID; }
  if (desc->abi_version != NF_ABI_VERSION) {
    set_error("desc.abi_version %d, library speaks %d", desc->abi_version, NF_ABI_VERSION); return NF_ERR_INVALID;
  }
  if (desc->kind < NF_ELEM_AFFINE_LOG_SCALE || desc->kind > NF_ELEM_RQS) {
    set_error("unknown element-wise bijector kind %d", desc->kind); return NF_ERR_INVALID;
  }
  if (n < 0) { set_error("n must be >= 0"); return NF_ERR_INVALID; }
  if (mem != NF_MEM_HOST && mem != NF_MEM_DEVICE) { set_error("unknown mem %d", mem); return NF_ERR_INVALID; }
  nf_flow_desc fd;
  memset(&fd, 0, sizeof(fd));
  int P = 2;
  if (desc->kind == NF_ELEM_RQS) {
    // the spline is the coupling kinds' own: a "flow" of event size 1 whose single dim is the transformed one
    fd.abi_version = NF_ABI_VERSION;
    fd.kind = NF_KIND_RQS_MASKED;
    fd.event_dim = 2;   // (validate_desc wants >= 2; the transform below is told D = 1)
    fd.num_layers = 1; fd.num_hidden = 1; fd.hidden[0] = 1;
    fd.activation = NF_ACT_RELU; fd.base = NF_BASE_STD_NORMAL; fd.first_mask_parity = 1;
    fd.rqs_bins = desc->rqs_bins; fd.rqs_range_min = desc->rqs_range_min; fd.rqs_range_max = desc->rqs_range_max;
    fd.rqs_min_bin_size = desc->rqs_min_bin_size; fd.rqs_min_knot_slope = desc->rqs_min_knot_slope;
    NF_TRY(validate_desc(&fd));   // the checks distrax.RationalQuadraticSpline makes in its constructor
    fd.event_dim = 1;
    P = 3 * desc->rqs_bins + 1;
  }
  if (n == 0) return NF_OK;
  if (!x || !params) { set_error("x or params is NULL"); return NF_ERR_INVALID; }
  if (!out_y && !out_logdet) return NF_OK;
  int ndev = 0;
  nf_device_count(&ndev);
  if (ndev == 0) { set_error("no HIP device visible: libnf_hip has no CPU fallback"); return NF_ERR_NO_DEVICE; }
  const Switches sw_now = Switches::from_env();
  UseSwitches use_sw(&sw_now);
  const bool host = mem == NF_MEM_HOST;
  const size_t nb = sizeof(float) * (size_t)n;
  // device buffers: x, params (staged from the host, or the caller's), y and logdet (the spline kernels update
  // y in place and ADD to logdet, so both always exist on the device here)
  float *dx = nullptr, *dp = nullptr, *dy = nullptr, *dl = nullptr;
  bool own_y = false, own_l = false;
  int s = NF_OK;
  hipError_t e = hipSuccess;
  auto fail = [&](const char* what) { s = hip_fail(e, what, __FILE__, __LINE__); };
  do {
    if (host) {
      if ((e = hipMalloc(&dx, nb)) != hipSuccess || (e = hipMalloc(&dp, nb * P)) != hipSuccess) { fail("hipMalloc"); break; }
      if ((e = hipMemcpy(dx, x, nb, hipMemcpyHostToDevice)) != hipSuccess ||
          (e = hipMemcpy(dp, params, nb * P, hipMemcpyHostToDevice)) != hipSuccess) { fail("hipMemcpy"); break; }
    } else {
      NF_HIP(hipDeviceSynchronize());  // x / params may have been produced on a handle's (non-blocking) stream
    }
    const float* cx = host ? dx : x;
    const float* cp = host ? dp : params;
    const bool rqs = desc->kind == NF_ELEM_RQS;
    if (host || (rqs && !out_y)) { if ((e = hipMalloc(&dy, nb)) != hipSuccess) { fail("hipMalloc"); break; } own_y = true; }
    else dy = out_y;
    if (host || (rqs && !out_logdet)) { if ((e = hipMalloc(&dl, nb)) != hipSuccess) { fail("hipMalloc"); break; } own_l = true; }
    else dl = out_logdet;
    if (rqs) {
      if ((e = hipMemcpyAsync(dy, cx, nb, hipMemcpyDeviceToDevice, nullptr)) != hipSuccess ||
          (e = hipMemsetAsync(dl, 0, nb, nullptr)) != hipSuccess) { fail("hipMemcpyAsync"); break; }
      s = generic::launch_transform(fd, dy, dl, cp, n, /*par=*/1, inverse ? 1 : 0, generic::make_rqs_const(fd), nullptr);
    } else {
      s = generic::launch_scalar_affine(cx, cp, dy, dl, n, inverse ? 1 : 0, desc->kind == NF_ELEM_AFFINE_SCALE, nullptr);
    }
    if (s != NF_OK) break;
    if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) { fail("hipStreamSynchronize"); break; }
    if (host) {
      if (out_y && (e = hipMemcpy(out_y, dy, nb, hipMemcpyDeviceToHost)) != hipSuccess) { fail("hipMemcpy"); break; }
      if (out_logdet && (e = hipMemcpy(out_logdet, dl, nb, hipMemcpyDeviceToHost)) != hipSuccess) { fail("hipMemcpy"); break; }
    }
  } while (0);
  hipFree(dx);
  hipFree(dp);
  if (own_y) hipFree(dy);
  if (own_l) hipFree(dl);
  return s;
}

int nf_base_sample(int base, int event_dim, uint64_t seed, int64_t rows, float* out, int mem) {
  if (base != NF_BASE_STD_NORMAL && base != NF_BASE_UNIFORM01) { set_error("unknown base %d", base); return NF_ERR_INVALID; }
  if (event_dim < 1 || rows < 0) { set_error("bad shape"); return NF_ERR_INVALID; }
  if (rows == 0) return NF_OK;
  if (!out) { set_error("out is NULL"); return NF_ERR_INVALID; }
  if (mem == NF_MEM_DEVICE) {
    NF_HIP(hipDeviceSynchronize());  // `out` may still be read by a kernel on a handle's stream
    NF_TRY(generic::launch_base_sample(out, rows, event_dim, base, seed, nullptr));
    NF_HIP(hipStreamSynchronize(nullptr));
    return NF_OK;
  }
  float* d = nullptr;
  NF_HIP(hipMalloc(&d, sizeof(float) * rows * event_dim));
  int s = generic::launch_base_sample(d, rows, event_dim, base, seed, nullptr);
  hipError_t e;
  if (s == NF_OK && (e = hipMemcpy(out, d, sizeof(float) * rows * event_dim, hipMemcpyDeviceToHost)) != hipSuccess)
    s = hip_fail(e, "hipMemcpy", __FILE__, __LINE__);
  hipFree(d);
  return s;
}

int nf_density_sample(int density, uint64_t seed, int64_t rows, float* out, int mem) {
  if (density < NF_DENSITY_BANANA || density > NF_DENSITY_BLOBS) { set_error("unknown density %d", density); return NF_ERR_INVALID; }
  if (rows < 0) { set_error("rows must be >= 0"); return NF_ERR_INVALID; }
  if (rows == 0) return NF_OK;
  if (!out) { set_error("out is NULL"); return NF_ERR_INVALID; }
  if (mem == NF_MEM_DEVICE) {
    NF_TRY(check_device_ptr(out, "out"));
    NF_HIP(hipDeviceSynchronize());
    NF_TRY(generic::launch_density_sample(out, rows, density, seed, nullptr));
    NF_HIP(hipStreamSynchronize(nullptr));
    return NF_OK;
  }
  float* d = nullptr;
  NF_HIP(hipMalloc(&d, sizeof(float) * rows * 2));
  int s = generic::launch_density_sample(d, rows, density, seed, nullptr);
  hipError_t e;
  if (s == NF_OK && (e = hipMemcpy(out, d, sizeof(float) * rows * 2, hipMemcpyDeviceToHost)) != hipSuccess)
    s = hip_fail(e, "hipMemcpy", __FILE__, __LINE__);
  hipFree(d);
  return s;
}

int nf_sample(nf_flow* f, uint64_t seed, int64_t rows, float* out_x, int mem) {
  UseSwitches use_sw(f ? &f->sw : nullptr);
  NF_TRY(check_ready(f, out_x, rows));
  if (rows == 0) return NF_OK;
  const int D = f->desc.event_dim;
  // generative direction: Chain.inverse for the masked kinds (Transformed.sample), forward_nvp
  // from layer 0 for the split kind (sample_nvp_chain)
  const int direction = f->desc.kind == NF_KIND_AFFINE_SPLIT ? NF_DIR_FIRST_TO_LAST_FORWARD : NF_DIR_CHAIN_INVERSE;
  ChainIO io;
  memset(&io, 0, sizeof(io));
  io.rows = rows;
  io.layer_begin = 0;
  io.layer_end = f->desc.num_layers;
  direction_flags(direction, &io.descending, &io.inverse_op);
  float* dst = out_x;
  if (mem != NF_MEM_DEVICE) {
    NF_TRY(ensure_stage(f, rows));
    dst = f->d_x;
  } else {
    NF_TRY(check_device_ptr(out_x, "out_x"));
  }
  NF_TRY(generic::launch_base_sample(dst, rows, D, f->desc.base, seed, f->stream));
  const bool fused2d = f->path == PATH_AFFINE2D_H64 || f->path == PATH_RQS2D_H64 || f->path == PATH_SPLIT2D_SMALL;
  if (mem != NF_MEM_DEVICE && fused2d && D == 2 && rows >= 2 * kPipeChunkRows &&
      f->desc.num_layers <= affine2d::kMaxResidentLayers && !f->sw.no_host_pipeline) {
    // Transformed.sample(10^6) with a host array out (the scripts' last step): the base draw for all rows in one
    // launch (the unchunked call's values), the generative walk chunk by chunk, a finished chunk leaving on the side
    // stream while the next one is walked
    DrainOnError drain(f);
    NF_TRY(ensure_pipe_events(f));
    const long long C = kPipeChunkRows, n = (rows + C - 1) / C;
    auto walk = [&](long long c) -> int {
      ChainIO seg = io;
      seg.rows = std::min(C, rows - c * C);
      seg.x = dst + c * C * 2;
      seg.y = dst + c * C * 2;
      NF_TRY(run_chain_device(f, seg, nullptr));
      NF_HIP(hipEventRecord(f->ev_pipe[c & 3], f->stream));
      return NF_OK;
    };
    NF_TRY(walk(0));
    for (long long c = 0; c < n; ++c) {
      if (c + 1 < n) NF_TRY(walk(c + 1));
      NF_HIP(hipStreamWaitEvent(f->side, f->ev_pipe[c & 3], 0));
      NF_HIP(hipMemcpyAsync(out_x + c * C * 2, dst + c * C * 2, sizeof(float) * std::min(C, rows - c * C) * 2,
                            hipMemcpyDeviceToHost, f->side));
    }
    NF_HIP(hipStreamSynchronize(f->side));
    NF_HIP(hipStreamSynchronize(f->stream));
    drain.armed = false;
    return NF_OK;
  }
  io.x = dst;
  io.y = dst;  // every chain kernel may run in place
  NF_TRY(run_chain_device(f, io, nullptr));
  if (mem != NF_MEM_DEVICE) {
    NF_HIP(hipMemcpyAsync(out_x, dst, sizeof(float) * rows * D, hipMemcpyDeviceToHost, f->stream));
    NF_HIP(hipStreamSynchronize(f->stream));
  }
  return NF_OK;
}

int nf_debug_read_f2(const float* x_dev, int64_t rows, float* out_dev) {
  if (!x_dev || !out_dev || rows < 1) { set_error("bad argument"); return NF_ERR_INVALID; }
  NF_TRY(generic::launch_debug_read_f2(x_dev, rows, out_dev, nullptr));
  NF_HIP(hipDeviceSynchronize());
  return NF_OK;
}

int nf_debug_gemm(const float* a_dev, int lda, const float* w_dev, const float* bias_dev, float* c_dev, int ldc,
                  int m, int n, int k, int act) {
  const Switches sw_now = Switches::from_env();  // a diagnostic without a handle: the environment as it stands
  UseSwitches use_sw(&sw_now);
  if (!a_dev || !w_dev || !bias_dev || !c_dev || m < 1 || n < 1 || k < 1 || lda < k || ldc < n || act < 0 || act > 2) {
    set_error("bad argument");
    return NF_ERR_INVALID;
  }
  NF_TRY(generic::launch_gemm(a_dev, lda, w_dev, bias_dev, c_dev, ldc, m, n, k, act, nullptr, nullptr));
  NF_HIP(hipDeviceSynchronize());
  return NF_OK;
}

}  // extern "C"

extern "C" {

int nf_timer_start(nf_flow* f) {
  if (!f) { set_error("flow handle is NULL"); return NF_ERR_INVALID; }
  NF_HIP(hipEventRecord(f->ev0, f->stream));
  return NF_OK;
}

int nf_timer_stop(nf_flow* f, float* out_ms) {
  if (!f || !out_ms) { set_error("bad argument"); return NF_ERR_INVALID; }
  NF_HIP(hipEventRecord(f->ev1, f->stream));
  NF_HIP(hipEventSynchronize(f->ev1));
  NF_HIP(hipEventElapsedTime(out_ms, f->ev0, f->ev1));
  return NF_OK;
}

int nf_flow_attach_comm(nf_flow* f, nf_comm* comm) {
  if (!f) { set_error("flow handle is NULL"); return NF_ERR_INVALID; }
  if (f->reduce_pending && f->comm) NF_TRY(flush_pending_reductions(f));  // results owed to the old communicator
  f->comm = comm;
  return NF_OK;
}

int nf_flow_collective_count(const nf_flow* f, uint64_t* out_count) {
  if (!f || !out_count) { set_error("bad argument"); return NF_ERR_INVALID; }
  *out_count = f->collectives;
  return NF_OK;
}

}  // extern "C"
