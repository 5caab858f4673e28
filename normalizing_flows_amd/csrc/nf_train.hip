// The training step behind nf_nll_grad / nf_train_step / nf_flow_get_weights: state set-up per
// kernel family, forward + backward + gradient gather (+ all-reduce), optimiser, image refresh, weight read-back.
// The reference's `update` (src/real_nvp_toy_densities.py:164-173, src/maf_toy_densities.py:168-176,
// src/maf_mnist.py:159-171) and `sgd_update` (src/real_nvp.py:295-306).  Split from nf_api.hip in round 4.
#include "nf_handle.h"

#include <string.h>

#include <algorithm>
#include <vector>

using namespace nf;

namespace nf {

// image <- canonical maps, derived from the pack functions themselves: pack a layer whose
// "weights" are their own canonical index + 1 and read back which index (and which power of
// two) landed in every image element.
// (last_out: width of the last Linear -- 2 for the affine flow, 2 (3K+1) for the spline flow; ids stay exact in float32)
template <typename PackFn>
static void derive_repack_map(PackFn pack, int image_floats, int cond_dim, int layer, std::vector<int>& src,
                              std::vector<signed char>& shift, int last_out = 2) {
  const int canon = 128 + 64 + 4096 + 64 + 64 * last_out + last_out;
  std::vector<float> ids(canon);
  for (int e = 0; e < canon; ++e) ids[e] = (float)(e + 1);
  const float* W1 = ids.data();
  const float* b1 = W1 + 128;
  const float* W2 = b1 + 64;
  const float* b2 = W2 + 4096;
  const float* W3 = b2 + 64;
  const float* b3 = W3 + 64 * last_out;
  std::vector<float> img(image_floats);
  pack(W1, b1, W2, b2, W3, b3, cond_dim, img.data());
  for (int e = 0; e < image_floats; ++e) {
    const float v = img[e];
    int sh = 0;
    float id = v;
    if (v == 0.f) { src.push_back(-1); shift.push_back(0); continue; }
    if (v > 1e12f) { sh = 64; id = ldexpf(v, -64); }
    else if (v < 1e-12f) { sh = -64; id = ldexpf(v, 64); }
    src.push_back(layer * canon + (int)(id + 0.5f) - 1);
    shift.push_back((signed char)sh);
  }
}

template <typename T>
static int upload(T** dst, const std::vector<T>& host) {
  NF_HIP(hipMalloc(dst, host.size() * sizeof(T)));
  NF_HIP(hipMemcpy(*dst, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  return NF_OK;
}

// flows whose gradient comes from the general plan: every masked kind without a fused backward
static const nf_flow_desc& d_of(const nf_flow* f) { return f->desc; }
// the fused spline flow has its own backward (nf_rqs2d_bwd.hip); NF_NO_RQS2D_BWD=1 keeps the general plan for A/B
static bool rqs_fused_backward(const nf_flow* f) {
  return f->path == PATH_RQS2D_H64 && !f->sw.no_rqs2d_bwd && rqs2d_bwd::supported_bins(f->desc.rqs_bins);
}
static bool trains_through_plan(const nf_flow* f) {
  const int k = f->desc.kind;
  return (f->path == PATH_GENERIC || (f->path == PATH_RQS2D_H64 && !rqs_fused_backward(f))) &&
         (k == NF_KIND_AFFINE_MASKED || k == NF_KIND_RQS_MASKED);
}

static int ensure_train(nf_flow* f, long long rows) {
  const bool split = f->path == PATH_SPLIT2D_SMALL && split2d::bwd_supported(f->desc.hidden[0]);
  const bool planned = trains_through_plan(f);
  const bool rqsf = rqs_fused_backward(f);
  if (f->path != PATH_AFFINE2D_H64 && !split && !planned && !rqsf) {
    set_error("the backward pass is built for the masked coupling kinds (affine and spline, any event size) and "
              "the split flow of src/real_nvp.py (D = 2, MLP 1-H-2, H = 4, 8, 16)");
    return NF_ERR_UNSUPPORTED;
  }
  nf_flow::Train& t = f->train;
  const nf_flow_desc& d = f->desc;
  const size_t n = f->weight_count;
  if (!t.ready && planned) {
    // canonical weights + Adam state on the device; gradients come from generic::Plan::backward
    // Adam's state and the weights it moves live in the plan's layout (masked rows of the first Linear and the
    // never-read parameter columns of the last one are not materialised: 52 % of the canonical vector for the MNIST
    // shape) -- no gather before and no scatter after a step, 40 % less optimiser traffic.  The canonical vector
    // (d_canon) and the canonical gradient (d_grad, for nf_nll_grad) are filled on demand.
    NF_HIP(hipMalloc(&t.d_canon, n * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_grad, n * sizeof(float)));
    NF_HIP(hipMemcpy(t.d_canon, f->h_weights.data(), n * sizeof(float), hipMemcpyHostToDevice));
    if (f->path == PATH_RQS2D_H64) NF_TRY(f->plan.build(d, f->h_weights.data()));
    NF_TRY(f->plan.prepare_backward());
    t.plan_layout = true;
    t.canon_stale = false;
    t.opt_count = f->plan.bwd.n_weights;
    NF_HIP(hipMalloc(&t.d_m, t.opt_count * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_v, t.opt_count * sizeof(float)));
    NF_HIP(hipMemset(t.d_m, 0, t.opt_count * sizeof(float)));
    NF_HIP(hipMemset(t.d_v, 0, t.opt_count * sizeof(float)));
    t.step = 0;
    if (f->d_adam_bias) { hipFree(f->d_adam_bias); f->d_adam_bias = nullptr; }
    t.ready = true;
  }
  if (!t.ready && split) {
    // the split kernels read the canonical layout directly: d_wimg IS the weight vector
    NF_HIP(hipMalloc(&t.d_grad, n * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_m, n * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_v, n * sizeof(float)));
    NF_HIP(hipMemset(t.d_m, 0, n * sizeof(float)));
    NF_HIP(hipMemset(t.d_v, 0, n * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_partials, (size_t)split2d::bwd_max_blocks() * d.num_layers * split2d::layer_floats(d.hidden[0]) * sizeof(float)));
    t.step = 0;
    if (f->d_adam_bias) { hipFree(f->d_adam_bias); f->d_adam_bias = nullptr; }
    t.ready = true;
  }
  if (!t.ready && rqsf) {
    // as the fused affine flow below: canonical weights, gradient and Adam state on the device; after an optimiser
    // step the forward and the backward image are refreshed from the canonical vector by index maps derived from the
    // pack functions themselves
    const int K = d.rqs_bins, P2 = 2 * (3 * K + 1);
    NF_HIP(hipMalloc(&t.d_canon, n * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_grad, n * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_m, n * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_v, n * sizeof(float)));
    NF_HIP(hipMemcpy(t.d_canon, f->h_weights.data(), n * sizeof(float), hipMemcpyHostToDevice));
    NF_HIP(hipMemset(t.d_m, 0, n * sizeof(float)));
    NF_HIP(hipMemset(t.d_v, 0, n * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_bwd_img, (size_t)d.num_layers * rqs2d_bwd::kImageFloats * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_partials, (size_t)rqs2d_bwd::grid_blocks() * d.num_layers * rqs2d_bwd::kPartialFloats * sizeof(float)));
    for (int c = 0; c < 2; ++c) {
      std::vector<int> map(rqs2d_bwd::canon_floats(K));
      std::vector<float> scale(rqs2d_bwd::canon_floats(K));
      rqs2d_bwd::build_gather_map(c, K, map.data(), scale.data());
      NF_TRY(upload(&t.d_gmap[c], map));
      NF_TRY(upload(&t.d_gscale[c], scale));
    }
    std::vector<int> fsrc, bsrc;
    std::vector<signed char> fshift, bshift;
    auto pack_f = [K](const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                      int cond, float* img) { rqs2d::pack_layer(W1, b1, W2, b2, W3, b3, cond, K, img); };
    auto pack_b = [K](const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                      int cond, float* img) { rqs2d_bwd::pack_layer(W1, b1, W2, b2, W3, b3, cond, K, img); };
    for (int l = 0; l < d.num_layers; ++l) {
      const int cond = (d.first_mask_parity + l) & 1;
      derive_repack_map(pack_f, rqs2d::kLayerFloats, cond, l, fsrc, fshift, P2);
      derive_repack_map(pack_b, rqs2d_bwd::kImageFloats, cond, l, bsrc, bshift, P2);
    }
    NF_TRY(upload(&t.d_fsrc, fsrc));
    NF_TRY(upload(&t.d_fshift, fshift));
    NF_TRY(upload(&t.d_bsrc, bsrc));
    NF_TRY(upload(&t.d_bshift, bshift));
    NF_TRY(launch_repack(t.d_canon, t.d_bsrc, t.d_bshift, t.d_bwd_img,
                         (size_t)d.num_layers * rqs2d_bwd::kImageFloats, f->stream));
    t.rqs_fused = true;
    t.step = 0;
    if (f->d_adam_bias) { hipFree(f->d_adam_bias); f->d_adam_bias = nullptr; }
    t.ready = true;
  }
  if (t.rqs_fused && rows > t.tape_rows) {
    NF_HIP(hipStreamSynchronize(f->stream));
    hipFree(t.d_tape);
    t.d_tape = nullptr;
    t.tape_rows = 0;
    NF_HIP(hipMalloc(&t.d_tape, sizeof(float) * 2 * (size_t)rows * d.num_layers));
    t.tape_rows = rows;
  }
  if (!t.ready) {
    NF_HIP(hipMalloc(&t.d_canon, n * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_grad, n * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_m, n * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_v, n * sizeof(float)));
    NF_HIP(hipMemcpy(t.d_canon, f->h_weights.data(), n * sizeof(float), hipMemcpyHostToDevice));
    NF_HIP(hipMemset(t.d_m, 0, n * sizeof(float)));
    NF_HIP(hipMemset(t.d_v, 0, n * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_bwd_img, (size_t)d.num_layers * affine2d_bwd::kImageFloats * sizeof(float)));
    NF_HIP(hipMalloc(&t.d_partials, (size_t)affine2d_bwd::grid_blocks() * d.num_layers * affine2d_bwd::kPartialFloats * sizeof(float)));
    for (int c = 0; c < 2; ++c) {
      std::vector<int> map(affine2d_bwd::kCanonFloats);
      std::vector<float> scale(affine2d_bwd::kCanonFloats);
      affine2d_bwd::build_gather_map(c, map.data(), scale.data());
      NF_TRY(upload(&t.d_gmap[c], map));
      NF_TRY(upload(&t.d_gscale[c], scale));
    }
    std::vector<int> fsrc, bsrc;
    std::vector<signed char> fshift, bshift;
    for (int l = 0; l < d.num_layers; ++l) {
      const int cond = (d.first_mask_parity + l) & 1;
      derive_repack_map(affine2d::pack_layer, affine2d::kLayerFloats, cond, l, fsrc, fshift);
      derive_repack_map(affine2d_bwd::pack_layer, affine2d_bwd::kImageFloats, cond, l, bsrc, bshift);
    }
    NF_TRY(upload(&t.d_fsrc, fsrc));
    NF_TRY(upload(&t.d_fshift, fshift));
    NF_TRY(upload(&t.d_bsrc, bsrc));
    NF_TRY(upload(&t.d_bshift, bshift));
    NF_TRY(launch_repack(t.d_canon, t.d_bsrc, t.d_bshift, t.d_bwd_img,
                         (size_t)d.num_layers * affine2d_bwd::kImageFloats, f->stream));
    t.step = 0;
    if (f->d_adam_bias) { hipFree(f->d_adam_bias); f->d_adam_bias = nullptr; }
    t.ready = true;
  }
  if (rows > t.state_rows) {
    hipFree(t.d_state_v); hipFree(t.d_state_g);
    t.d_state_v = t.d_state_g = nullptr;
    t.state_rows = 0;
    NF_HIP(hipMalloc(&t.d_state_v, sizeof(float) * d.event_dim * rows));
    NF_HIP(hipMalloc(&t.d_state_g, sizeof(float) * d.event_dim * rows));
    t.state_rows = rows;
  }
  return NF_OK;
}

// forward (z and the NLL sum) + backward walk + gradient gather (+ all-reduce); leaves d NLL / d w in
// train.d_grad and the NLL of this call in ring slot 0
// canon_grad: also deliver the gradient in the packed (canonical) layout (t.d_grad) -- nf_nll_grad; the training step of a
// plan-trained flow uses it where Plan::backward leaves it (plan.bwd.d_grad)
static int enqueue_nll_grad(nf_flow* f, const float* x_dev, long long rows_local, long long rows_global, bool canon_grad) {
  nf_flow::Train& t = f->train;
  const nf_flow_desc& d = f->desc;
  const int slot = 0;
  double* sums = f->d_block_sums + (size_t)slot * kBlockSumCap;
  if (f->tail_pending[slot]) NF_HIP(hipStreamWaitEvent(f->stream, f->ev_tail[slot], 0));
  int grid = 0;
  {
    ChainIO io;
    memset(&io, 0, sizeof(io));
    io.rows = rows_local;
    io.layer_begin = 0;
    io.layer_end = d.num_layers;
    direction_flags(default_direction(f), &io.descending, &io.inverse_op);
    io.x = x_dev;
    io.y = t.d_state_v;
    io.block_sums = sums;
    io.add_base = 1;
    io.base = d.base;
    // spline flows on the fused kernel train through the plan as well: its weights follow every
    // optimiser step on the device, the fused image is rebuilt on the host only when next needed
    if (trains_through_plan(f)) {
      NF_TRY(f->plan.ensure_tape(rows_local));
      f->plan.record_tape = true;
      const int s = run_chain_plan(f, io, &grid);
      f->plan.record_tape = false;
      NF_TRY(s);
    } else {
      if (t.rqs_fused) io.tape = t.d_tape;   // [L][rows_local][2]: the kernel indexes it with io.rows
      NF_TRY(run_chain_device(f, io, &grid, /*allow_x3=*/false));
    }
  }
  NF_HIP(hipEventRecord(f->ev_kernel[slot], f->stream));
  NF_HIP(hipStreamWaitEvent(f->side, f->ev_kernel[slot], 0));
  NF_TRY(launch_sum_blocks(sums, grid, f->d_sum + slot, f->side));
  if (f->comm) NF_TRY(flow_allreduce_f64(f, f->d_sum + slot, 1, f->side));
  NF_HIP(hipMemcpyAsync(f->h_ring + slot, f->d_sum + slot, sizeof(double), hipMemcpyDeviceToHost, f->side));
  NF_HIP(hipEventRecord(f->ev_tail[slot], f->side));
  f->tail_pending[slot] = true;
  f->reduce_pending &= ~(1ull << slot);
  f->ring_rows[slot] = rows_global;
  // backward: the layer log_prob applied last (layer 0) first
  if (trains_through_plan(f)) {
    NF_TRY(f->plan.backward(t.d_state_v, t.d_state_g, rows_local, rows_global, canon_grad ? t.d_grad : nullptr,
                            f->weight_count, f->stream));
    if (f->comm) {
      if (canon_grad) NF_TRY(flow_allreduce_f32(f, t.d_grad, f->weight_count, f->stream));
      else NF_TRY(flow_allreduce_f32(f, f->plan.bwd.d_grad, f->plan.bwd.n_weights, f->stream));
    }
    return NF_OK;
  }
  if (t.rqs_fused) {
    int nparts = 0;
    NF_TRY(rqs2d_bwd::launch_chain(t.d_bwd_img, d, t.d_tape, t.d_state_v, t.d_state_g, t.d_partials, rows_local, f->stream,
                                   &nparts));
    NF_TRY(rqs2d_bwd::launch_gather(t.d_partials, nparts, d, t.d_gmap[0], t.d_gmap[1], t.d_gscale[0], t.d_gscale[1],
                                    t.d_grad, rows_global, f->stream));
    if (f->comm) NF_TRY(flow_allreduce_f32(f, t.d_grad, f->weight_count, f->stream));
    return NF_OK;
  }
  if (f->path == PATH_SPLIT2D_SMALL) {
    NF_TRY(split2d::launch_bwd_chain(f->d_wimg, d.hidden[0], d.num_layers, d.activation, d.first_mask_parity,
                                     t.d_state_v, t.d_state_g, t.d_partials, t.d_grad, rows_local, rows_global,
                                     f->stream));
    if (f->comm) NF_TRY(flow_allreduce_f32(f, t.d_grad, f->weight_count, f->stream));
    return NF_OK;
  }
  const bool bwd_x3 = f->x3;  // the handle's matmul mode covers the backward products that feed other products
  if (bwd_x3) {
    const size_t per = (size_t)affine2d_bwd::x3_image_floats();
    if (!t.d_bwd_img_x3) NF_HIP(hipMalloc(&t.d_bwd_img_x3, per * d.num_layers * sizeof(float)));
    if (t.bwd_x3_stale) {
      NF_TRY(affine2d_bwd::launch_build_x3_images(t.d_bwd_img, t.d_bwd_img_x3, d.num_layers, f->stream));
      t.bwd_x3_stale = false;
    }
  }
  {
    int nparts = 0;
    const float* images = bwd_x3 ? t.d_bwd_img_x3 : t.d_bwd_img;
    NF_TRY(affine2d_bwd::launch_chain(images, d.num_layers, d.first_mask_parity, t.d_state_v, t.d_state_g, t.d_partials,
                                      rows_local, f->stream, &nparts, bwd_x3));
    NF_TRY(affine2d_bwd::launch_gather(t.d_partials, nparts, d.num_layers, d.first_mask_parity, t.d_gmap[0], t.d_gmap[1],
                                       t.d_gscale[0], t.d_gscale[1], t.d_grad, rows_global, f->stream));
  }
  if (f->comm) NF_TRY(flow_allreduce_f32(f, t.d_grad, f->weight_count, f->stream));
  return NF_OK;
}

static int stage_train_input(nf_flow* f, const float* x, long long rows, int mem, const float** x_dev) {
  if (mem == NF_MEM_DEVICE) {
    NF_TRY(check_device_ptr(x, "x"));
    *x_dev = x;
    return NF_OK;
  }
  NF_TRY(ensure_stage(f, rows));
  NF_HIP(hipMemcpyAsync(f->d_x, x, sizeof(float) * rows * f->desc.event_dim, hipMemcpyHostToDevice, f->stream));
  *x_dev = f->d_x;
  return NF_OK;
}

}  // namespace nf

extern "C" {

int nf_nll_grad(nf_flow* f, const float* x, int64_t rows_local, int64_t rows_global, double* out_nll,
                float* out_grad, int mem) {
  UseSwitches use_sw(f ? &f->sw : nullptr);
  NF_TRY(check_ready(f, x, rows_local));
  if (rows_local < 1 || rows_global < rows_local) { set_error("need 1 <= rows_local <= rows_global"); return NF_ERR_INVALID; }
  if (!out_grad) { set_error("out_grad is NULL"); return NF_ERR_INVALID; }
  NF_TRY(ensure_train(f, rows_local));
  const float* x_dev = nullptr;
  NF_TRY(stage_train_input(f, x, rows_local, mem, &x_dev));
  NF_TRY(enqueue_nll_grad(f, x_dev, rows_local, rows_global, true));
  const float* grad = f->train.d_grad;
  if (f->padded) {  // the caller's layout: its entries of the padded gradient
    if (!f->d_pad_map) {
      NF_HIP(hipMalloc(&f->d_pad_map, f->user_count * sizeof(int)));
      NF_HIP(hipMalloc(&f->d_pad_tmp, f->user_count * sizeof(float)));
      NF_HIP(hipMemcpy(f->d_pad_map, f->pad_map.data(), f->user_count * sizeof(int), hipMemcpyHostToDevice));
    }
    NF_TRY(launch_gather(f->d_pad_tmp, f->train.d_grad, f->d_pad_map, f->user_count, f->stream));
    grad = f->d_pad_tmp;
  }
  NF_HIP(hipMemcpyAsync(out_grad, grad, f->user_count * sizeof(float),
                        mem == NF_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, f->stream));
  NF_TRY(nf_flow_synchronize(f));
  if (out_nll) NF_TRY(nf_nll_result(f, 0, out_nll));
  return NF_OK;
}

int nf_train_step(nf_flow* f, const float* x, int64_t rows_local, int64_t rows_global, const nf_optimizer* opt,
                  double* out_nll, int mem) {
  UseSwitches use_sw(f ? &f->sw : nullptr);
  NF_TRY(check_ready(f, x, rows_local));
  if (!opt) { set_error("opt is NULL"); return NF_ERR_INVALID; }
  if (opt->kind != NF_OPT_SGD && opt->kind != NF_OPT_ADAM) { set_error("unknown optimizer %d", opt->kind); return NF_ERR_INVALID; }
  if (rows_local < 1 || rows_global < rows_local) { set_error("need 1 <= rows_local <= rows_global"); return NF_ERR_INVALID; }
  NF_TRY(ensure_train(f, rows_local));
  nf_flow::Train& t = f->train;
  const float* x_dev = nullptr;
  NF_TRY(stage_train_input(f, x, rows_local, mem, &x_dev));
  NF_TRY(enqueue_nll_grad(f, x_dev, rows_local, rows_global, false));
  t.step += 1;
  float* canon = f->path == PATH_SPLIT2D_SMALL ? f->d_wimg : t.d_canon;
  float* grad = t.d_grad;
  size_t count = f->weight_count;
  const unsigned* live = nullptr;
  if (t.plan_layout) { canon = f->plan.d_weights; grad = f->plan.bwd.d_grad; count = t.opt_count; live = f->plan.bwd.d_live_bits; }
  const int* skip = nullptr;
  if (f->skip_nonfinite) {
    if (!f->d_skip_flag) NF_HIP(hipMalloc(&f->d_skip_flag, sizeof(int)));
    if (!f->d_adam_bias) {
      // steps taken so far (all of them applied: the option was off) seed the device-side count
      const float seed[3] = {0.f, 0.f, (float)(t.step - 1)};
      NF_HIP(hipMalloc(&f->d_adam_bias, sizeof(seed)));
      NF_HIP(hipMemcpyAsync(f->d_adam_bias, seed, sizeof(seed), hipMemcpyHostToDevice, f->stream));
      NF_HIP(hipStreamSynchronize(f->stream));
    }
    NF_TRY(launch_nonfinite_flag(grad, count, f->d_skip_flag, f->stream));
    skip = f->d_skip_flag;
  }
  if (opt->kind == NF_OPT_ADAM)
    NF_TRY(launch_adam(canon, grad, t.d_m, t.d_v, count, opt->learning_rate, opt->beta1, opt->beta2,
                       opt->eps, t.step, f->stream, skip, f->d_adam_bias, live));
  else
    NF_TRY(launch_sgd(canon, grad, count, opt->learning_rate, f->stream, skip, live));
  if (f->path == PATH_AFFINE2D_H64) {
    NF_TRY(launch_repack2(t.d_canon, t.d_fsrc, t.d_fshift, f->d_wimg, (size_t)f->desc.num_layers * affine2d::kLayerFloats,
                          t.d_bsrc, t.d_bshift, t.d_bwd_img, (size_t)f->desc.num_layers * affine2d_bwd::kImageFloats,
                          f->stream));
    t.bwd_x3_stale = true;
  }
  if (t.rqs_fused)
    NF_TRY(launch_repack2(t.d_canon, t.d_fsrc, t.d_fshift, f->d_wimg, (size_t)f->desc.num_layers * rqs2d::kLayerFloats,
                          t.d_bsrc, t.d_bshift, t.d_bwd_img, (size_t)f->desc.num_layers * rqs2d_bwd::kImageFloats,
                          f->stream));
  if (trains_through_plan(f)) {
    NF_TRY(f->plan.refresh_after_step(f->stream));
    t.canon_stale = true;
    if (f->path == PATH_RQS2D_H64) f->wimg_stale = true;
    if (f->plan.x3) {  // the split-bf16 weight images are built on the host
      std::vector<float> w(f->weight_count);
      t.dirty = true;
      NF_TRY(get_weights_internal(f, w.data(), w.size()));
      NF_TRY(f->plan.build(d_of(f), w.data()));
      NF_TRY(f->plan.prepare_backward());
    }
  }
  t.dirty = true;
  f->x3_stale = true;
  if (out_nll) {
    NF_TRY(nf_flow_synchronize(f));
    NF_TRY(nf_nll_result(f, 0, out_nll));
  }
  return NF_OK;
}

int nf_flow_get_weights(nf_flow* f, float* packed, size_t count) {
  UseSwitches use_sw(f ? &f->sw : nullptr);
  if (!f || !packed) { set_error("bad argument"); return NF_ERR_INVALID; }
  if (!f->have_weights) { set_error("nf_flow_set_weights has not been called"); return NF_ERR_STATE; }
  if (count != f->user_count) { set_error("expected room for %zu weights, got %zu", f->user_count, count); return NF_ERR_INVALID; }
  if (!f->padded) return get_weights_internal(f, packed, count);
  std::vector<float> wide(f->weight_count);
  NF_TRY(get_weights_internal(f, wide.data(), wide.size()));
  for (size_t q = 0; q < count; ++q) packed[q] = wide[f->pad_map[q]];
  return NF_OK;
}

}  // extern "C"

namespace nf {

int get_weights_internal(nf_flow* f, float* packed, size_t count) {
  if (f->train.ready && f->train.dirty) {
    if (f->train.plan_layout && f->train.canon_stale) {
      NF_TRY(f->plan.scatter_weights_to_canonical(f->train.d_canon, f->stream));
      f->train.canon_stale = false;
    }
    NF_TRY(nf_flow_synchronize(f));
    NF_HIP(hipMemcpy(f->h_weights.data(), f->path == PATH_SPLIT2D_SMALL ? f->d_wimg : f->train.d_canon,
                     count * sizeof(float), hipMemcpyDeviceToHost));
    f->train.dirty = false;
  }
  memcpy(packed, f->h_weights.data(), count * sizeof(float));
  return NF_OK;
}

}  // namespace nf
