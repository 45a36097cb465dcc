/* locate_hip.h - C ABI of the MI355X (gfx950) kernels for the LocAtE generator/discriminator training step.
 *
 * The reference (ClashLuke/LocAtE) has no FFI: its operator boundary is the Python autograd.Function /
 * nn.Module protocol.  Each entry point below replaces the arithmetic of one
 * reference operator (cited per function as file:line in the reference tree) and is what a ctypes stub in
 * the reference would bind.
 *
 * Conventions
 *   - all data pointers are DEVICE pointers to fp32, NCHW, contiguous unless a batch stride is given;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); calls are asynchronous;
 *   - every function returns 0 on success, non-zero on error (1 = bad argument, 2 = launch failure) and
 *     never throws; locate_last_error() returns a thread-local message for the last failure;
 *   - the library never allocates: outputs and workspaces are owned by the caller; the *_workspace_bytes
 *     helpers give the required sizes; workspaces may be reused by the next call on the same stream;
 *   - entry points are stateless and re-entrant (forward on the caller's thread, backward on the autograd
 *     worker thread, each with its own stream).
 */
#ifndef LOCATE_HIP_H
#define LOCATE_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* locate_last_error(void);
int locate_abi_version(void);
/* architecture name ("gfx950..."), compute units and wavefront size of the current device */
int locate_device_info(char* arch_name, int name_len, int* cu_count, int* wave_size);

/* ---- RootTanh: y = (x^2+1)^(1/4) tanh x and its hand-written derivative (libs/activation.py:7-36) ---- */
int locate_roottanh_fwd(const float* x, float* y, int64_t n, void* absmax /* nullable, see locate_roottanh_bwd */, void* stream);
/* accumulate: gx +=.  absmax (nullable): locate_absmax_words() (= 1024: 32 words in use, one per 128-byte line) device words,
 * ZERO before the call; the kernel folds the largest |gx| into them (atomic max on the bit pattern - order-independent - spread
 * over cache lines so that a few thousand blocks do not queue on one; the tensor's maximum is the maximum of the words) for a
 * contraction that consumes gx in its fp16-piece form (precision 2 below) */
int locate_absmax_words(void);
int locate_roottanh_bwd(const float* x, const float* gy, float* gx, int64_t n, int accumulate, void* absmax, void* stream);
/* the same word for any tensor (tests, tools, weights packed outside an optimizer step) */
int locate_absmax(const float* x, int64_t n, void* absmax, void* stream);
/* generator output tanh (libs/models.py:66); backward takes the forward OUTPUT y */
/* style chain link (libs/block.py:119-125): out[r, :z] = latent[r, :], out[r, z:] = RootTanh(pre[r, :]) in one launch, and
 * its backward on the gradient's column slice in place (row stride in elements) */
int locate_act_cat_rows_fwd(const float* latent, const float* pre, float* out, int rows, int z, int w, void* stream);
/* g_add (nullable, [rows, w] contiguous): added to the result - the gradient pre receives as a norm's style scale */
int locate_act_rows_bwd(const float* pre, const float* g, int64_t g_row_stride, const float* g_add, float* gpre, int rows, int w,
                        void* stream);
/* out = (a + b) + c, each operand contiguous inside a batch element with its own batch stride (elements): the one-pass sum of
 * the three gradients that meet at a discriminator block's input (libs/block.py:38-52, libs/scale.py:28-34: the conv branch's
 * norm, the identity half of the concatenation, the skip branch's 1x1 conv) - the reference leaves two adds to autograd */
int locate_add3(const float* a, int64_t a_bs, const float* b, int64_t b_bs, const float* c, int64_t c_bs, float* out, int batch,
                int64_t per, void* stream);
int locate_tanh_fwd(const float* x, float* y, int64_t n, void* stream);
int locate_tanh_bwd(const float* y, const float* gy, float* gx, int64_t n, void* stream);

/* ---- InPlaceNorm: global scalar mean / unbiased std, per-channel or per-sample scale, per-channel bias
 *      (libs/inplace_norm.py:4-45).  stats = {mean, std}. ---- */
size_t locate_norm_stats_workspace_bytes(void);
int locate_norm_stats(const float* x, int64_t n, float* stats, void* workspace, void* stream);
/* fused forward in two launches (statistics + apply): out = (x - mean) * scale / std + bias; scale is [C]
 * (scale_per_sample = 0) or [B*C]; with_act = 1 stores RootTanh(out) INSTEAD of out (the conv stage that follows starts
 * with RootTanh, libs/conv.py:22-24).  groups > 1: the batch stacks `groups` independent forward calls of B/groups
 * samples each (the three discriminator passes of main.py:149-152), every one with its OWN mean/std.
 * stats_out = groups x {mean, std}; workspace: locate_norm_stats_workspace_bytes() */
/* pre_partial (nullable): statistics partials of x already produced by the kernel that wrote x (locate_gate_fwd_stats with
 * the same group count); the statistics pass over x is then skipped */
int locate_norm_fwd(const float* x, const float* scale, int scale_per_sample, const float* bias, float* out, int with_act,
                    float* stats_out, int B, int C, int hw, int groups, void* workspace, const double* pre_partial,
                    void* absmax /* nullable: largest |out|, see locate_roottanh_bwd */, void* stream);
size_t locate_norm_bwd_workspace_bytes(int B, int C);
/* full backward incl. the path through std (libs/inplace_norm.py:17-27 + ATen std backward); with_act = 1: g is the
 * gradient w.r.t. RootTanh(out) and RootTanh' (libs/activation.py:22-36) is applied from the recomputed out;
 * dscale / dbias sum over all groups (the reference accumulates the three passes' gradients, main.py:153-157) */
/* accumulate_dx / accumulate (locate_norm_bwd, locate_gate_bwd, locate_feature_pool_bwd): add into the output instead of
 * overwriting it - the second consumer of a forked tensor completes the gradient the first one started (ops.fork) */
int locate_norm_bwd(const float* x, const float* g, const float* stats, const float* scale, int scale_per_sample,
                    const float* bias, int with_act, float* dx, float* dscale, float* dbias, int B, int C, int hw, int groups,
                    void* workspace, int accumulate_dx, void* stream);
/* out[c] = sum over batch and space of g[b, c, :] (bias gradients: libs/scale.py:28-34, libs/linear.py:10) */
/* the same backward in TWO launches (no middle launch in the pass's dependent chain): dx and - with scale_per_sample - the
 * per-sample scale gradient; the per-channel dscale[c] / dbias[c] (parameter gradients) come from locate_fin_norm_channels at
 * the end of the pass, out of the plane sums S1, S2 this call leaves in its workspace at locate_norm_bwd_fused_plane_offset()
 * (S1 [B*C] floats, then S2): records p = {S1, S2, stats, dscale [C] | 0, dbias [C] | 0}, i = {B, C, groups} */
size_t locate_norm_bwd_fused_workspace_bytes(int B, int C);
size_t locate_norm_bwd_fused_plane_offset(void);
int locate_norm_bwd_fused(const float* x, const float* g, const float* stats, const float* scale, int scale_per_sample,
                          const float* bias, int with_act, float* dx, float* dscale_sample, int B, int C, int hw, int groups,
                          void* workspace, int accumulate_dx, void* stream);
int locate_fin_norm_channels(const void* records, int n, void* stream);
size_t locate_channel_sum_workspace_bytes(int B, int C, int hw);
int locate_channel_sum(const float* g, float* out, int B, int C, int hw, int64_t batch_stride, void* workspace, void* stream);

/* ---- residual gate out = (gamma a + 1) x (libs/merge.py:19-62), incl. dgamma = sum x^2 g as coded.
 *      a_per_plane = 1: `a` holds one value per (batch, channel) plane (stride-0 expand, libs/util_modules.py:6-12) */
int locate_gate_fwd(const float* x, const float* a, int a_per_plane, const float* gamma, float* out, int64_t planes, int hw,
                    void* stream);
/* the same gate plus, from its epilogue, the InPlaceNorm statistics partials (sum, sum of squares per block, fp64) of `out`
 * for `groups` calls stacked along the batch: stats_partial = locate_norm_stats_workspace_bytes() bytes, handed to
 * locate_norm_fwd as pre_partial (libs/block.py:44-52: a gate's output is what the next norm normalises) */
int locate_gate_fwd_stats(const float* x, const float* a, int a_per_plane, const float* gamma, float* out, int64_t planes,
                          int hw, int groups, double* stats_partial, void* stream);
size_t locate_gate_bwd_workspace_bytes(int64_t planes);
/* dgamma nullable: the kernel then only leaves its locate_gate_bwd_partials(planes, hw) block sums (doubles, at the start of
 * `workspace`; dgamma = their sum in index order) for the caller to add up - see locate_fin_sums - or to ignore */
int locate_gate_bwd_partials(int64_t planes, int hw);
int locate_gate_bwd(const float* x, const float* a, int a_per_plane, const float* gamma, const float* g, float* dx, float* da,
                    float* dgamma, int64_t planes, int hw, void* workspace, int accumulate_dx,
                    void* da_absmax /* nullable: largest |da| (full-map form), see locate_roottanh_bwd */, void* stream);

/* ---- softmax over the last dimension of [rows, n] (libs/attention.py:35,47) ---- */
int locate_softmax_fwd(const float* x, float* y, int64_t rows, int n, void* stream);
int locate_softmax_bwd(const float* y, const float* gy, float* gx, int64_t rows, int n, void* stream);

/* ---- skip-branch resampling / indexing (libs/scale.py:7-45, libs/merge.py:4-16) ---- */
int locate_upsample2x_fwd(const float* x, float* y, int64_t planes, int H, int W, void* stream);   /* bilinear, align_corners=False */
int locate_upsample2x_bwd(const float* gy, float* gx, int64_t planes, int H, int W, void* stream);
/* FeaturePooling(r = 2) followed by the x2 upsample - the generator's skip branch (libs/scale.py:7-16,37-38) - in ONE launch:
 * x is the RAW tensor of 2 * planes * H * W elements (FeaturePooling averages ADJACENT FLAT elements), y: [planes, 2H, 2W];
 * bit for bit locate_feature_pool_fwd(r = 2) + locate_upsample2x_fwd.  _bwd: the adjoint, gx = the raw gradient (accumulate != 0:
 * added to what gx holds - the second consumer of a forked tensor, as in locate_feature_pool_bwd). */
int locate_pool2_upsample2x_fwd(const float* x, float* y, int64_t planes, int H, int W, void* stream);
int locate_pool2_upsample2x_bwd(const float* gy, float* gx, int64_t planes, int H, int W, int accumulate, void* stream); /* H, W: forward input size */
int locate_avgpool2_fwd(const float* x, float* y, int64_t planes, int H, int W, void* stream);
int locate_avgpool2_bwd(const float* gy, float* gx, int64_t planes, int H, int W, int accumulate, void* stream);   /* H, W: forward input size */
int locate_feature_pool_fwd(const float* x, float* y, int64_t n_out, int r, void* stream);         /* mean of r adjacent flat elements */
int locate_feature_pool_bwd(const float* gy, float* gx, int64_t n_out, int r, int accumulate, void* stream);
/* dst[b, c, :] (+)= src[b, c, :], c < C, independent batch strides (torch.cat along channels and its backward) */
int locate_copy_channels(const float* src, float* dst, int B, int C, int hw, int64_t src_batch_stride,
                         int64_t dst_batch_stride, int accumulate, void* stream);

/* ---- spectral norm state machine (libs/spectral_norm.py:21-32): one power iteration, u/v updated in place,
 *      sigma = {sigma, 1/sigma}, wv = W v (kept for du).  W_bar viewed as [h, wd], h = shape[0]. ---- */
size_t locate_sn_workspace_bytes(int h, int wd);
int locate_sn_power_iter(const float* w, float* u, float* v, float* sigma, float* wv, int h, int wd, void* workspace,
                         void* stream);
/* all layers of a network in four launches; `table`: device array of records {w,u,v,sigma,wv,t,s,tpart pointers;
 * int h, wd, nchunk, pad} (locate_sn_table_record_bytes() each), scratch t[wd], s[h], tpart[ceil(h/64)*wd] */
size_t locate_sn_table_record_bytes(void);
int locate_sn_power_iter_batched(const void* table, int n_layers, int max_h, int max_wd, void* stream);
/* after locate_conv_wgrad(.., w_ref = W_bar, inv_scale = 1/sigma, inner_partial): gw (in/out) enters as G/sigma and
 * leaves as dW_bar = G/sigma + dsigma u v^T with dsigma = -<G, W_bar>/sigma^2; du = dsigma (W v) (nullable);
 * dsigma_out[0] = dsigma (nullable).  u, v are the CURRENT state (the reference's autograd sees the latest .data). */
int locate_sn_weight_bwd(const double* inner_partial, int n_partial, const float* u, const float* v, const float* sigma,
                         const float* wv, float* gw, float* du, float* dsigma_out, int h, int wd, void* stream);
/* the same for `groups` (<= 4) forward calls stacked along the batch of ONE layer call, each with its own sigma_k, v_k
 * (the reference runs them one after the other, main.py:149-152): gw enters as sum_k G_k/sigma_k (locate_conv_wgrad
 * with group scaling); dsigma_k = -<gy_k, y_k - bias>/sigma_k  (= -<G_k, W_bar>/sigma_k^2 taken on the activation side);
 * gw += (sum_k dsigma_k) u v^T; du = sum_k dsigma_k W v_k; dsigma_total_out = sum_k dsigma_k.
 * sigma_tab: {sigma_k, 1/sigma_k} pairs sigma_stride floats apart; wv rows wv_stride floats apart. */
size_t locate_sn_group_workspace_bytes(void);
int locate_sn_weight_bwd_grouped(const float* gy, int64_t gy_bs, const float* y, int64_t y_bs, const float* bias, int groups,
                                 int Bg, int M, int plane, const float* sigma_tab, int sigma_stride, const float* u,
                                 const float* v, const float* wv, int64_t wv_stride, float* gw, float* du,
                                 float* dsigma_total_out, int h, int wd, void* workspace, void* stream);
/* dv = (sum of the layer's 4 dsigma slots) * W^T u for all layers of `table` (records as for
 * locate_sn_power_iter_batched; field v = dv output, field sigma = the 4 slots, cleared afterwards) */
int locate_sn_dv_batched(const void* table, int n_layers, int max_h, int max_wd, void* stream);

/* ---- dense contractions as implicit GEMMs on the fp32 MFMA (libs/conv.py:14-20, libs/attention.py:18-46,
 *      libs/scale.py:25-34, libs/linear.py:10).  geom = {B, C, H, W, M, KH, KW, stride, pad_h, pad_w, OH, OW}
 *      describes the REGULAR convolution R: out[b,m,oh,ow] = sum w[m,c,kh,kw] in[b,c,oh*s-ph+kh,ow*s-pw+kw].
 *      The weight operand is a pre-packed K-major panel (locate_conv_pack_panel; redo only when W changes);
 *      `scale` (nullable device scalar) multiplies the contraction in the epilogue (1/sigma of spectral norm,
 *      libs/spectral_norm.py:31-32); *_bs are batch strides in elements; workspace = split-K slabs (may be 0). ---- */
size_t locate_conv_panel_bytes(const int* geom, int adjoint);
int locate_conv_pack_panel(const int* geom, int adjoint, const float* w, float* panel, void* stream);
/* all panels of a network in one launch (after an optimizer step): one HOST record of locate_conv_pack_job_bytes()
 * per panel, filled by locate_conv_pack_job (block_start = running sum of *blocks_out), uploaded by the caller */
size_t locate_conv_pack_job_bytes(void);
/* direct != 0: RE-packing of a panel that has been packed in full before, in one pass (piece planes straight from the weights,
 * fp32 rows only where a kernel reads them).  fp16-piece panels need weight_absmax for it: locate_absmax_words() device words
 * with the largest magnitude of w as it is now (locate_nadam_step leaves them per tensor); else the two-pass form is taken */
int locate_conv_pack_job(const int* geom, int adjoint, const float* w, float* panel, int block_start, void* job_out,
                         int* blocks_out, int direct, const void* weight_absmax);
/* any_f16: some job is a non-direct fp16-piece panel (its absmax header is cleared first); passes: bit 0 = some gather-kernel
 * panel is in the two-pass form (split launch), bit 1 = some WINDOW panel of fp16 pieces came without absmax words (absmax
 * pre-pass into the panel headers); both 0 when every job took the direct form (locate_conv_pack_job_is_direct): one launch */
int locate_conv_pack_panels(const void* jobs, int n_jobs, int total_blocks, int any_f16, int passes, void* stream);
int locate_conv_pack_job_is_direct(const void* job);
int locate_conv_pack_job_is_window(const void* job);
/* ---- window form of the dense contractions (csrc/convwin.hip): the gathered operand is staged in LDS once per block and 8
 *      reduction channels as the raw window of the map its 128 output columns see, every tap's MFMA fragment is read from that
 *      one image (libs/conv.py:14-20's k x k convs, both directions; the 1x1 contractions of libs/scale.py:25-34,
 *      libs/attention.py:44-46).  locate_conv_win_ok(geom, adjoint | fmt, x_bs, x): 1 when this geometry / direction / operand
 *      alignment has the form.  The caller then uses panel format bit 2 (`adjoint | fmt | 4`) in locate_conv_panel_bytes /
 *      _pack_panel / _pack_job, passes `precision | 16` to locate_conv_fwd / locate_conv_dgrad and sizes the split-K workspace
 *      with locate_conv_win_workspace_bytes.  Same arithmetic and accuracy as the gather kernels at each precision. ---- */
int locate_conv_win_ok(const int* geom, int adjoint_fmt, int64_t x_bs, const void* x);
size_t locate_conv_win_workspace_bytes(const int* geom, int adjoint_fmt);
size_t locate_conv_fwd_workspace_bytes(const int* geom);
/* arrival counters for split-K launches that combine their partial tiles INSIDE the launch (the tile's last-arriving block
 * sums them in a fixed order: bit-reproducible): locate_conv_counter_bytes() bytes of device memory, zero before their first
 * use, left zero by every completed call, never shared by launches that may run concurrently (one block per layer and
 * direction is what the Python layer keeps).  `counters` may be NULL: a second kernel then sums the partial tiles. */
size_t locate_conv_counter_bytes(void);
/* scale_group_batch = 0: `scale` is one scalar; > 0: batch element b uses scale[(b / scale_group_batch) * scale_stride].
 * precision (locate_conv_fwd / _dgrad / _wgrad): 0 = fp32-faithful products (the reference's arithmetic: both operands split
 * exactly into three bf16 pieces, six bf16 MFMAs per slice, fp32 accumulation); 1 = bf16 operands (both operands rounded to
 * nearest-even bf16, one MFMA per slice, fp32 accumulation and fp32 storage) - the mixed-precision variant BASELINE.json
 * configs[1] names; tolerance against the fp32 path stated in tests/test_gpu_bf16.py. */
int locate_conv_fwd(const int* geom, const float* x, int64_t x_bs, const float* panel, const float* scale,
                    int scale_group_batch, int scale_stride, const float* bias, float* y, int64_t y_bs, void* workspace,
                    void* counters, int precision, const void* x_absmax, const void* act_epilogue, void* stream);
/* act_epilogue (nullable): HOST pointer to
 *   struct { void* act_out; int64_t act_bs; const void* lat; int64_t lat_bs; int32_t lat_z, pad;
 *            const void* mul_pre; int64_t mul_bs; void* out_absmax; }
 * act_out: the launch also writes act_out[b, m, pixel] = RootTanh(y[b, m, pixel]) (batch stride act_bs elements) - the
 *   activation between the two convs of a stage (libs/conv.py:19-20, libs/attention.py:44-46, libs/linear.py:12-13) without a
 *   launch and a read of its own; on 1x1 maps, with lat, it copies lat[n, 0 .. lat_z) in front of row n
 *   (act_out[n * act_bs - lat_z ..]): the next style link's input cat([latent, activated]) (libs/block.py:119-125).
 * mul_pre (locate_conv_dgrad; excludes act_out): the contraction's result is multiplied by RootTanh'(mul_pre[b, c, pixel])
 *   (batch stride mul_bs) before it is stored - the input gradient of a conv whose input was RootTanh(mul_pre) leaves the launch
 *   as the gradient of mul_pre itself (the separate locate_roottanh_bwd launch and its read of gx disappear).
 * out_absmax (nullable): locate_absmax_words() zeroed words that receive the largest magnitude of act_out (of gx with mul_pre). */
/* precision: 0 = fp32-faithful with three bf16 pieces per operand (six bf16 MFMAs per 32x32x16 slice); 1 = operands rounded
 * to bf16 (one MFMA); 2 = fp32-faithful with TWO fp16 pieces per operand (three fp16 MFMAs): both operands are scaled by a
 * power of two into fp16's range - the weights when the panel is packed (panel format bit: `adjoint | 2` in
 * locate_conv_panel_bytes / locate_conv_pack_panel / locate_conv_pack_job), the gathered tensor inside the kernel from
 * *x_absmax, the bit pattern of its largest magnitude (locate_absmax, or the absmax outputs of the producing kernels) - and
 * the exact inverse factors are applied to the accumulators.  Same fp32-level accuracy as precision 0 (tools/bench_conv.py
 * --check), half the matrix instructions. */
/* data adjoint of R (= ConvTranspose2d forward with weight [C_in = M, C_out = C, KH, KW]); panel: adjoint = 1 */
size_t locate_conv_dgrad_workspace_bytes(const int* geom);
int locate_conv_dgrad(const int* geom, const float* gy, int64_t gy_bs, const float* panel, const float* scale,
                      int scale_group_batch, int scale_stride, const float* bias, float* gx, int64_t gx_bs, void* workspace,
                      void* counters, int precision, const void* gy_absmax, const void* act_epilogue, void* stream);
/* gw[m,c,kh,kw] = inv_scale * sum_{b,oh,ow} gy[b,m,oh,ow] x[b,c,oh*s-ph+kh,ow*s-pw+kw] (deterministic split reduction).
 * With w_ref (= W_bar) and inner_partial the same pass emits locate_conv_wgrad_partials(geom) partial sums (double)
 * of <UNSCALED gw, W_bar>, which the spectral-norm backward needs; inv_scale, w_ref, inner_partial are nullable.
 * scale_group_batch > 0: gy of batch element b is weighted by inv_scale[(b / scale_group_batch) * scale_stride] instead
 * (gw = sum_k G_k / sigma_k over stacked forward calls; <= 4 groups; w_ref and inner_partial must be null). */
/* (layers with M % 192 == 0 on maps with even OH * OW and OW run on 192 x 128 tiles of the paired-load kernels only: gy must then
 * be 8-byte aligned with an even batch stride - any contiguous tensor is) */
size_t locate_conv_wgrad_workspace_bytes(const int* geom);
int locate_conv_wgrad_partials(const int* geom);
/* deferred_reduce (nullable; host memory, locate_slab_reduce_record_bytes() bytes): the split reduction of this launch is not
 * run but written there; gw / inner_partial are complete after locate_slab_reduce_batch() has run the record (at most
 * locate_slab_reduce_max() records per call; the workspace stays untouched until then).  locate_slab_reduce_record_blocks() == 0:
 * the geometry has no split reduction, nothing is pending.  One launch then finishes ALL weight gradients of a backward pass. */
int locate_conv_wgrad(const int* geom, const float* x, int64_t x_bs, const float* gy, int64_t gy_bs, float* gw,
                      const float* w_ref, const float* inv_scale, int scale_group_batch, int scale_stride,
                      double* inner_partial, void* workspace, int precision, const void* x_absmax, const void* gy_absmax,
                      void* deferred_reduce, void* stream);
/* Stacked calls (scale_group_batch > 0) WITH w_ref + inner_partial: the split reduction is laid out so that no slab crosses a
 * call boundary and emits the partial sums of <G_k / sigma_k, W_bar> per call - inner_partial[call][partial], groups x
 * locate_conv_wgrad_group_partials(geom, groups) doubles, workspace of locate_conv_wgrad_group_workspace_bytes(geom, groups) bytes.
 * Their sum over the partials equals <gy_k, y_k - bias> (locate_fin_sn_dots), without reading gy and y again.  0 partials: this
 * geometry cannot (1x1 maps, one output pixel, call lengths that do not divide into whole slabs) - use locate_fin_sn_dots. */
int locate_conv_wgrad_group_partials(const int* geom, int groups);
size_t locate_conv_wgrad_group_workspace_bytes(const int* geom, int groups);
size_t locate_slab_reduce_record_bytes(void);
int locate_slab_reduce_max(void);
int locate_slab_reduce_record_blocks(const void* record);
int locate_slab_reduce_batch(const void* records, int n, void* stream);

/* The weight gradients of ALL small-map layers of one backward pass in ONE launch: the style chain's linears
 * (libs/linear.py:7-15, libs/block.py:112-127), the channel gates' squeeze convs, the discriminator's layers on 1x1 maps, its
 * last 5x5 conv and its head (one output pixel) - in the reference ~19 separate autograd weight-gradient kernels per iteration.
 * locate_wgrad_batch_record() writes the launch of locate_conv_wgrad for one layer into a host record (same argument meaning)
 * and returns its block count, or 0 when the geometry is not such a layer (launch it with locate_conv_wgrad then);
 * locate_wgrad_batch() runs up to locate_wgrad_batch_max() packed records; results are those of the single launches. */
size_t locate_wgrad_batch_record_bytes(void);
int locate_wgrad_batch_max(void);
int locate_wgrad_batch_record(const int* geom, const float* x, int64_t x_bs, const float* gy, int64_t gy_bs, float* gw,
                              const float* w_ref, const float* inv_scale, int scale_group_batch, int scale_stride,
                              double* inner_partial, void* record);
int locate_wgrad_batch(const void* records, int n, void* stream);

/* ---- grouped convolutions of the SEPARABLE switch (libs/config.py:53; replaces the torch.nn.Conv2d /
 *      ConvTranspose2d(groups = ...) forward + autograd backward under libs/conv.py:14-18 and libs/attention.py:15-21).
 *      Depthwise: geom as above describes the REGULAR depthwise conv R between a "big" side [B, C, H, W] and a "small"
 *      side [B, M, OH, OW]; the side with more channels has one weight row [KH*KW] per channel, the other side has
 *      1/mult as many channels (channel o of the wide side pairs with channel o / mult of the narrow side):
 *        Conv2d(C, C*mult, groups=C)           forward = dwconv_fwd   (M = C*mult), input gradient = dwconv_dgrad
 *        ConvTranspose2d(M, M*mult, groups=M)  forward = dwconv_dgrad (C = M*mult), input gradient = dwconv_fwd
 *      w is the layer's weight tensor as stored (no packing).  scale / scale_group_batch / scale_stride as in locate_conv_fwd. ---- */
int locate_dwconv_fwd(const int* geom, const float* x, int64_t x_bs, const float* w, const float* scale, int scale_group_batch,
                      int scale_stride, float* y, int64_t y_bs, void* stream);
int locate_dwconv_dgrad(const int* geom, const float* gy, int64_t gy_bs, const float* w, const float* scale,
                        int scale_group_batch, int scale_stride, float* gx, int64_t gx_bs, void* stream);
/* gw[row, kh, kw] = inv_scale * sum_{b,oh,ow} small[b, row or row/mult, oh, ow] big[b, row/mult or row, oh*s-ph+kh, ow*s-pw+kw]
 * (deterministic two-stage reduction; square kernels up to 5 x 5).  w_ref / inner_partial / per-call scales as in
 * locate_conv_wgrad; locate_dwconv_wgrad_partials(geom) doubles are written. */
size_t locate_dwconv_wgrad_workspace_bytes(const int* geom);
int locate_dwconv_wgrad_partials(const int* geom);
int locate_dwconv_wgrad(const int* geom, const float* big, int64_t big_bs, const float* small, int64_t small_bs, float* gw,
                        const float* w_ref, const float* inv_scale, int scale_group_batch, int scale_stride,
                        double* inner_partial, void* workspace, void* stream);
/* Conv2d(G*cpg, G, kernel = the whole S x S map, groups = G) (libs/attention.py:15-21): y[b, g] = scale * <w[g, :], x[b, g, :]>
 * over L = cpg*S*S contiguous elements; x [B, G*L] with batch stride x_bs, w [G, L], y / gy [B, G] dense. */
int locate_groupdot_fwd(const float* x, int64_t x_bs, const float* w, const float* scale, int scale_group_batch,
                        int scale_stride, float* y, int B, int G, int L, void* stream);
int locate_groupdot_dgrad(const float* gy, const float* w, const float* scale, int scale_group_batch, int scale_stride,
                          float* gx, int64_t gx_bs, int B, int G, int L, void* stream);
int locate_groupdot_wgrad_partials(int G, int L);
int locate_groupdot_wgrad(const float* x, int64_t x_bs, const float* gy, float* gw, const float* w_ref, const float* inv_scale,
                          int scale_group_batch, int scale_stride, double* inner_partial, int B, int G, int L, void* stream);

/* ---- end-of-backward finalisers, batched over all layers of a backward pass (finalise.hip): what only feeds PARAMETER
 *      gradients - spectral norm's rank-1 term with du / dsigma (libs/spectral_norm.py:31-32 under autograd), the stacked
 *      calls' <gy_k, y_k - bias>, the gates' dgamma sums (libs/merge.py:33-38) - in one launch per kind instead of two or
 *      three small launches per layer.  `records`: HOST array of n records of locate_fin_record_bytes() = 112 bytes
 *      { const void* p[8]; int64 l[2]; int32 i[8]; } passed on BY VALUE in the kernel arguments (no device table):
 *        dots : p = {gy, y, bias|0, partial out [groups][np]}, l = {gy_bs, y_bs}, i = {groups, Bg, M, plane};
 *               np = locate_fin_sn_dot_partials(Bg, M, plane)
 *        rank1: p = {partial (double), sigma table, u, v, wv, gw (in/out), du|0, dsigma_out|0}, l = {wv_stride},
 *               i = {npartial per call, groups (0 = one call: locate_sn_weight_bwd's arithmetic; k >= 1 = stacked:
 *               locate_sn_weight_bwd_grouped's), sigma_stride, h, wd}
 *        sums : p = {partials (double), out (float)}, i = {count}
 *      Same arithmetic and summation order as the per-layer entry points: bit-identical results. ---- */
size_t locate_fin_record_bytes(void);
int locate_fin_sn_dot_partials(int Bg, int M, int plane);
int locate_fin_sn_dots(const void* records, int n, void* stream);
int locate_fin_sn_rank1(const void* records, int n, void* stream);
int locate_fin_sums(const void* records, int n, void* stream);
/* channel sums out[c] = sum_{b,hw} g[b,c,hw] (bias gradients: libs/scale.py:28-34, libs/linear.py:10) for all biases of a pass:
 * records p = {g, out [C], partials [slices][C] | 0}, l = {batch stride}, i = {B, C, hw}; slices = locate_fin_channel_slices */
int locate_fin_channel_slices(int B, int C, int hw);
int locate_fin_channel_sums(const void* records, int n, void* stream);

/* ---- gradient bucket pack / unpack of the data-parallel exchange (no reference counterpart: libs/config.py:10-11 is single
 *      device).  tensors: DEVICE array of {float* grad; float* flat; int64 n} records (locate_multi_copy_record_bytes() = 24);
 *      chunks: DEVICE (tensor index, chunk index) int pairs in pieces of locate_multi_copy_chunk_elems();
 *      direction 0: flat <- grad;  1: grad <- flat * scale ---- */
size_t locate_multi_copy_record_bytes(void);
int locate_multi_copy_chunk_elems(void);
int locate_multi_copy(const void* tensors, const void* chunks, int n_chunks, int direction, float scale, void* stream);

/* ---- fused multi-tensor Nadam (libs/nadam.py:31-89); per-tensor (step, m_schedule) state lives on device ---- */
size_t locate_nadam_tensor_record_bytes(void);   /* {float* p; const float* g; float* m; float* v; double* sched; int64 n;
                                                     uint32* absmax (nullable: locate_absmax_words() words, receive max |p| after the step)} */
int locate_nadam_chunk_elems(void);
int locate_nadam_step(const void* tensors, void* coef, const void* chunks, int n_tensors, int n_chunks, double lr, double beta1,
                      double beta2, double eps, double schedule_decay, double weight_decay, void* stream);

/* ---- loss glue (main.py:149-156,164-169, libs/utils.py:133-134, libs/grad_penalty.py:1-2): values and the
 *      gradients w.r.t. the discriminator outputs ---- */
int locate_d_loss(const float* d_true, const float* d_fake, const float* d_aug, int B, float gamma, float* losses,
                  float* g_true, float* g_fake, float* g_aug, void* stream);
int locate_g_loss(const float* d_fake, int B, float* loss, float* g_fake, void* stream);

#ifdef __cplusplus
}
#endif
#endif
