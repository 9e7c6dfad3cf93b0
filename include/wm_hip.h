/*
 * wm_hip.h -- C ABI of libwm_hip.so, the MI355X (gfx950) kernel library behind the
 * watermark embed -> attack -> extract training step.
 *
 * The reference (yingqichao/video-watermarking-forgery-detection) has no native code and
 * no FFI: every GPU kernel it runs is reached through torch.nn modules.  The entry points
 * below are therefore the operator boundary a maintainer of the reference would bind with
 * ctypes to replace, module by module, the torch ops listed next to
 * each function ("replaces:" = reference file:line, paths relative to the reference root).
 *
 * Conventions
 *   - plain C: raw device pointers + explicit sizes, no torch / C++ types;
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library never
 *     allocates, frees or keeps a pointer after the call returns;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant, and the library
 *     holds no mutable state between calls (no knobs, no environment variables, no set-then-call hints; the
 *     only per-thread datum is the text of the last error); safe under hipGraph capture.  The A/B switches
 *     (wm_debug_*, WM_NO_* variables) of tools/ exist only in the -DWM_DEBUG build lib/libwm_hip_dbg.so;
 *   - return value: 0 = ok, <0 = error (WM_E_*); wm_last_error_string() gives the text of
 *     the last error on the calling thread.  No C++ exception crosses the boundary;
 *   - activations are NHWC ("pixel-major"): element (b,h,w,c) lives at
 *     base[((b*H + h)*W + w) * ld + c], ld >= C the per-pixel channel stride (lets a tensor
 *     be a channel-slice of a wider one).  dtype of activations/packed weights:
 *     WM_F32 (parity path, exact f32 MFMA) or WM_BF16 (production path, f32 accumulate).
 *     Parameters, gradients of parameters, statistics and images at the attack boundary
 *     are always f32; images at the attack boundary are NCHW planes like the reference.
 */
#ifndef WM_HIP_H
#define WM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WM_F32 0
#define WM_BF16 1
#define WM_F16 2   /* the reference's autocast dtype (IRNcrop_model.py:340); same kernels and rates as WM_BF16, 10-bit mantissa, needs loss scaling */

#define WM_OK 0
#define WM_E_BADARG (-1)   /* null pointer, non-positive size, unsupported dtype */
#define WM_E_SHAPE (-2)    /* shape / alignment the kernels do not support */
#define WM_E_HIP (-3)      /* a HIP runtime call or launch failed */

const char* wm_last_error_string(void);
int wm_abi_version(void);

/* A BatchNorm-backward finalisation that rides on another launch (wm_conv3x3_wgrad_fin: a few extra workgroups of the
 * weight-gradient slab reduction) instead of costing one of its own.  Same arguments and result as wm_bn_bwd_finalize_raw
 * (mean != NULL: rows hold sum(gz), sum(gz*y)) / wm_bn_bwd_finalize (mean == NULL); nparts <= 256. */
typedef struct WmBnBwdFin {
    const float* partials; int nparts; int C; int CP; double count; const float* gamma; const float* mean; const float* invstd;
    float* dgamma; float* dbeta; int accumulate; float* coef;
} WmBnBwdFin;

/* ------------------------------------------------------------------ block-JPEG attack
 * replaces: noise_layers/jpeg.py:226-240 (Jpeg.forward), :259-273 (JpegSS), :295-306 (JpegMask)
 * with helpers :52-211.  x,y: [B,3,H,W] f32 NCHW.  mode 0 = round, 1 = round_ss, 2 = mask.
 * tables: host pointer to 128 floats = luminance[64] then chrominance[64] quantisation
 * tables, already scaled/rounded/clamped as std_quantization does (ignored for mode 2).
 * subsample: 0 or 2.  One fused kernel: x255, pad, RGB->YUV, (subsample), DCT, quantise,
 * de-quantise, IDCT, YUV->RGB, crop, /255.  Algorithmic traffic 24 B/px. */
#define WM_JPEG_ROUND 0
#define WM_JPEG_SS 1
#define WM_JPEG_MASK 2
int wm_jpeg_fwd(const float* x, float* y, int B, int H, int W, int mode, const float* tables,
                int subsample, void* stream);
/* ... and, act16 non-NULL, the attacked image a second time as [B][H][W][16] pixels of dtype act_dtype (channels 0..2, zero tail): the
 * input of the decoder's image-fed first layer (hidden_models/decoder.py:15), without wm_nchw_to_nhwc's launch. */
int wm_jpeg_fwd_act(const float* x, float* y, void* act16, int act_dtype, int B, int H, int W, int mode, const float* tables,
                    int subsample, void* stream);
/* gx = d(sum(gy*y))/dx.  x is read only for mode 1 (derivative of round_ss needs the
 * pre-rounding coefficient); mode 0 writes zeros (torch.round has zero gradient). */
int wm_jpeg_bwd(const float* x, const float* gy, float* gx, int B, int H, int W, int mode,
                const float* tables, int subsample, void* stream);

/* ------------------------------------------------------------------ DiffJPEG
 * replaces: utils/JPEG.py:256-291 (compress_jpeg) + :431-469 (decompress_jpeg) as composed
 * by DiffJPEG.forward :535-540.  H,W multiples of 16.  rounding 0 = torch.round,
 * 1 = round_only_at_0, 2 = diff_round.  factor = quality_to_factor(quality). */
int wm_diffjpeg_fwd(const float* x, float* y, int B, int H, int W, int rounding, float factor,
                    void* stream);
int wm_diffjpeg_bwd(const float* x, const float* gy, float* gx, int B, int H, int W, int rounding,
                    float factor, void* stream);

/* ------------------------------------------------------------------ stencil / resample attacks
 * All on [B,C,H,W] f32 NCHW planes (N = B*C planes of H x W).
 * gauss3: replaces noise_layers/gaussian_blur.py:53-56 (depthwise 3x3, zero pad 1); w9 = host
 *         pointer to the 9 weights.  Backward of a symmetric stencil = the same stencil. */
int wm_stencil3_fwd(const float* x, float* y, int N, int H, int W, const float* w9, void* stream);
/* median k x k (k = 3 or 5), zero padding: replaces noise_layers/middle_filter.py:5-13
 * (kornia MedianBlur).  idx (int8 per pixel, may be NULL) records which window tap was
 * selected so the backward can route the gradient to it. */
int wm_median_fwd(const float* x, float* y, int8_t* idx, int N, int H, int W, int k, void* stream);
int wm_median_bwd(const float* gy, const int8_t* idx, float* gx, int N, int H, int W, int k,
                  void* stream);
/* bicubic (A=-0.75, align_corners=False, no antialias) and bilinear resampling of the
 * sub-rectangle [h0,h0+hs) x [w0,w0+ws) of x[N,H,W] to y[N,OH,OW]; replaces
 * F.interpolate in noise_layers/resize.py:42-51 and noise_layers/crop.py:48-53.
 * clamp01 != 0 clamps the output to [0,1] (resize.py:53).  The backward is the transpose in
 * gather form (deterministic, no atomics): gx[N,H,W] gets zero outside the rectangle;
 * y_clamped (may be NULL) is the clamped forward output: where it is not strictly inside (0,1)
 * the clamp was active and the incoming gradient is dropped. */
#define WM_BILINEAR 0
#define WM_BICUBIC 1
int wm_resample_fwd(const float* x, float* y, int N, int H, int W, int h0, int hs, int w0, int ws,
                    int OH, int OW, int kind, int clamp01, void* stream);
int wm_resample_bwd(const float* gy, const float* y_clamped, float* gx, int N, int H, int W,
                    int h0, int hs, int w0, int ws, int OH, int OW, int kind, void* stream);
/* The same backward (replaces: autograd of the same F.interpolate calls) in two separable passes through a caller-owned workspace
 * tmp[N][OH][W] floats: x pass (gy, y_clamped -> tmp), y pass (tmp -> gx); 3x faster at the Resize attack's ratios; deterministic;
 * sums in a different order than the gather form (agreement ~1e-7 relative). */
int wm_resample_bwd_sep(const float* gy, const float* y_clamped, float* gx, float* tmp, int N, int H, int W,
                        int h0, int hs, int w0, int ws, int OH, int OW, int kind, void* stream);
/* Quantization: round(255 x)/255; replaces models/modules/Quantization.py:7-14 (the backward
 * is the identity and needs no kernel). */
int wm_quant_fwd(const float* x, float* y, size_t n, void* stream);

/* The clip loader's device path (replaces: /root/reference/data/Dataloader.py:22-57, the per-frame numpy -> cv2.resize -> torch chain):
 * decoded frames uint8 x[N][H][W][C] (C <= 4) -> float planes y[N][C][H][W] = x * scale (1/255 for images); the resize itself is
 * wm_resample_fwd(WM_BILINEAR) over the whole frame (half-pixel centres, no antialias = cv2.INTER_LINEAR's float arithmetic). */
int wm_u8_hwc_to_planes(const unsigned char* x, float* y, int N, int H, int W, int C, float scale, void* stream);
/* ------------------------------------------------------------------ layout / packing
 * image planes [B,C,H,W] f32 -> NHWC dtype with per-pixel stride ld, written at channel
 * offset c0; channels [c0+C, c0+C+zero_tail) are zero-filled (padding for MFMA K). */
int wm_nchw_to_nhwc(const float* x, void* y, int B, int C, int H, int W, int ld, int c0,
                    int zero_tail, int dtype, void* stream);
/* NHWC dtype (stride ld, offset c0) -> [B,C,H,W] f32 */
int wm_nhwc_to_nchw(const void* x, float* y, int B, int C, int H, int W, int ld, int c0, int dtype,
                    void* stream);
/* per-sample vector v[B,L] f32 broadcast over H x W into NHWC at channel offset c0
 * (the expanded message of hidden_models/encoder.py:34-37). */
int wm_broadcast_to_nhwc(const float* v, void* y, int B, int L, int H, int W, int ld, int c0,
                         int dtype, void* stream);
/* the tail of the encoder's concat tensor in one pass: channels [c0, c0+tail) of every pixel =
 * [message v[b, 0..L) | image planes img[b, 0..3) | zeros]  (encoder.py:34-40).  c0, tail, ld multiples of
 * the 16-byte vector width. */
int wm_concat_tail(const float* msg, const float* img, void* y, int B, int L, int H, int W, int ld, int c0,
                   int tail, int dtype, void* stream);
/* the whole concat row in one pass: y[..., :C] = relu(scale*x+shift) (the encoder features, fused BN+ReLU),
 * y[..., C:] = [message(L) | image(3) | 0]; x NHWC stride ldx, y NHWC stride ld. */
int wm_concat_full(const void* x, int ldx, const float* scale, const float* shift, const float* msg, const float* img,
                   void* y, int B, int C, int L, int H, int W, int ld, int dtype, void* stream);
/* relu(scale*x+shift) of an NHWC tensor copied into another NHWC tensor (channel slice):
 * materialises a BatchNorm+ReLU output where a consumer cannot fuse it (concat). */
int wm_bnrelu_copy(const void* x, int ldx, const float* scale, const float* shift, void* y, int ldy,
                   int c0, size_t npix, int C, int dtype, void* stream);
/* conv weight [Cout,Cin,3,3] f32 (PyTorch layout) -> packed [9][CoutP][CinP] dtype, zero padded.
 * perm (host int[Cin], may be NULL): packed input channel perm[ci] holds reference channel ci.
 * transpose != 0 builds the dgrad operand instead: [9][CinP][CoutP] with taps flipped. */
int wm_pack_w3x3(const float* w, void* wp, int Cout, int Cin, int CoutP, int CinP, const int* perm,
                 int transpose, int dtype, void* stream);
/* batched form: all convs of a network in ONE launch (the weights change every optimiser step, so every
 * step re-packs them).  jobs_dev: DEVICE array of njobs records of 48 bytes each,
 *   { const float* w; void* wp; const int* perm ( DEVICE int[Cin] or NULL ); int Cout, Cin, CoutP, CinP, transpose, pad; }
 * with the meaning of wm_pack_w3x3's arguments; max_elems = the largest 9*RowsP*ColsP among the jobs. */
int wm_pack_w3x3_batch(const void* jobs_dev, int njobs, size_t max_elems, int dtype, void* stream);

/* ------------------------------------------------------------------ conv 3x3, stride 1, pad 1
 * replaces: nn.Conv2d(…,3,1,padding=1) inside hidden_models/conv_bn_relu.py:11-15 and
 * network/UNet.py:67-97, as an implicit GEMM on MFMA (M = pixels, N = Cout, K = 9*Cin).
 *   x  : NHWC [B,H,W,Cin] stride ldx.  If in_scale/in_shift != NULL the kernel applies
 *        relu(in_scale[c]*x+in_shift[c]) while staging (fused BatchNorm+ReLU of the producer);
 *        zero padding is applied after that transform, as the reference pads the activated map.
 *   wp : packed weights from wm_pack_w3x3 (transpose=0), [9][CoutP][CinP]; CinP == Cin here.
 *   bias: f32[nbias] (nbias <= CoutP, the real Cout) or NULL.   y: NHWC [B,H,W,CoutP] stride ldy (raw conv output).
 *   stat_partials: NULL, or f32[wm_conv3x3_nparts(B,H,W,Cin,CoutP,dtype)][2][CoutP]: per-workgroup sums of y
 *        and y^2 (from the f32 accumulators) for the BatchNorm batch statistics.
 * Cin multiple of 16 (bf16) / 8 (f32); CoutP multiple of 32.  The same entry point computes
 * dgrad when given dy and the transposed pack. */
int wm_conv3x3_nparts(int B, int H, int W, int Cin, int CoutP, int dtype);
int wm_conv3x3_fwd(const void* x, int ldx, const void* wp, const float* bias, int nbias,
                   const float* in_scale, const float* in_shift, void* y, int ldy, float* stat_partials, int B, int H,
                   int W, int Cin, int CoutP, int dtype, int sweep_reverse, void* stream);
/* sweep_reverse (wm_conv3x3_fwd, wm_conv3x3_dgrad_applyfused / _bwdstats, wm_conv3x3_wgrad_fin): != 0 sweeps the pixel tiles
 * backwards.  A kernel that starts where the producer of its input stopped finds the freshest part of that tensor in the
 * Infinity Cache (256 MB against 134 MB per tensor at B=16, 256x256); the host alternates the direction along a chain of
 * layers.  Results do not depend on it except for the summation order inside the per-workgroup statistics rows.
 * weight gradient: dW[co,ci,kh,kw] = sum_{b,h,w} a[b,h+kh-1,w+kw-1,ci] * dy[b,h,w,co], with
 * a = x or relu(in_scale*x+in_shift).  Writes f32 partial slabs ws[nslabs][9][CinP][CoutP]
 * (nslabs = wm_conv3x3_wgrad_nslabs) and reduces them into dw[Cout,Cin,3,3] (PyTorch layout,
 * overwritten or accumulated).  perm as in wm_pack_w3x3 (device int[Cin] or NULL). */
int wm_conv3x3_wgrad_nslabs(int B, int H, int W);
size_t wm_conv3x3_wgrad_ws_bytes(int B, int H, int W, int CinX, int CoutY);
/* CinX / CoutY: channel counts of the x / dy tensors (padded, packed order);
 * Cin / Cout: the reference parameter's dims (dw is [Cout,Cin,3,3]). */
int wm_conv3x3_wgrad(const void* x, int ldx, int CinX, const float* in_scale, const float* in_shift,
                     const void* dy, int lddy, int CoutY, float* ws, float* dw, int accumulate, int B,
                     int H, int W, int Cin, int Cout, const int* perm_dev, int dtype, void* stream);
/* wm_conv3x3_wgrad / wm_conv3x3_wgrad_gvfused whose slab-reduction launch also carries a BatchNorm-backward finalisation of
 * ANOTHER layer (fin may be NULL): in a backward sweep the input-gradient kernel of layer l emits the sums of layer l-1
 * (wm_conv3x3_dgrad_bwdstats / _applyfused), and layer l's weight gradient is the next launch anyway. */
int wm_fin_rider_enabled(void);
int wm_conv3x3_wgrad_fin(const void* x, int ldx, int CinX, const float* in_scale, const float* in_shift, const void* dy, int lddy,
                         int CoutY, float* ws, float* dw, int accumulate, int B, int H, int W, int Cin, int Cout,
                         const int* perm_dev, int dtype, const WmBnBwdFin* fin, int sweep_reverse, void* stream);
/* The same with the BatchNorm-backward APPLY pass fused, for bf16 image-fed first layers (CinX <= 16) whose input needs
 * no gradient: dy is formed from g (gradient wrt the ReLU output, stride ldg), y (the raw conv output, stride ldy),
 * stats4 = f32[4][CoutY] = scale | shift | mean | invstd and coef = wm_bn_bwd_finalize's f32[3][CoutY] while the tile is
 * staged, and never written to memory (no wm_bn_bwd_apply, no dy tensor). */
int wm_conv3x3_wgrad_bnfused_supported(int CinX, int CoutY, int dtype);
int wm_conv3x3_wgrad_bnfused(const void* x, int ldx, int CinX, const void* g, int ldg, const void* y, int ldy, int CoutY,
                             const float* stats4, const float* coef, float* ws, float* dw, int accumulate, int B, int H, int W,
                             int Cin, int Cout, int dtype, void* stream);
/* The pair for a ConvBNRelu whose output was globally pooled (hidden_models/decoder.py:16-24, discriminator.py:14-22:
 * the gradient wrt its ReLU output is one row per sample, gvec f32[B][CoutY], already divided by H*W).  Both kernels
 * read the layer's raw conv output y instead of dy and apply the BatchNorm backward while staging (bf16, CinX = 64,
 * CoutY in {64, 32}); results are bit-identical to wm_bn_bwd_apply followed by wm_conv3x3_wgrad / wm_conv3x3_fwd.
 * wpt: the transposed packed filter (wm_pack_w3x3 with transpose = 1), dx: [B,H,W,CinP] dense. */
int wm_conv3x3_gvfused_supported(int CinX, int CoutY, int dtype);
int wm_conv3x3_wgrad_gvfused(const void* x, int ldx, int CinX, const float* in_scale, const float* in_shift, const float* gvec,
                             const void* y, int ldy, int CoutY, const float* stats4, const float* coef, float* ws, float* dw,
                             int accumulate, int B, int H, int W, int Cin, int Cout, int dtype, void* stream);
int wm_conv3x3_wgrad_gvfused_fin(const void* x, int ldx, int CinX, const float* in_scale, const float* in_shift, const float* gvec,
                                 const void* y, int ldy, int CoutY, const float* stats4, const float* coef, float* ws, float* dw,
                                 int accumulate, int B, int H, int W, int Cin, int Cout, int dtype, const WmBnBwdFin* fin, void* stream);
int wm_conv3x3_dgrad_gvfused(const void* y, int ldy, int CoutY, const void* wpt, const float* gvec, const float* stats4,
                             const float* coef, void* dx, int B, int H, int W, int CinP, int dtype, void* stream);
/* Input gradient (bf16, CoutY in {64,32} -> CinP = 64) whose epilogue also reduces the BatchNorm-backward sums of the
 * ConvBNRelu that FEEDS this layer (conv_bn_relu.py:11-15 stacked as in decoder.py:16-24): dx is that layer's g, and with
 * its raw output ry [B,H,W,64] and r_scale / r_shift the kernel emits partials f32[wm_conv3x3_nparts(..)][2][64] =
 * sum(gz), sum(gz*y) per channel (gz = bf16(dx)*[r_scale*ry + r_shift > 0]) -- wm_bn_bwd_reduce over (dx, ry) is not
 * needed; finish with wm_bn_bwd_finalize_raw.  dx is WRITTEN as gz (multiplied by the feeding layer's ReLU mask, like
 * wm_conv3x3_bwd_fused's; every consumer applies that mask itself, twice is the identity), so it may be handed to
 * wm_conv3x3_bwd_fused with g_premasked = 1.  src = this layer's dy, or (gvec, stats4, coef non-NULL) this layer's raw
 * output with the apply pass fused as in wm_conv3x3_dgrad_gvfused. */
/* The ordinary 64 -> 64 layer (gradient g a dense bf16 tensor [B,H,W,64]): input gradient with the BatchNorm-backward APPLY
 * pass fused.  The kernel reads g and the layer's raw output y (dense), forms dy from stats4 / coef while staging, writes
 * dy_out [B,H,W,64] (bit-identical to wm_bn_bwd_apply's; input of the wm_conv3x3_wgrad that follows; NULL: not written -- an input
 * gradient whose weight gradient nobody reads, e.g. the discriminator under the generator's loss, hidden.py:85-103) and dx = conv(dy, wpt)
 * [B,H,W,CinP], CinP = 64, or 32 for an image-fed layer (wpt [9][CinP][64]).
 * ry / r_scale / r_shift / partials: optional, all or none -- the sums of the feeding layer as in
 * wm_conv3x3_dgrad_bwdstats. */
int wm_conv3x3_dgrad_applyfused_supported(int CoutY, int CinP, int dtype);
int wm_conv3x3_dgrad_applyfused(const void* g, const void* y, const float* stats4, const float* coef, const void* wpt,
                                void* dy_out, void* dx, const void* ry, const float* r_scale, const float* r_shift,
                                float* partials, int B, int H, int W, int CinP, int dtype, int sweep_reverse, void* stream);
/* The whole backward of a 64 -> 64 body layer fed by another ConvBNRelu (16-bit activations), in ONE pass over its operands:
 * wm_conv3x3_bwd_fused reads g, y (this layer: gradient wrt its ReLU output, raw conv output) and xr (the FEEDING layer's raw conv
 * output, with its BatchNorm in_scale / in_shift), forms dy while staging and produces
 *   dx [B,H,W,64]              = conv(dy, wpt) * [in_scale * xr + in_shift > 0]: the gradient wrt the feeding layer's ReLU output,
 *                                already multiplied by that ReLU's mask (every consumer applies the mask itself; twice is the identity),
 *   partials [nwg][2][64]      = the feeding layer's BatchNorm-backward sums of dx (as wm_conv3x3_dgrad_bwdstats),
 *   ws [nwg][9][64][64]        = per-workgroup slabs of the weight gradient sum dy (x) ReLU(in_scale * xr + in_shift),
 * without writing dy or re-reading xr (537 MB instead of 939 MB per layer at B = 16, 256x256).  g_premasked != 0: g is such a masked
 * gradient (the staging then skips the mask arithmetic; the results are the same).  Exactly one of g and gvec is non-NULL: gvec
 * f32 [B][64] is the gradient of a globally pooled layer, one row per sample (B <= wm_conv3x3_bwd_fused_gvec_max_batch(); dy is then
 * formed from y alone, as in wm_conv3x3_dgrad_gvfused).  nwg = wm_conv3x3_bwd_fused_nwg(B,H,W).
 * wm_conv3x3_bwd_fused_reduce: dw [Cout][Cin][3][3] (+)= the sum of the slabs (wm_conv3x3_wgrad_fin's reduction, incl. `fin`). */
int wm_conv3x3_bwd_fused_supported(int dtype);
int wm_conv3x3_bwd_fused_supported_shape(int B, int H, int W, int dtype);   /* ... and B*H*W*64 < 2^31 (32-bit element offsets) */
int wm_conv3x3_bwd_fused_nwg(int B, int H, int W);
int wm_conv3x3_bwd_fused_gvec_max_batch(void);
/* which kernel wm_conv3x3_bwd_fused launches for these arguments: 8 = the role-split 8-wave form (csrc/bwd_ws8.hip: whole 8x16 tiles and a
 * premasked tensor gradient or a per-sample gradient), 1 = the single-role form (csrc/bwd_ws.hip: everything else) */
int wm_conv3x3_bwd_fused_kernel(int H, int W, int g_premasked, int has_gvec);
int wm_conv3x3_bwd_fused(const void* g, const float* gvec, const void* y, const float* stats4, const float* coef, const void* wpt,
                         const void* xr, const float* in_scale, const float* in_shift, void* dx, float* partials, float* ws, int B,
                         int H, int W, int dtype, int g_premasked, int sweep_reverse, void* stream);
int wm_conv3x3_bwd_fused_reduce(float* ws, float* dw, int accumulate, int B, int H, int W, int Cin, int Cout, const WmBnBwdFin* fin,
                                void* stream);
/* The encoder's after-concat layer WITHOUT the concat (16-bit dtypes).  replaces: /root/reference/hidden_models/encoder.py:34-41
 * (expand the message to [B,L,H,W], torch.cat([message, features, image]), then the 97 -> 64 ConvBNRelu of encoder.py:25).
 * A 3x3 conv is linear in its input channels and a message plane is constant over the image, so
 *   conv(cat) = conv64(features) + P,   P = conv3(image) + bias + sum_{taps inside the image} sum_l W[:, c_msg + l, tap] * msg[b, l].
 * wm_concat_side_fwd writes P [B,H,W,64] (dtype) in one pass from the f32 NCHW image planes img [B,3,H,W]; w [64][Cin][3][3] is the
 * layer's f32 weight in the reference's layout, the message channels are [c_msg, c_msg + L), the image channels [c_img, c_img + 3);
 * bias f32[64] or NULL; msg f32 [B][L].  Scratch owned by the caller: wside (4 KB, 16-byte aligned), mbias f32 [B][9][64].
 * wm_conv3x3_fwd_addin: y = conv3x3(relu(in_scale * x + in_shift), wp) + addend, dense [B,H,W,64] tensors, wp [9][64][64] from
 * wm_pack_w3x3 (the feature channels of w through its channel map); the BatchNorm statistics partials
 * f32[wm_conv3x3_nparts(B,H,W,64,64,dtype)][2][64] are taken AFTER the sum, so (y, partials) are the after-concat layer's.
 * Backward of the message channels: wm_concat_side_msg_wgrad, dw[:, c_msg + l, tap] (+)= sum_b msg[b,l] * S[b,tap,:], S = the sums of
 * dy [B,H,W,64] over the pixels for which `tap` lies inside the image; scratch: partial f32 [B][wm_concat_side_partial_rows()][64],
 * S f32 [B][9][64]; dw f32 [64][Cin][3][3].  (Feature and image channels of dw: wm_conv3x3_wgrad with a channel map.)
 * H, W >= 2; L <= 64; dtype WM_BF16 or WM_F16. */
int wm_concat_side_partial_rows(void);
int wm_concat_side_fwd(const float* img, const float* w, const float* bias, const float* msg, void* wside, float* mbias, void* P,
                       int B, int H, int W, int Cin, int c_msg, int L, int c_img, int dtype, void* stream);
int wm_concat_side_msg_wgrad(const void* dy, const float* msg, float* partial, float* S, float* dw, int accumulate, int B, int H,
                             int W, int Cin, int c_msg, int L, int dtype, void* stream);
int wm_conv3x3_fwd_addin(const void* x, const void* wp, const float* in_scale, const float* in_shift, const void* addend, void* y,
                         float* stat_partials, int B, int H, int W, int dtype, int sweep_reverse, void* stream);
/* conv + bias + ELU as ONE launch and that layer's backward in two (16-bit dtypes; rows f1 / f2).  replaces: the coupling subnets'
 * `elu(conv(x))` pairs, /root/reference/models/invertible_net.py:326-366 (ResBlock: conv1..conv4 + nn.ELU), whose autograd backward is
 * elu' * g, a bias column sum, the input gradient and the weight gradient -- five passes over the data here before.
 *   wm_conv3x3_fwd_elu       : out [B,H,W,64] = elu(conv3x3(x, wp) + bias); x [B,H,W,Cin] stride ldx, Cin in {16, 32, 64}; the pre-activation
 *                              is not stored: elu'(z) = out > 0 ? 1 : out + 1.
 *   wm_conv3x3_dgrad_elufused: g [B,H,W,64] = the gradient wrt out; forms gz = g * (out > 0 ? 1 : out + 1) while staging, writes
 *                              dx [B,H,W,CinP] = conv3x3(gz, wpt) (wpt from wm_pack_w3x3, transposed; CinP in {64, 32}; or CinP = 16 with
 *                              wpt packed to 32 rows, the upper 16 zero -- dx keeps its 16-channel stride), gz_out [B,H,W,64]
 *                              (NULL: not wanted) and bias_partials f32 [wm_conv3x3_dgrad_elufused_nparts(B,H,W)][64]: per-workgroup
 *                              column sums of gz (before its rounding to 16 bits).
 *   wm_conv3x3_wgrad_bias    : dw [Cout][Cin][3][3] (+)= the weight gradient from (x, gz) as wm_conv3x3_wgrad (ws: the same scratch), and in
 *                              the same reduction launch db [Cout] (+)= the column sums of bias_partials [nparts][CoutY]. */
int wm_conv3x3_fwd_elu_supported(int Cin, int CoutP, int dtype);
int wm_conv3x3_fwd_elu(const void* x, int ldx, const void* wp, const float* bias, int nbias, void* out, int B, int H, int W, int Cin,
                       int dtype, int sweep_reverse, void* stream);
int wm_conv3x3_dgrad_elufused_supported(int CinP, int dtype);
int wm_conv3x3_dgrad_elufused_nparts(int B, int H, int W);
int wm_conv3x3_dgrad_elufused(const void* g, const void* out, const void* wpt, void* dx, void* gz_out, float* bias_partials, int B,
                              int H, int W, int CinP, int dtype, int sweep_reverse, void* stream);
int wm_conv3x3_wgrad_bias(const void* x, int ldx, int CinX, const void* gz, int ldgz, int CoutY, float* ws, float* dw, int accumulate,
                          int B, int H, int W, int Cin, int Cout, int dtype, const float* bias_partials, int nparts, float* db,
                          int db_accumulate, void* stream);
/* The backward of an IMAGE-FED first ConvBNRelu (3 -> 64 channels; replaces autograd's backward of conv_bn_relu.py:11-15 for the layers of
 * decoder.py:16 / discriminator.py:13, whose input image needs a gradient) in ONE pass (csrc/bwd_ws16.hip): reads g, y [B,H,W,64] (gradient
 * wrt the layer's ReLU output, its raw conv output; stats4 / coef as wm_conv3x3_dgrad_applyfused) and the layer's input x [B,H,W,16] (the
 * image, 3 real channels), forms dy while staging and produces dx [B,H,W,16] = conv(dy, wpt) (wpt [9][16][64] from wm_pack_w3x3, transposed)
 * and dw [Cout][Cin][3][3] (+)= sum dy (x) x through ws (f32 [wm_conv3x3_bwd_fused16_nwg][9][16][64] slabs + wm_conv3x3_wgrad's reduction).
 * dy is never written: 335 MB per launch at B = 16, 256x256 instead of the two-kernel form's 637.  Whole-tile shapes only
 * (_supported: H % 8 == 0, W % 16 == 0, 16-bit dtype); g_premasked as wm_conv3x3_bwd_fused. */
int wm_conv3x3_bwd_fused16_supported(int B, int H, int W, int dtype);
int wm_conv3x3_bwd_fused16_nwg(int B, int H, int W);
int wm_conv3x3_bwd_fused16(const void* g, const void* y, const float* stats4, const float* coef, const void* wpt, const void* x, void* dx,
                           float* ws, float* dw, int accumulate, int B, int H, int W, int Cin, int Cout, int dtype, int g_premasked,
                           int sweep_reverse, void* stream);
int wm_conv3x3_dgrad_bwdstats_supported(int CoutY, int CinP, int dtype);
int wm_conv3x3_dgrad_bwdstats(const void* src, int lds, int CoutY, const void* wpt, const float* gvec, const float* stats4,
                              const float* coef, const void* ry, const float* r_scale, const float* r_shift, void* dx,
                              float* partials, int B, int H, int W, int CinP, int dtype, int sweep_reverse, void* stream);

/* ------------------------------------------------------------------ BatchNorm2d (training)
 * replaces nn.BatchNorm2d + nn.ReLU of conv_bn_relu.py:12-14 / UNet.py:67-97.
 * finalize: partial sums -> batch mean / biased var; scale = gamma*invstd,
 * (every *_finalize / colsum call treats its `partials` buffer as scratch: with more than 64 rows
 * it first folds them in place to 64 rows, so the pointer is const only in name),
 * shift = beta - mean*scale; running stats updated with `momentum` and the unbiased var
 * (count/(count-1)), as torch does.  C real channels, CP padded (scale=shift=0 there). */
int wm_bn_finalize(const float* partials, int nparts, int C, int CP, double count, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float momentum,
                   float eps, float* scale, float* shift, float* mean, float* invstd, void* stream);
/* backward through ReLU + BatchNorm given g = dL/d(relu output) and the raw conv output y:
 *   pass 1 (reduce): partial sums of gz = g*[scale*y+shift>0] and gz*xhat   -> partials
 *   finalize       : dgamma, dbeta (f32[C], overwritten or accumulated), coefficient vectors
 *   pass 2 (apply) : dy = scale*(gz - c1 - xhat*c2), written as NHWC dtype; also column sums
 *                    of dy for the conv bias gradient.
 * g may be a per-sample vector gvec[B,CP] (gradient of a global average pool, already
 * divided by H*W) instead of a tensor: pass g=NULL and gvec. */
int wm_bn_bwd_nparts(size_t npix);
int wm_bn_bwd_reduce(const void* g, int ldg, const float* gvec, const void* y, int ldy,
                     const float* scale, const float* shift, const float* mean, const float* invstd,
                     float* partials, int B, size_t hw, int CP, int dtype, void* stream);
int wm_bn_bwd_finalize(const float* partials, int nparts, int C, int CP, double count,
                       const float* gamma, const float* invstd, float* dgamma, float* dbeta,
                       int accumulate, float* coef /* [3][CP]: a=gamma*invstd, c1, c2 */, void* stream);
/* the same for partial rows that hold sum(gz), sum(gz*y) (wm_conv3x3_dgrad_bwdstats): xhat is applied here */
int wm_bn_bwd_finalize_raw(const float* partials, int nparts, int C, int CP, double count, const float* gamma,
                           const float* mean, const float* invstd, float* dgamma, float* dbeta, int accumulate,
                           float* coef, void* stream);
int wm_bn_bwd_apply(const void* g, int ldg, const float* gvec, const void* y, int ldy,
                    const float* scale, const float* shift, const float* mean, const float* invstd,
                    const float* coef, void* dy, int lddy, float* dbias_partials, int B, size_t hw,
                    int CP, int dtype, void* stream);
/* column sums of partial rows: out[c] (+)= sum_p partials[p][c] (bias gradients). */
int wm_colsum_finalize(const float* partials, int nparts, int C, int ldp, float* out, int accumulate,
                       void* stream);

/* ------------------------------------------------------------------ heads
 * global average pool of relu(scale*y+shift): replaces nn.AdaptiveAvgPool2d((1,1)) at
 * hidden_models/decoder.py:24 / discriminator.py:16.  out f32[B,CP]. */
int wm_avgpool_slices(size_t hw);
int wm_bnrelu_avgpool(const void* y, int ldy, const float* scale, const float* shift, float* out,
                      float* ws /* f32[B * wm_avgpool_slices(hw) * CP] */, int B, size_t hw, int CP, int dtype,
                      void* stream);
/* Training form: out3 f32[3][B][CP] = the pooled mean, and per (sample, channel) the number of active pixels N+ = #[z > 0]
 * and S+ = the sum of y over them (ws: 3x the size above).  The pooled layer's gradient is one value per (sample, channel),
 * so its BatchNorm-backward sums follow from N+ / S+ without a pass over y: wm_pooled_bn_bwd_rows writes the B partial rows
 * rows f32[B][2][CP] = (gvec*N+, gvec*S+) for wm_bn_bwd_finalize_raw (replaces wm_bn_bwd_reduce with gvec). */
int wm_bnrelu_avgpool_stats(const void* y, int ldy, const float* scale, const float* shift, float* out3, float* ws, int B,
                            size_t hw, int CP, int dtype, void* stream);
int wm_pooled_bn_bwd_rows(const float* gvec, const float* npos, const float* ysum, int B, int CP, float* rows, void* stream);
/* the two steps in one launch: dgamma, dbeta, coef of the pooled layer straight from (gvec, N+, S+) */
int wm_bn_bwd_finalize_pooled(const float* gvec, const float* npos, const float* ysum, int B, int C, int CP, double count,
                              const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                              int accumulate, float* coef, void* stream);
int wm_pool_stats_enabled(void);
/* 1x1 conv Cin->Cout (Cout <= 4) on relu(scale*y+shift): replaces nn.Conv2d(64,3,1) at
 * hidden_models/encoder.py:28,42 and the sigmoid head of network/UNet.py:41-43,65.
 * w f32[Cout,Cin], bias f32[Cout]; out f32 NCHW [B,Cout,H,W]; act 0 = none, 1 = sigmoid. */
int wm_conv1x1_head_fwd(const void* y, int ldy, const float* scale, const float* shift, const float* w,
                        const float* bias, float* out, int B, size_t hw, int Cin, int Cout, int act,
                        int dtype, void* stream);
/* ... and, act16 non-NULL (act == 0): the result a second time as [B][HW][16] pixels of the activation dtype (channels 0..Cout-1, zero
 * tail) -- the tensor the image-fed first layers of the decoder / discriminator read (hidden_models/decoder.py:15, discriminator.py:12),
 * which wm_nchw_to_nhwc would otherwise make from `out` in a launch of its own. */
int wm_conv1x1_head_fwd_act(const void* y, int ldy, const float* scale, const float* shift, const float* w,
                            const float* bias, float* out, void* act16, int B, size_t hw, int Cin, int Cout, int act,
                            int dtype, void* stream);
/* backward: gout f32 NCHW [B,Cout,H,W] (for act=1 the caller passes the gradient wrt the
 * pre-sigmoid logits) -> g NHWC dtype [B,H,W,Cin] (gradient wrt the ReLU output), and
 * partial sums for dw[Cout,Cin], dbias[Cout]: partials f32[nparts][Cout*(Cin+1)].
 * bn_partials (may be NULL): f32[nparts][2][Cin] = per-workgroup sum(gz), sum(gz*y) of the ConvBNRelu that produced y (gz = g as
 * stored * [scale*y+shift > 0]) -- the rows wm_bn_bwd_finalize_raw takes, so that layer needs no wm_bn_bwd_reduce pass. */
int wm_conv1x1_head_nparts(size_t npix);
int wm_conv1x1_head_bwd(const void* y, int ldy, const float* scale, const float* shift, const float* w,
                        const float* gout, void* g, int ldg, float* partials, float* bn_partials, int B, size_t hw, int Cin,
                        int Cout, int dtype, void* stream);

/* ------------------------------------------------------------------ UNet pieces
 * 2x2 max pool of relu(scale*y+shift): replaces nn.MaxPool2d(2,2) (network/UNet.py:14-20).
 * Writes the activated full-resolution map too when act_out != NULL (the skip tensor). */
int wm_bnrelu_maxpool2(const void* y, int ldy, const float* scale, const float* shift, void* pooled,
                       int ldp, void* act_out, int lda, int c0a, int B, int H, int W, int C, int dtype,
                       void* stream);
/* gradient of the pool scattered back to full resolution, added to g_skip when given. */
int wm_maxpool2_bwd(const void* y, int ldy, const float* scale, const float* shift, const void* gpooled,
                    int ldgp, const void* g_skip, int ldgs, void* g, int ldg, int B, int H, int W, int C,
                    int dtype, void* stream);
/* ConvTranspose2d k=2 s=2 (network/UNet.py:24-38) on relu(scale*x+shift):
 * w f32[Cin,Cout,2,2], bias f32[Cout]; y NHWC [B,2H,2W,*] at channel offset c0. */
int wm_upconv2x2_fwd(const void* x, int ldx, const float* scale, const float* shift, const float* w,
                     const float* bias, void* y, int ldy, int c0, int B, int H, int W, int Cin, int Cout,
                     int dtype, void* stream);
/* backward: gx = gradient wrt the (activated) input, NHWC [B,H,W,Cin]; w_t = the weight
 * rearranged to [(i,j,co)][Cin] f32 (weight.permute(2,3,1,0)); dw_partials f32
 * [wm_upconv2x2_dw_chunks(B,H,W)][Cin+1][4*Cout]: per-chunk partial sums of the weight gradient in
 * PyTorch order (row ci, column co*4+i*2+j) with the per-column sums of gy in the extra row Cin
 * (bias gradient = sum over the 4 taps); reduce with wm_colsum_finalize. */
int wm_upconv2x2_dw_chunks(int B, int H, int W);
int wm_upconv2x2_bwd(const void* x, int ldx, const float* scale, const float* shift, const float* w_t,
                     const void* gy, int ldgy, int c0, void* gx, int ldgx, float* dw_partials,
                     int B, int H, int W, int Cin, int Cout, int dtype, void* stream);
/* MFMA form of the same layer for bf16 with Cin % 64 == 0 and Cout % 16 == 0 (network/UNet.py:14-38 at every level):
 * four 1x1 GEMMs over the input pixels + pixel shuffle.  wm_upconv2x2_pack builds the two 16-bit (bf16 / f16) operands from the
 * PyTorch weight [Cin][Cout][2][2]: wf [(ij,co)][Cin] (forward) and wb [Cin][(ij,co)] (dgrad).
 *   fwd  : y[b,2h+i,2w+j,c0+co] = bias[co] + sum_ci relu(scale*x+shift)[b,h,w,ci] * w[ci,co,i,j]
 *   dgrad: gx[b,h,w,ci] = sum_(i,j,co) gy[b,2h+i,2w+j,c0+co] * w[ci,co,i,j]   (gradient wrt the ACTIVATED input)
 *   wgrad: dw[ci,co,i,j] (+)= sum_(b,h,w) relu(scale*x+shift)[b,h,w,ci] * gy[b,2h+i,2w+j,c0+co];  dbias[co] (+)= sum gy
 *          partial: f32[wm_upconv2x2_wgrad_nsplit][Cin][4*Cout], bias_partial: f32[nsplit][4*Cout] (scratch). */
int wm_upconv2x2_mfma_supported(int Cin, int Cout, int dtype);
int wm_upconv2x2_pack(const float* w, void* wf, void* wb, int Cin, int Cout, int dtype, void* stream);   /* dtype: WM_BF16 or WM_F16 */
int wm_upconv2x2_fwd_mfma(const void* x, int ldx, const float* scale, const float* shift, const void* wf, const float* bias,
                          void* y, int ldy, int c0, int B, int H, int W, int Cin, int Cout, int dtype, void* stream);
int wm_upconv2x2_dgrad_mfma(const void* gy, int ldgy, int c0, const void* wb, void* gx, int ldgx, int B, int H, int W, int Cin,
                            int Cout, int dtype, void* stream);
int wm_upconv2x2_wgrad_nsplit(int B, int H, int W, int Cin, int Cout);
int wm_upconv2x2_wgrad_mfma(const void* x, int ldx, const float* scale, const float* shift, const void* gy, int ldgy, int c0,
                            float* partial, float* bias_partial, float* dw, float* dbias, int accumulate, int B, int H, int W,
                            int Cin, int Cout, int dtype, void* stream);

/* ------------------------------------------------------------------ losses / optimiser
 * sum((a-b)^2) partials and gradient 2*w*(a-b)/n of nn.MSELoss (hidden.py:37,90).
 * gscale_dev (here and in wm_bce_logits / wm_message_loss / wm_bce_logits_target / wm_mse_fwd_bwd_gated; may be NULL): a device
 * scalar multiplied into gscale -- the loss scale of mixed-precision training (wm_amp_*), which lives on the device. */
int wm_mse_fwd_bwd(const float* a, const float* b, float* grad_a, float gscale, const float* gscale_dev, float* loss_partials,
                   int nparts, size_t n, void* stream);
/* hidden.py:85-101 at the encoded image, one pass: out [B][C][HW] f32 = (f32)g[b][q][c0 + c] + gscale * (a - b), with g the NHWC input
 * gradient of the discriminator's first layer (dtype, pixel stride ld; C <= 4 real channels from c0), a = encoded, b = cover (f32 NCHW);
 * loss_partials [nparts] = partial sums of (a - b)^2 over the blocks (wm_hidden_metrics adds them).  Same values as wm_nhwc_to_nchw +
 * wm_mse_fwd_bwd + wm_axpy(1.0). */
int wm_image_grad_mse(const void* g, int ld, int c0, const float* a, const float* b, float* out, float gscale, const float* gscale_dev,
                      float* loss_partials, int nparts, int B, int C, int H, int W, int dtype, void* stream);
/* y = a + s*b (f32), used to combine image-space gradients. */
int wm_axpy(float* a, const float* b, float s, size_t n, void* stream);
/* Adam / AdamW step over one flat f32 parameter buffer (torch.optim.Adam semantics,
 * hidden.py:24-25: lr 1e-3, betas (0.9,0.999), eps 1e-8, weight_decay 0, no amsgrad).
 * step = 1-based step count.  decoupled != 0 gives AdamW. */
int wm_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int decoupled, int step, float grad_scale,
                 void* stream);
/* The same step for a hipGraph-captured training step (a Python step per batch per rank, /root/reference/train.py:99-109, replayed
 * instead of re-enqueued): the two step-count-dependent constants come from DEVICE memory instead of the argument list.
 * wm_adam_hyper (host, no launch): out2 = {lr / (1 - beta1^step), sqrt(1 - beta2^step)} -- exactly the values wm_adam_step derives;
 * wm_adam_step_dev reads them from hyper_dev (device float[2]), which the caller refreshes before each replay: bit-identical
 * to wm_adam_step(step). */
int wm_adam_hyper(float lr, float beta1, float beta2, int step, float* out2);
int wm_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int decoupled, const float* hyper_dev, float grad_scale, void* stream);
/* sum of squares partials (clip_grad_norm_) */
/* nn.Linear after the global average pool (hidden_models/decoder.py:26,32-34, discriminator.py:18,24-26), B ~ 16, I,O <= 64:
 * fwd: out[B,O] = pooled[:, :I] @ w[O,I]^T + bias.
 * bwd: dw[O,I] (+)= g_out^T @ pooled, db[O] (+)= sum_b g_out, gvec[B,CP] = (g_out @ w) * inv_hw zero padded to CP -- the
 *      per-sample gradient vector wm_bn_bwd_* take for a globally pooled output. */
int wm_linear_head_fwd(const float* pooled, int ldp, const float* w, const float* bias, float* out, int B, int I, int O,
                       void* stream);
int wm_linear_head_bwd(const float* pooled, int ldp, const float* w, const float* g_out, float* dw, float* db,
                       int accumulate, float* gvec, int CP, float inv_hw, int B, int I, int O, void* stream);
/* The four small launches a pooled head runs in a row -- wm_linear_head_fwd, the loss (kind 0: wm_bce_logits against the constant label
 * `target`; kind 1: wm_message_loss against messages [B][O]), wm_linear_head_bwd and wm_bn_bwd_finalize_pooled of the pooled ConvBNRelu --
 * as ONE launch of one workgroup, each stage with the arithmetic (and operation order) of the kernel it replaces.
 * out3 [3][B][CP] = wm_bnrelu_avgpool_stats' result (pooled mean, N+, S+); logits [B][O]; loss_out [1] (kind 0) / [2] (kind 1: MSE, bitwise
 * error); dw [O][I], db [O] (+)=; gvec [B][CP]; C, count, gamma, mean, invstd, dgamma, dbeta, coef [3][CP] as wm_bn_bwd_finalize_pooled.
 * wm_pooled_head_supported: the sizes one workgroup's LDS holds (B*CP <= 2048, O*I <= 4096, B*O <= 1024). */
int wm_pooled_head_supported(int B, int CP, int I, int O);
int wm_pooled_head(const float* out3, int B, int CP, int I, int O, const float* w, const float* bias, int kind, float target,
                   const float* messages, float gscale, const float* gscale_dev, float* logits, float* loss_out, float* dw, float* db,
                   int accumulate, float* gvec, float inv_hw, int C, double count, const float* gamma, const float* mean,
                   const float* invstd, float* dgamma, float* dbeta, float* coef, void* stream);
/* nn.BCEWithLogitsLoss (mean) of n logits against a constant label (hidden_models/hidden.py:68-97): *loss_out = the
 * loss, grad_out[n] (may be NULL) = gscale * d loss / d logits.  One small launch. */
int wm_bce_logits(const float* logits, float target, int n, float gscale, const float* gscale_dev, float* loss_out, float* grad_out,
                  void* stream);
/* decoder message loss (hidden.py:96-99,109-111) on n = B*L values: out2[0] = mean (d-m)^2,
 * out2[1] = mean |clip(round(d),0,1) - m| (the bitwise error), grad_out[n] (may be NULL) = gscale * (d - m). */
int wm_message_loss(const float* decoded, const float* messages, int n, float gscale, const float* gscale_dev, float* out2,
                    float* grad_out, void* stream);
/* the seven scalars hidden.py:105-113 logs, in one launch: out7 = [w_adv*adv + w_enc*enc + w_dec*dec, enc, dec, bit error,
 * adv, d_cover, d_encoded] with enc = sum(enc_partials[nparts]) / n_img (wm_mse_fwd_bwd's rows), msg2 = wm_message_loss's
 * out2, adv / d_cover / d_enc = device scalars written by wm_bce_logits. */
int wm_hidden_metrics(const float* enc_partials, int nparts, double n_img, const float* msg2, const float* adv,
                      const float* d_cover, const float* d_enc, float w_adv, float w_enc, float w_dec, float* out7, void* stream);
int wm_sumsq(const float* x, size_t n, float* partials, int nparts, void* stream);

/* ------------------------------------------------------------------ mixed precision: torch.cuda.amp.GradScaler on the device
 * replaces: models/IRNcrop_model.py:143 (GradScaler()), :407 (scaler.scale(loss).backward()), :413-416 (scaler.step x2, scaler.update()).
 * state: f32[WM_AMP_STATE] device array = [0] loss scale, [1] growth tracker, [2] growth_factor, [3] backoff_factor,
 * [4] growth_interval, [8+k] found_inf of optimiser k, [12+k] step count of optimiser k (k < 4).  The loss kernels take &state[0]
 * as gscale_dev, so every gradient of the backward is multiplied by the scale; then, per optimiser,
 *   wm_amp_found_inf : found_inf[k] = !isfinite(sum of its wm_sumsq rows)       (GradScaler.unscale_'s inf check)
 *   wm_adam_step_amp : the Adam step on g / scale, skipped when found_inf[k]; bias corrections from the device step count,
 *                      which a skipped step does not advance (like torch's per-parameter `step`)
 * and once per iteration wm_amp_update: GradScaler.update() (backoff on any inf, growth after growth_interval clean iterations),
 * advances the step counts of the optimisers that stepped and clears the flags.  No host synchronisation anywhere. */
#define WM_AMP_STATE 16
int wm_amp_found_inf(const float* const* sumsq_partials, const int* nparts, int ngroups, float* state, int k, void* stream);
int wm_adam_step_amp(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int decoupled, float grad_scale, const float* amp_state, int k, void* stream);
int wm_amp_update(float* state, int noptimizers, void* stream);

/* ------------------------------------------------------------------ tamper-localisation branch (elementwise, f32 NCHW planes)
 * replaces: models/IRNcrop_model.py:320-322 (clamp_with_grad), :344-345 / :372-373 (Quantization after the clamp),
 * :348 (splice), :379-388 (PSNR of the int-truncated images, metrics.py:30-46, and the 1.0 / 0.8 forward weight at 33 dB),
 * :378,391-393 (BCEWithLogitsLoss on the predicted mask), :410-412 (clip_grad_norm_).
 * wm_clamp_quant_fwd: y = round(255*clamp(x,0,1))/255 (backward of both = identity: nothing to launch).
 * wm_splice_fwd: q = clamp_quant(enc); fwd_q (may be NULL) = q; tampered (may be NULL) = q*(1-mask) + prev*mask with mask
 *   [B,1,HW] broadcast over the C planes; psnr_partials [wm_splice_nparts(B*C*HW)] doubles (with real; both may be NULL) =
 *   partial sums of (int(255*real) - int(255*q))^2.
 * wm_psnr_gate: out2[0] = PSNR (0 when the images are equal, like metrics.py:41-42), out2[1] = PSNR < threshold ? w_below : w_above.
 * wm_mse_fwd_bwd_gated: wm_mse_fwd_bwd with the gradient scale multiplied by the device scalar gate_dev[0].
 * wm_bce_logits_target: *loss_out = mean BCE-with-logits(p, target), grad_out (may be NULL) = gscale * d loss / d p;
 *   partials = scratch [nparts <= 2048]; chain_sigmoid != 0: p is a sigmoid output s(z) (the UNet head, network/UNet.py:65) and
 *   grad_out is taken wrt z (multiplied by p*(1-p)).
 * wm_masked_axpy: a += g * (1 - mask) (the splice's backward onto the gradient of the encoded image).
 * wm_mask_threshold: out[i] = p[i] > threshold (uint8 tamper mask).
 * wm_clip_coef: partials[k][nparts[k]] = wm_sumsq rows of up to 4 flat gradient buffers clipped TOGETHER;
 *   out2[0] = min(1, max_norm / (total_norm + 1e-6)), out2[1] = total_norm.   wm_scale_dev: x *= scale_dev[0]. */
int wm_clamp_quant_fwd(const float* x, float* y, size_t n, void* stream);
int wm_splice_nparts(size_t n);
int wm_splice_fwd(const float* enc, const float* real, const float* prev, const float* mask, float* fwd_q, float* tampered,
                  double* psnr_partials, int B, int C, size_t HW, void* stream);
int wm_psnr_gate(const double* psnr_partials, int nparts, double n, float threshold, float w_below, float w_above, float* out2,
                 void* stream);
int wm_mse_fwd_bwd_gated(const float* a, const float* b, float* grad_a, float gscale, const float* gate_dev, const float* gscale_dev,
                         float* loss_partials, int nparts, size_t n, void* stream);
int wm_bce_logits_target(const float* p, const float* target, size_t n, float gscale, const float* gscale_dev, float* partials, int nparts,
                         float* loss_out, float* grad_out, int chain_sigmoid, void* stream);
int wm_masked_axpy(float* a, const float* g, const float* mask, int B, int C, size_t HW, void* stream);
int wm_mask_threshold(const float* p, float threshold, uint8_t* out, size_t n, void* stream);
int wm_clip_coef(const float* const* partials, const int* nparts, int ngroups, float max_norm, float* out2, void* stream);
int wm_scale_dev(float* x, size_t n, const float* scale_dev, void* stream);

/* ------------------------------------------------------------------ general layer family
 * replaces, for models/networks.py:631-749 (Discriminator), models/conditional_jpeg_generator.py:40-79 (conv()), :83-96 (ResBlock),
 * :185-200 (QFAttention), :202-374 (FBCNN), :697-826 (QF_predictor): nn.Conv2d / nn.ConvTranspose2d / nn.Linear of any kernel size
 * up to 5x5, stride 1 or 2, zero padding, with bias; their input and weight gradients; the activations; AdaptiveAvgPool2d((1,1));
 * symm_pad (:865-885) / ReplicationPad2d; nn.utils.spectral_norm (networks.py:1381-1385); the Bayar constraint (:814-817).
 * Activations NHWC [B,H,W,Cp] of `dtype` (WM_F32 / WM_BF16 / WM_F16) with the channel stride Cp a multiple of 16 and the padding
 * channels ignored by every consumer; parameters, per-sample vectors and their gradients f32 in torch's layouts.
 * wm_gconv_pack : w [Cout][Cin][KH][KW] f32 -> wp [KH*KW][RP][CP] dtype, rows = Cout, cols = Cin (transpose 0) or rows = Cin,
 *                 cols = Cout (transpose 1: the operand of the input gradient / of ConvTranspose2d's forward); padding zero.
 * wm_gconv_fwd  : dgrad 0: out [B,OH,OW,NC] = conv(in [B,IH,IW,KC], w [taps][NC][KC]) + bias [NC] (may be NULL);
 *                 dgrad 1: in = the conv's output-side tensor [B,IH,IW,KC], out = its input-side tensor [B,OH,OW,NC]
 *                 (the conv's input gradient, or ConvTranspose2d's forward), w packed with transpose 1.
 * wm_gconv_wgrad: dw [Cout][Cin][KH][KW] (+)= sum_pixels dout x in; dbias [Cout] (+)= column sums of dout (may be NULL);
 *                 partial = f32 scratch of wm_gconv_wgrad_scratch_floats(..) floats.  IH/IW/KC describe `in`, OH/OW/NC `dout`.
 * wm_gcolsum    : out [Creal] (+)= column sums of x [npix][C]; scratch: wm_gcolsum_scratch_floats(npix, C) floats.
 * wm_unary_fwd / _bwd: kind 0 ReLU, 1 LeakyReLU(0.2), 2 GELU (erf), 3 ELU, 4 Sigmoid, 5 Tanh; the backward reads the INPUT x --
 *                 except kind 6 (backward only): the ELU derivative from the layer's OUTPUT, out > 0 ? 1 : out + 1 (wm_conv3x3_fwd_elu keeps no input).
 * wm_add_scaled : out = a + alpha * b.
 * wm_qfatt_fwd  : out = x + gamma[b,c] * res + beta[b,c] (gamma / beta f32 [B][ldv]); wm_qfatt_bwd: gres = gamma * g,
 *                 ggamma[b,c] = sum_p g * res, gbeta[b,c] = sum_p g (the gradient wrt x is g itself); scratch:
 *                 wm_qfatt_bwd_scratch_floats(B, hw, ldv) floats.
 * wm_gpool_fwd  : out f32 [B][C] = mean over the hw pixels of x [B,hw,C]; wm_gpool_bwd: gx = g / hw.
 * wm_pad_nchw_to_nhwc(_bwd): x [B,C,H,W] f32 -> out [B,H+top+bottom,W+left+right,CP], mode 0 symmetric, 1 replicate (pads may
 *                 be 0: a plain layout change); the backward sums every padded position back onto its source pixel.
 * wm_gunpack_nchw(_bwd): the top-left H x W window and first C channels of x [B,PH,PW,CP] -> [B,C,H,W] f32, and its adjoint.
 * wm_spectral_norm_fwd: W [M][N] = weight_orig.view(Cout,-1); do_iter != 0 (training): v = normalize(W^T u), u = normalize(W v)
 *                 in place (eps 1e-12); sigma = u.(W v); Wsn = W / sigma; scratch: wm_spectral_norm_scratch_floats(M, N) floats.
 *                 wm_spectral_norm_bwd: gW (+)= (G - <G,Wsn> u v^T)/sigma
 *                 (u, v detached, as torch computes them under no_grad); partial = f32[256] scratch.
 * wm_bayar_constrain: every 5x5 filter of w [nfilters][25] in place: centre := 0, filter /= its sum, centre := -1. */
int wm_gconv_pack(const float* w, void* wp, int Cout, int Cin, int KH, int KW, int RP, int CP, int transpose, int dtype, void* stream);
int wm_gconv_fwd(const void* in, const void* w, const float* bias, void* out, int B, int IH, int IW, int KC, int OH, int OW, int NC,
                 int KH, int KW, int stride, int pad, int dgrad, int dtype, void* stream);
int wm_gconv_wgrad_nsplit(int B, int OH, int OW, int KC, int NC, int KH, int KW);
int wm_gconv_wgrad(const void* dout, const void* in, float* partial, float* dw, float* dbias, int accumulate, int B, int IH, int IW,
                   int KC, int OH, int OW, int NC, int KH, int KW, int stride, int pad, int Cout, int Cin, int dtype, void* stream);
size_t wm_gconv_wgrad_scratch_floats(int B, int OH, int OW, int KC, int NC, int KH, int KW);
size_t wm_gcolsum_scratch_floats(size_t npix, int C);
int wm_gcolsum(const void* x, size_t npix, int C, float* out, int Creal, int accumulate, float* scratch, int dtype, void* stream);
int wm_unary_fwd(const void* x, void* y, size_t n, int kind, int dtype, void* stream);
int wm_unary_bwd(const void* x, const void* gy, void* gx, size_t n, int kind, int dtype, void* stream);
/* gx = gy * act'(x) over x [npix][C] AND out [Creal] (+)= the column sums of gx as stored -- the bias gradient of the nn.Conv2d whose
 * output the activation took (conditional_jpeg_generator.py / invertible_net.py:326-366's conv -> ELU / ReLU pairs): one pass over the
 * data + a reduce of the split partials.  part: wm_unary_bwd_colsum_scratch_floats(npix, C) floats. */
size_t wm_unary_bwd_colsum_scratch_floats(size_t npix, int C);
int wm_unary_bwd_colsum(const void* x, const void* gy, void* gx, size_t npix, int C, int kind, float* part, float* out, int Creal,
                        int accumulate, int dtype, void* stream);
int wm_add_scaled(const void* a, const void* b, void* out, size_t n, float alpha, int dtype, void* stream);
int wm_qfatt_fwd(const void* x, const void* res, const float* gamma, const float* beta, void* out, int B, size_t hw, int C, int ldv,
                 int dtype, void* stream);
size_t wm_qfatt_bwd_scratch_floats(int B, size_t hw, int ldv);
int wm_qfatt_bwd(const void* g, const void* res, const float* gamma, void* gres, float* ggamma, float* gbeta, float* scratch, int B, size_t hw,
                 int C, int ldv, int dtype, void* stream);
int wm_gpool_fwd(const void* x, float* out, int B, size_t hw, int C, int dtype, void* stream);
int wm_gpool_bwd(const float* g, void* gx, int B, size_t hw, int C, int dtype, void* stream);
int wm_pad_nchw_to_nhwc(const float* x, void* out, int B, int C, int H, int W, int left, int right, int top, int bottom, int mode,
                        int CP, int dtype, void* stream);
int wm_pad_nchw_to_nhwc_bwd(const void* gp, float* gx, int B, int C, int H, int W, int left, int right, int top, int bottom, int mode,
                            int CP, int dtype, void* stream);
int wm_gunpack_nchw(const void* x, float* out, int B, int C, int H, int W, int PH, int PW, int CP, int dtype, void* stream);
int wm_gunpack_nchw_bwd(const float* g, void* gx, int B, int C, int H, int W, int PH, int PW, int CP, int dtype, void* stream);
size_t wm_spectral_norm_scratch_floats(int M, int N);
int wm_spectral_norm_fwd(const float* W, float* u, float* v, float* sigma, float* Wsn, float* scratch, int M, int N, int do_iter, void* stream);
int wm_spectral_norm_bwd(const float* G, const float* Wsn, const float* u, const float* v, const float* sigma, float* partial,
                         float* gW, int M, int N, int accumulate, void* stream);
int wm_bayar_constrain(float* w, int nfilters, void* stream);

/* ------------------------------------------------------------------ invertible embedder pieces
 * replaces, for models/invertible_net.py: HaarDownsampling / HaarUpsampling (:178-292: F.conv2d / F.conv_transpose2d with the fixed
 * 2x2 Haar filters, groups = channels), RNVPCouplingBlock's affine (:140-141,153-173), and the channel narrow / cat around the
 * subnets (:150-151,175,318-322,363).  NHWC tensors of `dtype`; the coupling subnets' convolutions are wm_gconv_*.
 * wm_haar: up 0 (analysis): in [B,2H,2W,CPin] with C channels -> out [B,H,W,CPout], channel 4c+k = fac * (Haar filter k of channel c),
 *          k = 0 sum, 1 horizontal, 2 vertical, 3 diagonal difference; up 1 (synthesis): in [B,H,W,CPin] with 4C channels ->
 *          out [B,2H,2W,CPout] with C channels, scaled by fac.  Each is the other's adjoint (= its backward) for equal fac; padding
 *          channels of `out` are written as zero.  Bit 1 of `up` (values 2, 3): HaarDownsampling(order_by_wavelet=True) (:207-218,
 *          :225-233) -- the 4C channels ordered wavelet-major, k C + c.
 * wm_chan_copy: dst[p][doff + c] = src[p][soff + c] for c < n over npix pixels (strides in elements).
 * wm_coupling_fwd: rev 0: y = e(s) * x + t; rev 1: y = (x - t) / e(s); e(s) = exp(clamp * (2 sigmoid(s) - 1)) + eps.
 * wm_coupling_bwd: gradients wrt x, s, t from g = dL/dy; v = x for rev 0, v = the forward's OUTPUT y for rev 1. */
int wm_haar(const void* in, void* out, int B, int H, int W, int C, int CPin, int CPout, float fac, int up, int dtype, void* stream);
int wm_chan_copy(const void* src, void* dst, size_t npix, int sstride, int soff, int dstride, int doff, int n, int dtype, void* stream);
/* dst [npix][dstride] written WHOLE in one launch: a's channel window [aoff, aoff + na) lands at channel adst, b's (b may be NULL) at bdst,
 * zero elsewhere: x.narrow / torch.cat on the channel dimension (invertible_net.py:135-145, :358) and their adjoints on stride-padded NHWC. */
int wm_chan_place(const void* a, int astride, int aoff, int adst, int na, const void* b, int bstride, int boff, int bdst, int nb, void* dst,
                  int dstride, size_t npix, int dtype, void* stream);
int wm_coupling_fwd(const void* x, const void* s, const void* t, void* y, size_t n, float clamp, float eps, int rev, int dtype, void* stream);
int wm_coupling_bwd(const void* g, const void* v, const void* s, void* gx, void* gs, void* gt, size_t n, float clamp, float eps, int rev,
                    int dtype, void* stream);

/* ------------------------------------------------------------------ losses of the literal IRNrhi step
 * replaces, in models/IRNrhi_model.py:425-560: nn.SmoothL1Loss (:148,476,481), nn.BCELoss (:147,492-493,505), nn.CrossEntropyLoss
 * (:156,454,485), torch.clamp(x,0,1) with its gradient mask (:430,472), PSNR of the postprocess()ed images (:527, metrics.py:30-46).
 * Each loss call writes the mean loss to a device scalar and, when grad_out != NULL, d loss / d input for an upstream gradient of 1
 * (the host scales it by the loss weight with wm_scale_dev).  partials: f32 scratch [nparts <= 2048] (doubles for the PSNR: feed
 * them to wm_psnr_gate).  f32 only. */
int wm_smooth_l1(const float* a, const float* b, size_t n, float beta, float* partials, int nparts, float* loss_out, float* grad_out,
                 void* stream);
int wm_bce_prob(const float* p, float target, size_t n, float* partials, int nparts, float* loss_out, float* grad_out, void* stream);
int wm_cross_entropy(const float* logits, const long long* labels, int B, int K, int ld, float* loss_out, float* grad_out, void* stream);
int wm_clamp01_fwd(const float* x, float* y, size_t n, void* stream);
int wm_clamp01_bwd(const float* x, const float* g, float* gx, size_t n, void* stream);
int wm_psnr255_partials(const float* a, const float* b, size_t n, double* partials, int nparts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WM_HIP_H */
