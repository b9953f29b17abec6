/* vg_api.h — C ABI of libvg_hip.so, the MI355X (gfx950) kernel library behind the VAE-GAN
 * training step.
 *
 * The reference (adnanemagourt/vae-gan) has no FFI of its own: its hot path is the PyTorch op
 * stream issued by <arch>/model.py and <arch>/main.py:68-111.  Each entry point below therefore cites the
 * reference lines whose arithmetic it replaces.  Conventions:
 *   - plain pointers + sizes only; every buffer (incl. workspaces) is caller-owned device memory;
 *   - all work is enqueued on the hipStream_t passed as `void* stream`; no hidden syncs;
 *   - return 0 on success, <0 on error; message via vg_last_error() (thread-local);
 *   - activations are NHWC with a padded channel stride (`cs`, multiple of 32 for bf16 / 16 for
 *     f32); `dtype` selects the storage/arithmetic type of activations and packed weights
 *     (VG_BF16: bf16 storage + bf16 MFMA with f32 accumulate; VG_F32: f32 storage + f32 MFMA).
 *     Master weights, gradients, statistics and optimizer state are always f32.
 */
#ifndef VG_API_H
#define VG_API_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VG_F32 = 0, VG_BF16 = 1 };
enum { VG_OK = 0, VG_ERR_BAD_ARG = -1, VG_ERR_HIP = -2, VG_ERR_UNSUPPORTED = -3 };
enum { VG_ACT_NONE = 0, VG_ACT_LRELU = 1, VG_ACT_SIGMOID = 2 };

#define VG_STATS_REPLICAS 32 /* per-channel f64 accumulators are spread over this many copies */

/* Reproducible runs (no counterpart in the reference, whose CUDA kernels are not reproducible either).
 * on != 0: every per-channel statistics buffer ([VG_STATS_REPLICAS][...] f64 slots in the entry points below) is
 * accumulated as 64-bit fixed-point bins instead of doubles (integer adds: any order gives the same bits; the caller
 * still only zeroes the buffer and hands it back to the consuming entry point; across ranks it must be summed as
 * int64), the kernels that split a reduction over blocks through f32 atomics (im2col-row weight gradient, split-K
 * forward) run unsplit, and the VQ EMA sums are taken in row order.  Two runs of the same step then give the same
 * bits.  Process-wide; set it before the buffers of a step are filled, not between a producer and its consumer. */
void vg_set_deterministic(int on);
int vg_get_deterministic(void);

const char* vg_last_error(void);
int vg_version(void);

/* Geometry of one convolution "weight object" W[O][I][R][S] with stride s, padding p, as
 * nn.Conv2d(I, O, R, stride=s, padding=p) sees it.  nn.ConvTranspose2d(I_t, O_t, R, s, p) is the
 * same object with O = I_t, I = O_t and forward/dgrad swapped. */
typedef struct {
  int n, h, w;        /* batch and spatial size of the conv INPUT (x)            */
  int oh, ow;         /* spatial size of the conv OUTPUT (y)                     */
  int ci, co;         /* logical channel counts I and O                          */
  int xcs, ycs;       /* channel strides (padded) of the x-side / y-side tensors */
  int r, s;           /* kernel height/width                                     */
  int stride, pad;
  int dtype;          /* VG_F32 | VG_BF16                                        */
} vg_conv_desc;

/* Development/ablation switch: 1 = im2col-row implicit GEMM only, 2 = + spatial-tile (halo) kernel where it
 * applies, 3 = + ping-pong spatial-tile kernel where it applies (default); + 16: the forward of 4x4 stride-2 convs
 * stays on the im2col-row kernel instead of the parity-plane spatial-tile kernel (A/B timing). */
void vg_set_conv_kernel(int version);
/* Same for the weight gradient: 1 = per-tap pixel-reduction kernel only, 3 = + spatial-tile kernel (default). */
void vg_set_wgrad_kernel(int version);
/* Name of the gather-GEMM kernel variant the calling thread's last vg_conv_fwd / vg_conv_dgrad launched
 * (e.g. "igemm3<bf16,4,2,2,4,3x3>"); bench.py keys its live per-kernel timings on it. */
const char* vg_last_conv_kernel(void);
/* The same for the calling thread's last vg_conv_wgrad ("wgrad3<4,2,1>", "wgrad3s2<4,1,1>", "wgrad<bf16,per-tap>"):
 * parity tests assert which weight-gradient kernel they covered. */
const char* vg_last_wgrad_kernel(void);

/* ---- weight packing ---------------------------------------------------------------------- */
/* Bytes of the packed forward (transposed=0) or dgrad (transposed=1) weight image. */
size_t vg_conv_packed_bytes(const vg_conv_desc* d, int transposed);
/* OIHW f32 master weight -> packed [phase][Npad][tap][Cpad] image in `dtype`, zero padded.
 * Replaces the implicit weight layout handling inside torch's conv (e.g. vq_vae_gan/model.py:146). */
int vg_conv_pack_weight(const vg_conv_desc* d, int transposed, const float* w_oihw, void* packed, void* stream);

/* The same for many layers in one launch (all packed images of one optimizer after its step).  vg_conv_pack_item
 * fills one table entry of vg_conv_pack_item_bytes() bytes on the HOST for (desc, transposed, master weight, packed
 * image); the caller copies the table to the device once (the pointers do not change between steps) and
 * vg_conv_pack_batch re-packs all `n_items` entries, which must share `dtype`. */
size_t vg_conv_pack_item_bytes(void);
int vg_conv_pack_item(const vg_conv_desc* d, int transposed, const float* w_oihw, void* packed, void* item);
int vg_conv_pack_batch(const void* items_dev, int n_items, int dtype, void* stream);

/* ---- the three conv primitives ----------------------------------------------------------- */
/* y = act(conv(x, W) + bias) [+ y if accumulate].  `stats` (optional): f64 [VG_STATS_REPLICAS][2][Npad]
 * zeroed by the caller; receives per-channel sum / sum of squares of the (dtype-rounded) pre-activation
 * output, for the BatchNorm that follows.  Replaces nn.Conv2d forward (beta_vae_gans/model.py:52,59;
 * vq_vae_gan/model.py:146,153,161,179,193,211,216,221; hierarchical_vae_gan/model.py:96,116,135,...)
 * and nn.ConvTranspose2d's input-gradient. */
int vg_conv_fwd(const vg_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                double* stats, int act, int accumulate, void* stream);
/* Same result as vg_conv_fwd (no stats / accumulate) for shapes with few output tiles and a long reduction:
 * the K loop is cut into `ksplit` slices (<=0: chosen by the library) whose f32 partial sums meet in `ws`
 * (vg_conv_splitk_ws_bytes(d) bytes).  Carries the Linear layers of beta_vae_gans/model.py:74-75,87-88,107,139
 * (a Linear over [B,F] is a 1x1 conv over an NHWC [B,1,1,F] tensor; an f32 [out][in] weight IS its packed image). */
size_t vg_conv_splitk_ws_bytes(const vg_conv_desc* d);
int vg_conv_fwd_splitk(const vg_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                       int act, int ksplit, void* ws, void* stream);
/* dx = act(convT(dy, W) + bias) [+ dx]: the conv's input gradient, and nn.ConvTranspose2d's forward
 * (vq_vae_gan/model.py:186; beta_vae_gans/model.py:123).  w_packed is the transposed image. */
int vg_conv_dgrad(const vg_conv_desc* d, const void* dy, const void* w_packed_t, const float* bias, void* dx,
                  double* stats, int act, int accumulate, void* stream);

/* The input-gradient launch of the layer that CONSUMED a BatchNorm's output produces that BatchNorm's upstream
 * gradient; these variants also take the BatchNorm backward sums (per channel: sum dz, sum dz*xhat, with
 * dz = dout * LeakyReLU'(.)) in the kernel's epilogue, replacing one vg_bn_bwd_reduce pass over two tensors
 * (torch.nn.BatchNorm2d's backward inside every Sequential of <arch>/model.py, e.g. vq_vae_gan/model.py:143-158).
 * x: the BatchNorm's raw input, same NHWC layout as the gradient being written; y: its output (only when the mask
 * must be read there: residual BatchNorm), else the mask is x*scale+shift > 0; sums: f64
 * [VG_STATS_REPLICAS][2][cpad], zeroed by the caller.  *fused = 1 when the kernel variant that ran took the sums
 * (the ping-pong spatial-tile kernels, bf16); 0 otherwise: sums untouched, call vg_bn_bwd_reduce. */
typedef struct vg_bn_bwd_stats {
  const void* x;
  const void* y;
  const float* mean;
  const float* invstd;
  const float* scale;
  const float* shift;
  double* sums;
  int cpad;
  int act;
  int mask_out; /* != 0: a launch that fuses stores dz = dout * act'(.) instead of dout (rounded to the tensor's dtype;
                   the sums are those of the stored values): the BatchNorm backward then runs vg_bn_bwd_apply with
                   act = VG_ACT_NONE and y = NULL, and a residual BatchNorm's skip gradient is that tensor itself */
} vg_bn_bwd_stats;
int vg_conv_dgrad_bnstats(const vg_conv_desc* d, const void* dy, const void* w_packed_t, void* dx, int accumulate,
                          const vg_bn_bwd_stats* bs, int* fused, void* stream);
int vg_conv_fwd_bnstats(const vg_conv_desc* d, const void* x, const void* w_packed, void* y, int accumulate,
                        const vg_bn_bwd_stats* bs, int* fused, void* stream);
/* dW[O][I][R][S] (f32, OIHW) = sum over pixels of dy ⊗ im2col(x).  `ws` is f32 scratch of
 * vg_conv_wgrad_ws_bytes(d) bytes (one packed image of dW for the per-tap kernel; up to ~75 MB for the spatial-tile
 * kernels, which write one partial image per tile group with plain stores and sum them in a fixed order: no
 * atomics, bit-reproducible).  Serves both Conv2d.weight.grad and ConvTranspose2d.weight.grad
 * (roles of x / dy swapped by the caller).  accumulate != 0: dW += (the gradient goes straight into a
 * parameter's .grad, which the training step zeroed: a parameter used twice per backward sums up). */
size_t vg_conv_wgrad_ws_bytes(const vg_conv_desc* d);
int vg_conv_wgrad(const vg_conv_desc* d, const void* x, const void* dy, float* dw_oihw, void* ws, int accumulate,
                  void* stream);

/* ---- thin layers: convolutions with a 3-channel side, bf16 (csrc/thin.hip) --------------------
 * The image-side convolutions of every model (first conv of the encoders and discriminators, e.g.
 * vq_vae_gan/model.py:146 and :211; last conv of the decoders, :193) and their gradients.  The 3-channel tensor is
 * NHWC with a channel stride of FOUR (8 bytes per pixel, channel 3 zero): desc.xcs == 4 (thin input) or desc.ycs == 4
 * (thin output); the other side has <= 64 channels at stride 64.  The master OIHW f32 weights are read directly (no
 * packed image).  vg_thin_conv_supported: 0 = not a thin layer (use the vg_conv_* entry points on 32-padded tensors),
 * 1 = thin input (3x3 stride 1 pad 1, or 4x4 stride 2 pad 1), 2 = thin output (3x3 stride 1 pad 1).
 * ws (forward, input gradient): vg_thin_conv_ws_bytes() bytes of scratch for the bf16 weight image of the launch. */
int vg_thin_conv_supported(const vg_conv_desc* d);
size_t vg_thin_conv_ws_bytes(void);
int vg_thin_conv_fwd(const vg_conv_desc* d, const void* x, const float* w_oihw, const float* bias, void* y, double* stats,
                     int act, void* ws, void* stream);
int vg_thin_conv_dgrad(const vg_conv_desc* d, const void* dy, const float* w_oihw, void* dx, void* ws, void* stream);
size_t vg_thin_conv_wgrad_ws_bytes(const vg_conv_desc* d);
int vg_thin_conv_wgrad(const vg_conv_desc* d, const void* x, const void* dy, float* dw_oihw, void* ws, int accumulate,
                       void* stream);

/* ---- the PatchGAN tail: Conv2d(C, 1, 4, stride 1, padding 1), bf16 (csrc/tail.hip) -------------
 * The last layer of every discriminator (vq_vae_gan/model.py:221, beta_vae_gans/model.py:176,
 * hierarchical_vae_gan/model.py:330: 512 -> 1 on a 16x16 map at 256x256 inputs).  One output channel is a dot product
 * per pixel, not a GEMM tile: forward and weight gradient run on the vector ALUs straight from the master OIHW f32
 * weights (rounded to bf16 like the packed images); results agree with vg_conv_fwd / vg_conv_wgrad up to the f32
 * summation order.  The weight gradient writes one partial image per wave into `ws`
 * (vg_conv1_wgrad_ws_bytes(d) bytes) and sums them in a fixed order: bit-reproducible.  The input gradient stays on
 * vg_conv_dgrad.  vg_conv1_supported: co == 1, 4x4, stride 1, pad 1, ci % 8 == 0, output width <= 32, ycs in 8..64. */
int vg_conv1_supported(const vg_conv_desc* d);
int vg_conv1_fwd(const vg_conv_desc* d, const void* x, const float* w_oihw, const float* bias, void* y, int act,
                 void* stream);
size_t vg_conv1_wgrad_ws_bytes(const vg_conv_desc* d);
int vg_conv1_wgrad(const vg_conv_desc* d, const void* x, const void* dy, float* dw_oihw, void* ws, int accumulate,
                   void* stream);

/* per-channel column sum: out[c] (f32) = sum_m a[m][c], c < c_valid.  bias gradients.
 * ws: VG_STATS_REPLICAS*cs f64 of scratch.  accumulate != 0: out[c] +=. */
int vg_channel_sum(const void* a, int64_t rows, int cs, int c_valid, int dtype, float* out, double* ws, int accumulate,
                   void* stream);

/* vg_channel_sum for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc): no counterpart in the reference.  Same result as vg_channel_sum; the f64 replica scratch is a
 * static device array inside the library (256 KiB per GPU, held until the process ends), so there is no `ws`
 * argument and the call allocates nothing.  Limit: cs <= 1024 (VG_ERR_BAD_ARG beyond, with a message).  All calls on
 * one GPU share that array: calls that overlap on one GPU (two streams, two threads) race; calls on one stream are
 * ordered and each clears what it uses.  The FIRST call on a device resolves the array's address (hipGetSymbolAddress,
 * which may load the code object): make one warm-up call per device outside any graph capture; every later call only
 * enqueues a memset and two kernels.  Same layouts as vg_channel_sum, the bf16 cs == 4 image-side layout included. */
int vg_channel_sum_noalloc(const void* a, int64_t rows, int cs, int c_valid, int dtype, float* out, int accumulate,
                           void* stream);

/* ---- layout ------------------------------------------------------------------------------ */
/* NCHW f32 [n][c][h][w]  <->  NHWC `dtype` [n][h][w][cs] (channels >= c written as zero). */
int vg_nchw_to_nhwc(const float* src, void* dst, int n, int c, int h, int w, int cs, int dtype, void* stream);
int vg_nhwc_to_nchw(const void* src, float* dst, int n, int c, int h, int w, int cs, int dtype, void* stream);

/* dst[p][y][x] = mean of the 2x2 block of src (NCHW f32, `planes` = n*c): the 'area' reconstruction target of the
 * documented non-reference shim D1 (the beta decoder emits 128x128, beta_vae_gans/model.py:104,121-126). */
int vg_avgpool2_nchw(const float* src, float* dst, int64_t planes, int h, int w, void* stream);

/* ---- BatchNorm (training mode) + LeakyReLU(0.2) + residual ---------------------------------- */
/* From the f64 stats of vg_conv_fwd: mean/invstd (biased var, eps), affine scale/shift, and the
 * running-stat update with momentum (unbiased var) + num_batches_tracked += 1, applied `updates` times
 * (updates = 2 stands for the two bit-identical encoder forwards of one VQ / hierarchical iteration,
 * vq_vae_gan/main.py:78,99, of which only one is executed).
 * nn.BatchNorm2d training forward (every <arch>/model.py; e.g. vq_vae_gan/model.py:18-19,147,154). */
int vg_bn_finalize(const double* stats, int c, int cpad, int64_t count, const float* gamma, const float* beta,
                   float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches,
                   int updates, float* mean, float* invstd, float* scale, float* shift, void* stream);
/* stats[VG_STATS_REPLICAS][2][cpad] (f64, zeroed by caller) += per-channel sum / sum of squares of x:
 * for a BatchNorm whose input is not a conv output (beta_vae_gans/model.py:117). */
int vg_channel_stats(const void* x, double* stats, int64_t rows, int cs, int c, int cpad, int dtype, void* stream);
/* Data parallel with global-batch statistics (no counterpart in the single-device reference, whose nn.BatchNorm2d sees
 * the whole batch, e.g. vq_vae_gan/model.py:143-158): vg_stats_collapse reads the `nslots`
 * accumulators of a [VG_STATS_REPLICAS][nslots] statistics buffer (nslots = 2 * cpad for a BatchNorm) into a dense f64
 * vector, the only thing the ranks exchange (one all-reduce of nslots doubles); vg_stats_load puts the summed vector back
 * into the buffer in the format the finalize kernels read (either accumulator mode, see vg_set_deterministic). */
int vg_stats_collapse(const double* stats, int nslots, double* vec, void* stream);
int vg_stats_load(const double* vec, int nslots, double* stats, void* stream);
/* y = act(x*scale[c] + shift[c] (+ res)); act per VG_ACT_*.  x,res,y: [rows][cs] `dtype`. */
int vg_bn_act_fwd(const void* x, const void* res, void* y, const float* scale, const float* shift, int64_t rows,
                  int cs, int c, int act, int dtype, void* stream);
/* Backward, pass 1: sums[rep][2][cpad] (f64, zeroed) += {sum dz, sum dz*xhat} with
 * dz = dout * (y > 0 ? 1 : 0.2) (act=LRELU) and xhat = (x-mean)*invstd.
 * y may be NULL when the forward had no residual and act is LRELU or NONE: the mask is then recomputed as
 * x*scale+shift > 0 with the forward's own scale/shift (vg_bn_finalize), which saves reading y. */
int vg_bn_bwd_reduce(const void* dout, const void* y, const void* x, const float* mean, const float* invstd,
                     const float* scale, const float* shift, double* sums, int64_t rows, int cs, int c, int cpad,
                     int act, int dtype, void* stream);
/* vg_bn_bwd_reduce that also writes the masked gradient dz = dout * act'(.) (dz may be dout: in place) and sums the
 * values as stored.  For a residual BatchNorm (the tail of ResidualBlock, vq_vae_gan/model.py:24-27: out = lrelu(bn2(.) +
 * x)) the apply pass then reads two tensors instead of three and writes one instead of two: vg_bn_bwd_apply(dout = dz,
 * y = NULL, act = VG_ACT_NONE, dres = NULL), the skip gradient being dz. */
int vg_bn_bwd_reduce_dz(const void* dout, const void* y, const void* x, const float* mean, const float* invstd,
                        const float* scale, const float* shift, double* sums, void* dz, int64_t rows, int cs, int c,
                        int cpad, int act, int dtype, void* stream);
/* Backward, pass 2: dx = gamma*invstd*(dz - sum_dz/n - xhat*sum_dzxhat/n); dres = dz (optional);
 * dgamma = sum_dzxhat, dbeta = sum_dz (f32 [c]). */
/* NOTE: `sums` must be followed by 3*cpad f32 of scratch (coefficients), i.e. allocate
 * VG_STATS_REPLICAS*2*cpad f64 + 3*cpad f32. */
/* y may be NULL under the same conditions as in vg_bn_bwd_reduce.  accumulate != 0: dgamma, dbeta +=.
 * dxsum (optional, f32 [c]) += per-channel sum of the stored dx: the bias gradient of the conv that produced x
 * (nn.Conv2d bias backward, e.g. vq_vae_gan/model.py:17-18), taken here instead of by one more pass over dx.
 * count = number of rows the sums run over (0: = rows; data parallel with global-batch statistics: the rows of
 * all ranks, `sums` having been summed over the ranks); dgamma/dbeta are the sums times dgb_scale (1, or 1/world
 * in that mode, where every rank holds the global sums and the gradient exchange adds them up again). */
int vg_bn_bwd_apply(const void* dout, const void* y, const void* x, const float* mean, const float* invstd,
                    const float* scale, const float* shift, const float* gamma, const double* sums, void* dx,
                    void* dres, float* dgamma, float* dbeta, int accumulate, float* dxsum, int64_t rows, int64_t count,
                    float dgb_scale, int cs, int c, int cpad, int act, int dtype, void* stream);
/* Plain elementwise activation backward (no BN): dx = dout * act'(y). */
int vg_act_bwd(const void* dout, const void* y, void* dx, int64_t n, int act, int dtype, void* stream);
/* a += b (elementwise, `dtype`): gradient fan-in of the residual skip. */
int vg_add_inplace(void* a, const void* b, int64_t n, int dtype, void* stream);

/* ---- Vector quantiser (vq_vae_gan/model.py:68-134) ---------------------------------------- */
/* idx[i] = argmin_k (|z_i|^2 + |E_k|^2) - 2 z_i.E_k   in f32, that op order, first-min ties
 * (model.py:88-95).  z: [rows][d] f32 row-major (NHWC flat), codebook [k][d] f32.
 * norms_ws: optional caller-owned scratch of k floats (receives |E_k|^2); with it and d in {64,128,256} the
 * register-resident search kernel is used (same arithmetic and indices, several times faster on large codebooks). */
int vg_vq_assign(const float* z, const float* codebook, int64_t rows, int k, int d, int64_t* idx, float* norms_ws,
                 void* stream);
/* The same indices as vg_vq_assign, bit for bit, for large codebooks (vq_vae_gan/model.py:88-95 at num_embeddings =
 * 8192, embedding_dim = 256: BASELINE configs[4]): approximate distances on the bf16 MFMA with a rigorous error bound
 * pick the candidates of every row (twice the cheap product: first the bound on the row's minimum, then the codes
 * within the bound of it), the candidates' distances are then taken with the exact-f32 instruction sequence of
 * vg_vq_assign, smallest distance, then smallest index.  A row with more than 32 candidates (or none) makes the call
 * redo the whole search with the exact kernel (still the same result; *fallback_flag_out, a device int, optional,
 * receives 1 then, else 0).  d in {64, 128, 256}; ws: vg_vq_shortlist_ws_bytes(rows, k, d) bytes, 256-byte aligned. */
size_t vg_vq_shortlist_ws_bytes(int64_t rows, int k, int d);
int vg_vq_assign_shortlist(const float* z, const float* codebook, int64_t rows, int k, int d, int64_t* idx, void* ws,
                           int* fallback_flag_out, void* stream);
/* zq = z + (E[idx] - z); sse += sum (E[idx]-z)^2 (f64 scalar, zeroed by caller); optional EMA
 * statistics counts[k] (f32) and esum[k][d] (f32), zeroed by caller (model.py:98-119). */
int vg_vq_gather(const float* z, const float* codebook, const int64_t* idx, int64_t rows, int k, int d, float* zq,
                 double* sse, float* counts, float* esum, void* stream);
/* EMA + renormalise (model.py:114-129): cluster = cluster*decay + (1-decay)*counts; ema_w likewise;
 * codebook = ema_w / ((cluster+1e-5)/(n+k*1e-5)*n). */
int vg_vq_ema_update(float* ema_cluster, float* ema_w, float* codebook, const float* counts, const float* esum,
                     int k, int d, float decay, float* nsum_scratch /* 1 f32 */, void* stream);
/* dz = g_zq + (*g_loss) * coef * (z - zq) where coef = 2*commitment/numel (model.py:105-107,132);
 * zq is the forward's output (its value is the pre-EMA code vector).  g_zq / g_loss may be null. */
int vg_vq_bwd(const float* g_zq, const float* g_loss, const float* z, const float* zq, int64_t n, float coef,
              float* dz, void* stream);

/* ---- loss reductions (<arch>/model.py loss_function) ------------------------------------------- */
/* out[0] (f64, zeroed) += sum (a-b)^2 over valid channels; a,b NHWC `dtype` [rows][cs]. */
int vg_sqdiff_sum(const void* a, const void* b, int64_t rows, int cs, int c, int dtype, double* out, void* stream);
/* da = (*g) * coef * (a - b) on valid channels, 0 on padding. */
int vg_sqdiff_bwd(const void* a, const void* b, const float* g, float coef, int64_t rows, int cs, int c, int dtype,
                  void* da, void* stream);
/* out[0] += sum over x of f(x): mode 0: x ; 1: relu(1-x) ; 2: relu(1+x).  x f32 flat. */
int vg_reduce_f32(const float* x, int64_t n, int mode, double* out, void* stream);
/* dx = (*g)*coef*f'(x) for the same modes. */
int vg_reduce_f32_bwd(const float* x, const float* g, float coef, int64_t n, int mode, float* dx, void* stream);
/* z = mu + eps*exp(0.5*logvar);  kl[0] (f64) += sum(1 + lv - mu^2 - exp(lv))  (beta model.py:211-216,258). */
int vg_reparam_kl_fwd(const float* mu, const float* logvar, const float* eps, int64_t n, float* z, double* kl, void* stream);
/* dmu = gz + (*gkl)*ckl*(-2mu)... see kernel; gradients of z and of the KL sum w.r.t. mu/logvar. */
int vg_reparam_kl_bwd(const float* mu, const float* logvar, const float* eps, const float* gz, const float* gkl,
                      float ckl, int64_t n, float* dmu, float* dlogvar, void* stream);
/* out[c] = sum_r a[r][c] for a row-major f32 [rows][cols] matrix with few rows (Linear bias gradients). */
int vg_colsum_f32(const float* a, int rows, int64_t cols, float* out, void* stream);
/* torch.clamp(x, lo, hi) (beta_vae_gans/model.py:91, hierarchical_vae_gan/model.py:168).  g == NULL: forward
 * y = clamp(x); g != NULL: backward y = g where lo <= x <= hi, else 0. */
int vg_clamp(const float* x, const float* g, float lo, float hi, float* y, int64_t n, void* stream);
/* f64 scalar -> f32 scalar with scale: dst[0] = (float)(src[0]*scale). */
int vg_scalar_finish(const double* src, double scale, float* dst, void* stream);

/* ---- fused Adam over a flat parameter buffer (<arch>/main.py:34-44, torch.optim.Adam defaults) --- */
int vg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                 float eps, int step, float grad_scale, void* stream);

/* ---- spatial self attention (hierarchical_vae_gan/model.py:24-47) ---------------------------------- */
/* o[b][i][:] = sum_j softmax_j(q_i . k_j) v[b][j][:] with NO 1/sqrt(d) scale (model.py:37-42).  All f32, NHWC-flat:
 * q,k [b][hw][d], v,o [b][hw][c]; lse [b][hw] (log-sum-exp per query, kept for the backward).  hw % 32 == 0,
 * d even <= 32, c in {32,64,128,256}.  The [hw,hw] matrix is never materialised. */
int vg_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int b, int hw, int d, int c,
                void* stream);
/* Gradients of the above w.r.t. q, k, v given d_o; delta_ws: b*hw f32 of scratch. */
int vg_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                float* dq, float* dk, float* dv, float* delta_ws, int b, int hw, int d, int c, void* stream);
/* The same attention on the bf16 MFMA (v_mfma_f32_32x32x16_bf16), for the bf16 compute mode: q, k [b*hw][32] and
 * v [b*hw][c] are the bf16 NHWC tensors the 1x1 convs wrote (q/k channels beyond the real width are zero padding),
 * o / d_o / dq / dk / dv stay f32.  Scores are exact products of the stored bf16 values; P and dS are rounded to bf16
 * once.  vg_attn_supported_bf16 says whether a shape is covered (hw % 128 == 0, d == 32, c in {128, 256}). */
int vg_attn_supported_bf16(int hw, int d, int c);
int vg_attn_fwd_bf16(const void* q, const void* k, const void* v, float* o, float* lse, int b, int hw, int d, int c,
                     void* stream);
int vg_attn_bwd_bf16(const void* q, const void* k, const void* v, const float* o, const float* d_o, const float* lse,
                     float* dq, float* dk, float* dv, float* delta_ws, int b, int hw, int d, int c, void* stream);
/* y = gamma[0]*o + x (model.py:45); backward: d_o = gamma*gy, dgamma[0] (f64, zeroed by caller) += sum gy*o. */
int vg_gamma_residual(const float* o, const float* x, const float* gamma, float* y, int64_t n, void* stream);
int vg_gamma_residual_bwd(const float* gy, const float* o, const float* gamma, float* d_o, double* dgamma, int64_t n,
                          void* stream);
/* elementwise storage cast bf16 <-> f32 (n % 8 == 0). */
int vg_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream);

/* ---- VQ index entropy coder (vq_vae_gan/model.py:337-576) ---------------------------------------
 * The on-disk format is the reference's: (bytes, shape, probs), bytes = ArithmeticEncoder.encode's output. */
/* calculate_codebook_usage (model.py:540-554): probs[k] = histc(float(indices), bins=k, min=0, max=k-1) / sum.
 * exact_bins == 0 reproduces torch.histc's float32 binning floor(x / (k-1) * k) (at k = 8192 it files index 8190
 * under 8191); exact_bins != 0 counts index i in bin i.  counts_ws: k+1 ints of scratch.  n < 2^24. */
int vg_codebook_usage(const int64_t* indices, int64_t n, int k, int exact_bins, int* counts_ws, float* probs, void* stream);
/* The coder's table (model.py:369-380): probs == NULL: p = 1.0f/k; else p = probs / f32(sum probs) in float32; then
 * cum[0] = 0, cum[i+1] = cum[i] + (double)p[i] in that order.  cum: k+1 doubles. */
int vg_ac_table(const float* probs, int k, double* cum, void* stream);
/* Bytes one stream can need at worst (32 bits per symbol + termination). */
int64_t vg_ac_stream_capacity(int64_t symbols_per_stream);
/* ArithmeticEncoder.encode (model.py:355-445), precision 32, for n_streams independent streams of
 * symbols_per_stream indices each (the reference codes one stream per call: n_streams = 1), all with the table
 * `cum`.  Stream i is written to out + i*cap_per_stream, its length to nbytes[i]; status[i]: 0 ok, 1 a symbol with
 * probability 0 (the reference's loop never terminates on one), 2 capacity exceeded, 3 index outside [0,k),
 * 4 coder state outside the supported range.  One wave per stream. */
int vg_ac_encode(const int64_t* indices, int n_streams, int64_t symbols_per_stream, const double* cum, int k, uint8_t* out,
                 int64_t cap_per_stream, int* nbytes, int* status, void* stream);
/* Inverse of vg_ac_encode (bits past the end of a stream read as 0).  The reference's ArithmeticEncoder.decode
 * (model.py:447-537) does not invert its own encode (it advances its bit pointer with the symbol counter, :491 and
 * :530-532); this is the decoder of the format encode defines. */
int vg_ac_decode(const uint8_t* data, const int* nbytes, int n_streams, int64_t cap_per_stream, int64_t symbols_per_stream,
                 const double* cum, int k, int64_t* indices, int* status, void* stream);

/* ---- input transform (beta_vae_gans/data.py:106-119) ---------------------------------------------
 * One decoded RGB image (uint8, HWC, height*width*3 bytes on the device) and the draws of the reference's random
 * transforms for it: RandomHorizontalFlip, RandomVerticalFlip (data.py:108-109), then RandomResizedCrop's box
 * (i, j, h, w) in the coordinates of the flipped image (data.py:110).  Eval transform (data.py:116-119): no flips,
 * crop = the whole image. */
typedef struct {
  const uint8_t* src;
  int height, width;
  int flip_h, flip_v;
  int crop_i, crop_j, crop_h, crop_w;
} vg_aug_item;
/* Workspace bytes vg_augment_batch needs for these items (0 on invalid items: see vg_last_error). */
size_t vg_augment_ws_bytes(const vg_aug_item* items, int n, int out_h, int out_w);
/* out[n][3][out_h][out_w] f32 = ToTensor(resize(crop(flips(image)), (out_h, out_w), BILINEAR)) with Pillow's 8-bit ImagingResample
 * arithmetic (antialiased triangle filter, 22-bit fixed-point coefficients, horizontal pass rounded to uint8, then
 * vertical), bit for bit; a box of the output's size is the identity, i.e. plain ToTensor (data.py:146-148).
 * `items` is a HOST array (validated on the host: boxes inside images, sizes); images,
 * workspace and output are device memory. */
int vg_augment_batch(const vg_aug_item* items, int n, int out_h, int out_w, void* ws, size_t ws_bytes, float* out, void* stream);

/* Backward of a wide Linear layer y = x W^T + b (beta_vae_gans/model.py:74-75, 87-88 fc_mu / fc_logvar: 131072.. 524288 input
 * features, 128 outputs, a few dozen batch rows; all f32, row-major): one pass over the [fout][fin] matrix, no atomics.
 *   vg_linear_wgrad: dw[o][k] (+= if accumulate) = sum_b gy[b][o] * x[b][k]
 *   vg_linear_dgrad: dx[b][k] = sum_o gy[b][o] * w[o][k]
 * fin must be even; fout * 128 bytes of LDS (fout <= 1280).  The generic pixel-reduction path (vg_conv_wgrad on a 1x1
 * descriptor) stays for every other shape. */
int vg_linear_wgrad(const float* x, const float* gy, float* dw, int b, int fin, int fout, int accumulate, void* stream);
int vg_linear_dgrad(const float* gy, const float* w, float* dx, int b, int fin, int fout, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VG_API_H */
