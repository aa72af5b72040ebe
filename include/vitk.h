/*
 * vitk.h — C ABI of libvitk.so: hand-written gfx950 (MI355X) HIP kernels for the ViT training step.
 *
 * The reference (TaiMingLu/vision_transformers_torch_xla) is 100 % Python and has no native/FFI surface;
 * every entry point below therefore replaces an arithmetic call site of the reference's Python hot path
 * (cited per function as file:line under /root/reference).  The Python boundary that callers see is
 * create_model() / engine.train_one_epoch(); this header is what that host code binds through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing here allocates or frees memory
 *   - bf16 tensors are raw uint16 bit patterns; fp32 are float; all row-major with explicit leading dims
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it, no hidden syncs
 *   - return 0 on success, negative VITK_E* otherwise; vitk_last_error() gives a thread-local message
 *   - shapes are validated on the host before any launch; a rejected call launches nothing
 */
#ifndef VITK_H
#define VITK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITK_VERSION 100

#define VITK_OK 0
#define VITK_EINVAL (-22)   /* bad shape / alignment / null pointer            */
#define VITK_ENOSPC (-28)   /* workspace too small                             */
#define VITK_ELAUNCH (-5)   /* hipGetLastError() reported a launch failure     */

/* GEMM epilogues (vitk_gemm_nt) */
enum {
    VITK_EPI_PLAIN = 0,      /* out_bf16 = acc                                                        */
    VITK_EPI_BIAS = 1,       /* out_bf16 = acc + bias[n]                                              */
    VITK_EPI_BIAS_F32 = 2,   /* out_f32  = acc + bias[n]                                              */
    VITK_EPI_BIAS_GELU = 3,  /* out_bf16 = h = acc + bias[n];  out2_bf16 = gelu_erf(h)                 */
    VITK_EPI_BIAS_RES = 4,   /* out_bf16 = res[m,n] + rowscale[m / rows_per_group] * (acc + bias[n])   */
    VITK_EPI_PATCH = 5,      /* out_bf16[(m/P)*(P+pre)+pre+m%P, n] = acc + bias[n] + pos[pre+m%P, n]   */
    VITK_EPI_DGELU = 6,      /* out_bf16 = acc * gelu_erf'(aux_bf16[m,n])                              */
    VITK_EPI_BIAS_GELU_DG = 7, /* h = acc + bias[n] (fp32);  out_bf16 = gelu_erf'(h); out2_bf16 = gelu_erf(h): the forward of an MLP's
                                first linear that keeps the activation's DERIVATIVE for the backward pass instead of h          */
    VITK_EPI_MUL = 8         /* out_bf16 = acc * aux_bf16[m,n]  (the backward of the above: aux = the stored derivative)       */
};

int vitk_version(void);
const char* vitk_last_error(void);

/* ---- profiling: per-launch hipEvent brackets on the launch stream around the main kernel of each family.
 * No counterpart in the reference (it has no profiling hooks, engine.py:157-195 only keeps wall-clock deltas).
 * Work recorded per launch: floating-point operations 2*M*N*K for the GEMM kinds (MFMA-bound), algorithmic bytes for the
 * others (HBM-bound): LayerNorm forward 4 B/element, backward 6 B/element + 2 per fused residual-gradient read and per
 * scaled copy; attention forward reads qkv and writes out, backward reads qkv, out, dout and writes dqkv; AdamW 30 B per
 * parameter with the bf16 copy-out (28 without), + 8 with EMA. */
#define VITK_PROF_ALL (-1)
#define VITK_PROF_GEMM_NT 0
#define VITK_PROF_GEMM_TN 1
#define VITK_PROF_LN_FWD 2
#define VITK_PROF_LN_BWD 3
#define VITK_PROF_ATTN_FWD 4
#define VITK_PROF_ATTN_BWD 5
#define VITK_PROF_ADAMW 6
int vitk_prof_enable(int on);                                  /* 1: bracket the launches listed above with events */
/* sums over the recorded launches of one kind (or all): waits for their events; records stay until vitk_prof_reset().
 * total_flop carries the work unit of the kind (operations or bytes). */
int vitk_prof_read(int kind, double* total_ms, double* total_flop, int64_t* launches);
int vitk_prof_reset(void);

/* ---- LayerNorm.  Replaces timm LayerNorm = F.layer_norm(x,(D,),w,b,eps=1e-6) used at
 * models/vision_transformer.py:148,163,530,616.  x,y bf16 [M,D]; gamma,beta fp32 [D]; mean,rstd fp32 [M]. D%8==0, D<=1024 */
int vitk_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                       int M, int D, float eps, void* stream);
/* The same over a compact set of samples: output row r (r < M, M = kept samples * rows_per_sample) normalises input row
 * sample_map[r / rows_per_sample] * rows_per_sample + r % rows_per_sample; y, mean and rstd are compact. */
int vitk_layernorm_fwd_gather(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                              int M, int D, float eps, const int* sample_map, int rows_per_sample, void* stream);
/* dx_bf16 = LN'(dy) (+ dres_bf16 if non-null).  dgamma/dbeta fp32 [D]: overwritten (accumulate=0) or added to.
 * workspace: vitk_layernorm_bwd_workspace(M,D) bytes. */
size_t vitk_layernorm_bwd_workspace(int M, int D);
int vitk_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                       const void* dres, void* dx, float* dgamma, float* dbeta, int accumulate,
                       int M, int D, void* workspace, size_t workspace_bytes, void* stream);
/* vitk_layernorm_bwd without a workspace argument and without any allocation.  No counterpart in the reference: it is for
 * callers that cannot allocate device memory around the call - code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  The partial sums of dgamma / dbeta go to a scratch array inside the library (8 MB per GPU, held
 * until the process ends), sized for the largest accepted shape: any M, D <= 1024; anything else is rejected before a launch.
 * Results are bit-identical to vitk_layernorm_bwd.  All calls on one GPU share that GPU's scratch: calls that overlap on one
 * GPU (different streams, or different threads) race; calls on one stream are ordered and safe.  The first call on a device
 * looks the scratch address up (hipGetSymbolAddress): make it before stream capture begins. */
int vitk_layernorm_bwd_noalloc(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                               const void* dres, void* dx, float* dgamma, float* dbeta, int accumulate, int M, int D,
                               void* stream);
/* The same, and also dx_scaled_bf16[row] = rowscale[row / rows_per_group] * dx_bf16[row]: the gradient entering the next
 * residual branch under stochastic depth (timm DropPath in Block.forward, models/vision_transformer.py:164-171), written
 * in the same pass instead of by vitk_rowscale. */
int vitk_layernorm_bwd_scaled(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                              const void* dres, void* dx, const float* rowscale, int rows_per_group, void* dx_scaled,
                              float* dgamma, float* dbeta, int accumulate, int M, int D, void* workspace,
                              size_t workspace_bytes, void* stream);

/* Backward over M FULL rows where dy, mean and rstd are compact over the samples that took the branch: in_inv[sample] is
 * the sample's compact index or -1 (then dx = dres for its rows, and it adds nothing to dgamma / dbeta); with rowscale
 * (indexed by full sample) dx_scaled receives rowscale[sample] * dx compactly for the samples with out_inv[sample] >= 0
 * (out_inv null: all samples, full rows).  in_inv null: dy is full. */
int vitk_layernorm_bwd_gather(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                              const void* dres, void* dx, const float* rowscale, int rows_per_sample, void* dx_scaled,
                              const int* in_inv, const int* out_inv, float* dgamma, float* dbeta, int accumulate, int M,
                              int D, void* workspace, size_t workspace_bytes, void* stream);

/* The second half of any vitk_layernorm_bwd* call on its own.  Those calls sum dgamma / dbeta in two stages: per-block partial
 * rows in the workspace, then a fixed-order reduction.  Called with dgamma = dbeta = NULL they stop after the first stage, and this
 * entry point runs the second one later - on another stream if the caller orders it behind the first call (nothing inside a
 * backward pass reads the LayerNorm parameter gradients, torch autograd only needs them at the optimizer step,
 * engine.py:268-274) - from the same workspace, M and D.  Bit-identical to the one-call form. */
int vitk_layernorm_bwd_reduce(const void* workspace, size_t workspace_bytes, int M, int D, float* dgamma, float* dbeta,
                              int accumulate, void* stream);

/* ---- GEMM, bf16 MFMA, fp32 accumulate.
 * vitk_gemm_nt: C[M,N] = A[M,K] * B[N,K]^T with a fused epilogue.  Replaces nn.Linear forward/dgrad inside timm
 * Attention.qkv/.proj and Mlp.fc1/.fc2 (models/vision_transformer.py:149-159,164-171), the head (:618,990) and the
 * PatchEmbed conv-as-GEMM (:552-560,936).  K%8==0, lda/ldb%8==0, N%4==0, ldc%4==0. */
int vitk_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, int epilogue,
                 const float* bias, const void* res, int ldres, const float* rowscale, int rows_per_group,
                 void* out2, int ldout2, const void* aux, int ldaux, const float* pos, int patches, int prefix,
                 void* stream);
/* vitk_gemm_tn: C[N,K] (fp32) = A[M,N]^T * B[M,K] (weight gradient, contraction over the M rows), split over M into
 * slabs that a deterministic second kernel reduces; optionally also colsum(A) -> dbias[N].  accumulate: C += .
 * N%8==0, K%8==0, lda/ldb%8==0, ldc%4==0.  workspace: vitk_gemm_tn_workspace(M,N,K). */
/* bias + residual epilogue with the output (and residual) rows scattered by sample: GEMM row m, which belongs to compact
 * sample m / rows_per_sample, goes to row sample_map[m / rows_per_sample] * rows_per_sample + m % rows_per_sample of C and res.
 * Stochastic depth (timm DropPath, models/vision_transformer.py:161,173,176-177) multiplies the branch of a dropped sample by 0:
 * with this entry point the branch is computed for the kept samples only (A is compact) and lands in their rows; the rows of
 * the dropped samples are carried over by vitk_copy_samples.  Same results, less work. */
int vitk_gemm_nt_res_scatter(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                             const float* bias, const void* res, int ldres, const float* rowscale, int rows_per_group,
                             const int* sample_map, int rows_per_sample, void* stream);
/* The persistent kernels behind vitk_gemm_nt deal their tiles through counters the library keeps per (device, stream); a kernel
 * leaves them zeroed, and checks at its start that it found them so (a launch that never finished, or a captured graph replayed
 * beside launches of its capture stream, leaves or makes them dirty).  A violation is sticky: every later vitk_gemm_nt call
 * returns VITK_ELAUNCH and names it, until vitk_gemm_nt_sched_reset(stream) zeroes that stream's counters (in stream order) and
 * clears the state.  vitk_gemm_nt_sched_poke writes one of the 16 counter words of the stream, vitk_gemm_nt_sched_state reads them (tests of that check).  No
 * counterpart in the reference: library-internal state of kernels that stand in for torch's addmm (models/vision_transformer.py:149-171). */
int vitk_gemm_nt_sched_reset(void* stream);
int vitk_gemm_nt_sched_poke(void* stream, int word, unsigned value);
/* synchronises with the stream and copies its 16 counter words out (all zero between launches); *reporting: 0 = the kernels cannot
 * report (no host page could be mapped), 1 / 2 = they can.  Returns 1 while a violation is recorded, 0 otherwise, < 0 on error. */
int vitk_gemm_nt_sched_state(void* stream, unsigned* words16, int* reporting);
size_t vitk_gemm_tn_workspace(int M, int N, int K);
int vitk_gemm_tn(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                 float* dbias, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused multi-head self-attention, head dim 64, no mask, no dropout.  Replaces timm Attention.forward's
 * softmax(q k^T / sqrt(d)) v (F.scaled_dot_product_attention) reached from models/vision_transformer.py:176.
 * qkv bf16 [B*N, 3*H*64] (columns: [q|k|v] x head x 64, exactly nn.Linear(D,3D)'s output), out bf16 [B*N, H*64],
 * lse fp32 [B,H,N] = log sum exp of the scaled scores (saved for backward). */
int vitk_attention_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, float scale, void* stream);
/* dqkv bf16 [B*N, 3*H*64] from dout, qkv, out, lse; N <= 608.  No workspace is needed (the size query returns 0 and the
 * workspace arguments are ignored); both stay in the signature so that callers sized for a workspace keep working. */
size_t vitk_attention_bwd_workspace(int B, int N, int H);
int vitk_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                       int B, int N, int H, float scale, void* workspace, size_t workspace_bytes, void* stream);
/* The same for head dims other than 64 - a multiple of 8 up to 96 (48 and 72: my_vit_mini / my_vit_xs, models/my_vit.py:84-165) -
 * and up to 256 tokens: qkv [B*N, 3*H*head_dim], out / dout [B*N, H*head_dim], dqkv like qkv, lse [B, H, N]; no workspace.
 * head_dim 96 up to 224 tokens runs on the MFMA kernels (256-byte LDS rows, one workgroup per compute unit); everything else on
 * vector arithmetic (one thread per row) with the bf16 storage points of the MFMA kernels.  (The Python model pads heads of
 * 48 / 72 columns with zeros to 64 / 96 and so reaches the MFMA kernels for every size of the reference's table.) */
int vitk_attention_generic_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int head_dim, float scale, void* stream);
int vitk_attention_generic_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int N, int H,
                               int head_dim, float scale, void* stream);

/* ---- patch embedding front end / pooling tail.
 * im2col: NCHW fp32 image -> bf16 patch matrix [B*gh*gw, C*p*p], k = c*p*p + ky*p + kx, patch index gy*gw+gx
 * (the Conv2d(k=s=p) + flatten(2).transpose(1,2) of timm PatchEmbed, models/vision_transformer.py:552-560,936). */
int vitk_im2col_patches(const float* img, void* patches, int B, int C, int H, int W, int p, void* stream);
/* x[b, t, :] = prefix_tokens[t, :] + pos[t, :] for t < prefix  (models/vision_transformer.py:761-778) */
int vitk_prefix_tokens_fwd(const float* tokens, const float* pos, void* x, int B, int N, int D, int prefix, void* stream);
/* dpos[t,:] (+)= sum_b dx[b,t,:] for all t;  dtokens[t,:] (+)= sum_b dx[b,t,:] for t<prefix */
int vitk_pos_tokens_bwd(const void* dx, float* dpos, float* dtokens, int accumulate, int B, int N, int D, int prefix,
                        void* stream);
/* y[b,:] = mean_{t=first..N-1} x[b,t,:]  (global_pool_nlc 'avg', models/vision_transformer.py:431-433) */
int vitk_pool_avg_fwd(const void* x, void* y, int B, int N, int D, int first, void* stream);
/* dx[b,t,:] = dy[b,:]/(N-first) for t>=first else 0 */
int vitk_pool_avg_bwd(const void* dy, void* dx, int B, int N, int D, int first, void* stream);
/* gather / scatter of single token rows (pool 'token', models/vision_transformer.py:428-429; DeiT heads deit.py:108-119) */
int vitk_gather_rows(const void* x, void* y, int B, int N, int D, int token, void* stream);
int vitk_scatter_rows(const void* dy, void* dx, int B, int N, int D, int token, int zero_others, void* stream);

/* y[b, t-first, :] = x[b, t, :] for t >= first: the patch-token rows without the prefix tokens (operand of the
 * patch-embedding weight gradient; inverse of the torch.cat at models/vision_transformer.py:770-778) */
int vitk_slice_tokens(const void* x, void* y, int B, int N, int D, int first, void* stream);

/* dst[s] = src[s] for the whole samples s in sample_list (device array of `count` ints), bf16 [samples, elems_per_sample] */
int vitk_copy_samples(const void* src, void* dst, const int* sample_list, int count, int64_t elems_per_sample, void* stream);
/* dst (device) = pinned_src (page-locked host memory, device-accessible) as an ordinary launch on the stream: the small
 * per-step tables of the host-side DropPath draws (models/vision_transformer.py:161,173) reach the GPU without a
 * hipMemcpyAsync call.  bytes % 16 == 0; the host buffer must stay untouched until the launch has run. */
int vitk_copy_from_host(const void* pinned_src, void* dst, int64_t bytes, void* stream);

/* ---- elementwise helpers */
int vitk_rowscale(const void* x, const float* rowscale, int rows_per_group, void* y, int M, int D, void* stream);
/* the same, compact over nk samples: y[c*rpg + t] = rowscale[sample_map[c]] * x[sample_map[c]*rpg + t] */
int vitk_rowscale_gather(const void* x, const float* rowscale, int rows_per_group, const int* sample_map, int nk, void* y,
                         int D, void* stream);
int vitk_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int vitk_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
int vitk_transpose_bf16(const void* src, void* dst, int rows, int cols, void* stream);
/* The same for n matrices in ONE launch (the per-step refresh of all transposed weight copies).  desc is a DEVICE array of
 * n+1 records of five int64: {src address, dst address, rows, cols, first_tile}, first_tile = number of 64x64 tiles of the
 * records before it; record n only carries first_tile = total_tiles. */
int vitk_transpose_bf16_batched(const void* desc, int n, int total_tiles, void* stream);
int vitk_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream);

/* ---- device tail of the input pipeline: src = B decoded images in host layout [B, src_h, src_w, 3] (uint8 when src_is_u8,
 * else fp32), already on the device; dst = fp32 NCHW [B, 3, H, W], the window of H x W pixels at (top, left) of every image.
 * Per element, each step rounded on its own: flip_lr for the samples with flip[b] != 0 (flip: device uint8 [B] or NULL;
 * big_vision/pp/ops_image.py:69-82), value_range(vmin, vmax) = float(u)/255 for integers, then v*(vmax-vmin)+vmin
 * (big_vision/pp/ops_general.py:26-43), HWC -> CHW (datasets.py:301-307) and, when mean3/std3 (HOST arrays of 3 floats, or
 * both NULL) are given, (v - mean[c]) / std[c] (datasets.py:249-252).  W must be a multiple of 4. */
int vitk_preprocess_images(const void* src, int src_is_u8, const unsigned char* flip, float* dst, int B, int H, int W,
                           int src_h, int src_w, int top, int left, float vmin, float vmax, const float* mean3,
                           const float* std3, void* stream);
/* The same chain writing what the model's first GEMM reads instead of the fp32 NCHW batch: dst = the bf16 patch matrix
 * [B * (H/patch) * (W/patch), 3 * patch * patch], row = image b, patch gy * (W/patch) + gx, column = c * patch^2 + ky * patch + kx
 * (the operand order of Conv2d(3, D, k = s = patch) + flatten(2).transpose(1, 2), models/vision_transformer.py:552-560,936), each
 * value rounded to bf16 once.  Bit-identical to vitk_im2col_patches(vitk_preprocess_images(src)): the NHWC window goes straight to
 * the patch matrix, without the fp32 NCHW batch written and read in between (datasets.py:301-314 hands the reference's model CHW
 * float32).  patch % 8 == 0, H % patch == W % patch == 0. */
int vitk_preprocess_patches(const void* src, int src_is_u8, const unsigned char* flip, void* dst, int patch, int B, int H, int W,
                            int src_h, int src_w, int top, int left, float vmin, float vmax, const float* mean3,
                            const float* std3, void* stream);

/* ---- crop / resize stage of the input pipeline on the device: B decoded images of DIFFERENT sizes (uint8 HWC, 3 channels, packed
 * one after the other in src; already on the device) become the fp32 NCHW batch dst [B, 3, S, S].  Per image a descriptor says
 * which box of the source is resized to which size and which S x S window of the result is kept:
 *   training   decode_jpeg_and_inception_crop(S) (big_vision/pp/ops_image.py:36-66): box = the sampled crop, out = S x S, win = (0, 0)
 *   evaluation resize_small(R) | central_crop(S) (:85-134): box = the whole image, out = (new_h, new_w), win = the central offsets
 *              (negative where the resized image is smaller than S: zero padding, as tf.image.resize_with_crop_or_pad)
 * Arithmetic: u8 -> float32 * (1/255) (tf.image.convert_image_dtype), tf.image.resize(method, antialias=True) as the
 * ScaleAndTranslate op computes it - rows first, then columns, float32, weights kernel(|src + 0.5 - sample| / kernel_scale)
 * normalised per output index, kernel_scale = max(in / out, 1); kernel 0 = 'triangle' (BILINEAR), 1 = 'keyscubic' (BICUBIC) -
 * then flip_lr (desc.flip), value_range, the layout change and the normalisation exactly as vitk_preprocess_images.
 * The B descriptors are passed twice: desc_host (HOST memory) is validated before anything is launched - boxes inside their
 * images, images inside the src_bytes of src, box width <= 12 288, at most 512 vertical filter taps (a 60-fold reduction with the
 * cubic kernel) - and desc_dev is the same array on the device, read by the kernel.  S a multiple of 4, <= 1024. */
typedef struct {
    int64_t src_offset;            /* byte offset of the image's first pixel in src; rows are src_w * 3 bytes */
    int32_t src_h, src_w;
    int32_t box_y, box_x, box_h, box_w;
    int32_t out_h, out_w;
    int32_t win_y, win_x;
    int32_t flip, reserved;
} vitk_resample_desc;              /* 56 bytes */
int vitk_resample_images(const unsigned char* src, int64_t src_bytes, const vitk_resample_desc* desc_host,
                         const vitk_resample_desc* desc_dev, float* dst, int B, int S, int kernel, float vmin, float vmax,
                         const float* mean3, const float* std3, void* stream);
/* the same writing the model's bf16 patch matrix [B * (S/patch)^2, 3 * patch^2] (layout: vitk_preprocess_patches) instead of the
 * fp32 NCHW batch: bit-identical to vitk_im2col_patches(vitk_resample_images(...)).  Both entry points validate desc_host and
 * read desc_dev on the device; the kernel clamps every source offset it derives from desc_dev to the source buffer, so a device
 * copy that does not match the host one gives wrong pixels, never an access outside src. */
int vitk_resample_patches(const unsigned char* src, int64_t src_bytes, const vitk_resample_desc* desc_host,
                          const vitk_resample_desc* desc_dev, void* patches, int patch, int B, int S, int kernel, float vmin,
                          float vmax, const float* mean3, const float* std3, void* stream);

/* ---- mixup / cutmix in 'batch' mode on the device, in place on the fp32 NCHW batch x [B, per_sample]; sample b is paired
 * with sample B-1-b (x.flip(0)); B must be even.  Replaces timm.data.Mixup.__call__ as the reference applies it before
 * every training step (main.py:622-629, engine.py:132-133, 259-260).
 * mixup : x = x*lam + x.flip(0)*(1-lam), both products rounded before the sum (the reference's mul_ / add_ order). */
int vitk_mixup_batch(float* x, int B, int64_t per_sample, float lam, float one_minus_lam, void* stream);
/* cutmix: x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh] (the two samples of a pair swap the box) */
int vitk_cutmix_batch(float* x, int B, int C, int H, int W, int yl, int yh, int xl, int xh, void* stream);
/* soft targets fp32 [B,C]: one_hot(target, on, off)*lam + one_hot(target.flip(0), on, off)*(1-lam) (timm mixup_target) */
int vitk_mixup_targets(const int64_t* target, float* out, int ldo, int B, int C, float lam, float one_minus_lam,
                       float on_value, float off_value, void* stream);
/* timm.data.Mixup modes 'elem' and 'pair' (Mixup._mix_elem / _mix_pair; --mixup_mode is forwarded at main.py:229, 628): one
 * lambda, kind and box per sample.  Sample b is mixed with the ORIGINAL of sample B-1-b, in place (both samples of a pair are
 * read before either is written).  params_host is validated before the launch, the kernel reads the caller's device copy. */
typedef struct vitk_mixup_sample {
    int kind;              /* 0 untouched (lambda = 1), 1 mixup, 2 cutmix */
    float lam;             /* mixup: weight of the sample itself */
    float one_minus_lam;   /* mixup: weight of its partner (formed by the caller, in the precision the reference forms it) */
    int yl, yh, xl, xh;    /* cutmix: rows [yl, yh) and columns [xl, xh) are taken from the partner */
    int reserved;
} vitk_mixup_sample;
int vitk_mixup_elem(float* x, int B, int C, int H, int W, const vitk_mixup_sample* params_host, const vitk_mixup_sample* params_dev,
                    void* stream);
/* soft targets with one lambda per sample (fp32 [B] on the device): row b = one_hot(target[b])*lam[b] + one_hot(target[B-1-b])*(1-lam[b]) */
int vitk_mixup_targets_elem(const int64_t* target, float* out, int ldo, int B, int C, const float* lam_rows, float on_value,
                            float off_value, void* stream);

/* ---- losses on fp32 logits [B,C] with leading dim ld.  loss_rows fp32 [B] per-sample loss, dlogits fp32 [B,C]
 * already scaled by grad_scale (= upstream grad / B for 'mean').
 * Hard labels with label smoothing eps (eps=0: nn.CrossEntropyLoss; eps>0: timm LabelSmoothingCrossEntropy,
 * main.py:926-935):  loss = (1-eps)*nll + eps*mean_c(-logp). */
int vitk_xent_label_fwd_bwd(const float* logits, int ld, const int64_t* target, float smoothing, float* loss_rows,
                            float* dlogits, int lddl, float grad_scale, int B, int C, void* stream);
/* Soft targets (timm SoftTargetCrossEntropy): loss = sum_c -t*logp */
int vitk_xent_soft_fwd_bwd(const float* logits, int ld, const float* target, int ldt, float* loss_rows,
                           float* dlogits, int lddl, float grad_scale, int B, int C, void* stream);
/* KD term T^2 * KL(softmax(t/T) || log_softmax(s/T)) per row ('batchmean' divides by B on the host side),
 * main.py:939-970 / test_kd.py:43-88 */
int vitk_kd_kl_fwd_bwd(const float* student, int lds_, const float* teacher, int ldt, float temperature,
                       float* loss_rows, float* dstudent, int ldd, float grad_scale, int B, int C, void* stream);
/* argmax over classes, first maximal index (torch.max semantics), engine.py:283 */
int vitk_argmax_rows(const float* logits, int ld, int64_t* idx, int B, int C, void* stream);

/* ---- fused AdamW over a flat fp32 parameter range (torch.optim.AdamW semantics, optim_factory.py:248-249),
 * with bf16 copy-out of the updated weights and optional EMA (timm ModelEma, engine.py:198-202,273-274).
 * p,g,m,v fp32 [n]; p_bf16 may be null; ema may be null. step is the 1-based step count.  Hyper-parameters are doubles,
 * as the Python floats of the reference are: 1-beta, bias corrections and lr*weight_decay are formed in double. */
int vitk_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, float* ema, double ema_decay,
                    int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                    float grad_scale, void* stream);
/* ema = decay*ema + (1-decay)*p over a flat fp32 range (timm ModelEma.update, engine.py:198-202,273-274) */
int vitk_ema_update(float* ema, const float* p, double decay, int64_t n, void* stream);
/* g *= min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) with sumsq[0] read on the device (torch clip_grad_norm_, engine.py:175-177) */
int vitk_clip_grad_scale(float* g, int64_t n, const float* sumsq, float max_norm, void* stream);
/* sum of squares of a flat fp32 range -> out[0] (+)=, for clip_grad_norm_ (engine.py:175-177) */
int vitk_sumsq(const float* g, int64_t n, float* out, int accumulate, void* workspace, size_t workspace_bytes,
               void* stream);
size_t vitk_sumsq_workspace(int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* VITK_H */
