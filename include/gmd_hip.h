/*
 * gmd_hip.h -- C ABI of libgmd_hip.so: the MI355X (gfx950 / CDNA4) kernels behind
 * GM-Diffusion's Stage-3 hot path (SDR+GM denoising loop, VAE decode, gain-map HDR
 * recomposition).
 *
 * The reference (Guanys-dar/GM-Diffusion) is pure Python with NO FFI / plugin / C
 * interface: the heavy arithmetic is whatever torch/cuDNN/diffusers
 * dispatch.  This ABI is therefore new; each entry point cites the reference
 * expression (file:line, relative to the reference root) whose arithmetic it
 * replaces.  The Python host side (gm-diffusion_amd/gm_diffusion) binds it with
 * ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named *_host; no allocation inside;
 *     the caller owns all buffers (workspace sizes documented per call);
 *   - every call is stream-ordered on `stream` (a hipStream_t passed as void*) and
 *     never synchronises, so calls are hipGraph-capturable;
 *   - return value: GMD_OK or an error code; gmd_last_error() returns a thread-local
 *     message for the last failing call;
 *   - activations are channels-last: [B, H*W, C] ("NHWC"); `dtype` selects the
 *     activation/weight element type (GMD_BF16 / GMD_F16 = MFMA path, GMD_F32 = parity path);
 *     biases, norm affine parameters and statistics are always float32.
 */
#ifndef GMD_HIP_H
#define GMD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMD_ABI_VERSION 11
#define GMD_WS_TAIL_BYTES 65536 /* see "WORKSPACE CONTRACT" at gmd_gemm_nt */

#define GMD_OK 0
#define GMD_ERR_INVALID 1     /* bad argument (shape / alignment / null) */
#define GMD_ERR_LAUNCH 2      /* hipLaunch / runtime failure */
#define GMD_ERR_UNSUPPORTED 3 /* valid request this build does not implement */

#define GMD_F32 0
#define GMD_BF16 1
#define GMD_F16 2  /* IEEE half: same kernels and layouts as GMD_BF16, float16 elements / MFMA forms */
/* float32 tensors whose contraction runs on the matrix cores as three float16 products (x = hi + lo, hi = f16(x),
 * lo = f16(x - hi); a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with float32 accumulation: ~2^-22 relative per term, the order
 * of float32 rounding; operands must stay inside the float16 range, |x| <= 65504).  Accepted by gmd_gemm_nt, gmd_conv3x3
 * and gmd_attention only; every other entry point takes the same tensors as GMD_F32.  The reference runs its dual-UNet
 * pipeline in float32 (scripts/inference/experiments/formal_improved.py:199): this is that precision at matrix-core speed. */
#define GMD_F32S 3  /* both operands plain float32, split in the kernel */
#define GMD_F32SW 4 /* W operand pre-split by gmd_split_weights (weights: split once when a model is placed on the device) */
/* Round 4: float32 ACTIVATIONS stored pre-split -- every 32-element chunk of a row (128 bytes, the bytes of the 32 float32 values) holds
 * [hi 64 B | lo 64 B] in the order of gmd_split_weights.  As `dtype` of gmd_gemm_nt / gmd_conv3x3(_groupnorm): the A operand AND W are
 * pre-split (no conversion instruction in the main loop; results bit-identical to GMD_F32SW on the plain tensor).  As `out_dtype` of
 * gmd_gemm_nt (GEGLU and plain row epilogues of the split types) and as `dtype` of gmd_layernorm / gmd_groupnorm_colstats /
 * gmd_groupnorm_split / gmd_attention (there: Q, K, V^T plain as for GMD_F32S): float32 tensors in, the OUTPUT stored pre-split (rows
 * must be multiples of 32 elements, the buffer contiguous).  Such a tensor is
 * only ever an A (or W) operand of a contraction: nothing else reads it.  A pre-split operand (A under GMD_F32SA, W under GMD_F32SW /
 * GMD_F32SA) must start on a 128-byte boundary and have its leading dimension and batch stride in whole chunks (multiples of 32
 * elements): a column-offset view or a ragged leading dimension would be read as the wrong halves -- refused with GMD_ERR_INVALID. */
#define GMD_F32SA 5

/* epilogue activation for gmd_gemm_nt */
#define GMD_ACT_NONE 0
#define GMD_ACT_SILU 1
#define GMD_ACT_GEGLU 2 /* bf16 only: W rows interleaved in 16-row [value|gate] groups; writes [M, N/2] = h * gelu_erf(g) */
#define GMD_ACT_QUICK_GELU 3 /* x * sigmoid(1.702 x): CLIP text encoder MLP */

typedef void* gmd_stream_t;

int gmd_abi_version(void);
const char* gmd_last_error(void);

/* ------------------------------------------------------------------------------------
 * HDR tail
 * ---------------------------------------------------------------------------------- */

/* Fused Stage-3 tail, one pass over the two decoded images:
 *   sdr = clamp(sdr_dec/2+0.5, 0, 1), gm likewise      scripts/inference/generate_hdr.py:227-228,232-233
 *   u8  = (x*255) truncated                             generate_hdr.py:244-245
 *   hdr = (sdr^2.2 + eps)*(1 + gm*qmax) - eps           gm_diffusion/stage1/tone_mapping.py:68-70,
 *                                                       scripts/inference/experiments/formal_improved.py:34-45
 *   [clamp(hdr, 0, qmax+1) when flags&1]                tone_mapping.py:71
 *   hdr_file = hdr/(qmax+1)                             generate_hdr.py:27-29
 *   hdr_u16  = rint(clamp(hdr_file*65535, 0, 65535))    gm_diffusion/stage1/augmentations.py:38-41
 * Inputs: decoder outputs [B,3,H,W] (in_layout 0) or [B,H,W,3] (in_layout 1), in_dtype F32/BF16.
 * Outputs are [B,H,W,3]; any output pointer may be NULL. */
int gmd_hdr_tail(const void* sdr_dec, const void* gm_dec, int in_dtype, int in_layout,
                 int B, int H, int W, float qmax, float eps, int flags,
                 float* sdr_img, float* gm_img, uint8_t* sdr_u8, uint8_t* gm_u8,
                 float* hdr, float* hdr_file, uint16_t* hdr_u16, gmd_stream_t stream);

/* tone_mapping.py:60-71 apply_gm_to_sdr (clamp!=0) / formal_improved.py:34-45 (clamp==0); any shape, n elements */
int gmd_apply_gm_to_sdr(const float* gm, const float* sdr, float* out, int64_t n,
                        float qmax, float eps, int clamp, gmd_stream_t stream);
/* kind 0: linear_scale_tmo tone_mapping.py:14-18; 1: hard_clip_tmo :21-26;
 * 2: mu-log (fix_mulog_tmo :29-36 with mu=500, random_tmo_cuda :50-57 with the drawn mu);
 * 3: tmo_cuda :39-47 (mu=5000, /10 pre-scale);
 * 4: decode post-process clamp(x/2+0.5, 0, 1), gm_diffusion/pipelines/stable_diffusion_gm.py:606;
 * 5: x*mu (1/scaling_factor*latents, generate_hdr.py:225); 6: x/mu (latents/scaling_factor, stable_diffusion_gm.py:1094) */
int gmd_tmo(const float* in, float* out, int64_t n, int kind, float qmax, float mu, gmd_stream_t stream);
/* tone_mapping.py:74-90 gamut_compress on [B,3,HW] planar input */
int gmd_gamut_compress(const float* in, float* out, int B, int64_t HW, gmd_stream_t stream);
/* scripts/stage1/train_vqgan_lora.py:1133-1141: apply_gm_to_sdr(clamped) -> fix_mulog_tmo -> gamut_compress, planar [B,3,HW] */
int gmd_stage1_chain(const float* gm, const float* sdr, float* out, int B, int64_t HW, float qmax, gmd_stream_t stream);
/* augmentations.py:38-41 discretize_to_uint16; out_float and/or out_codes may be NULL */
int gmd_discretize_u16(const float* in, float* out_float, uint16_t* out_codes, int64_t n, gmd_stream_t stream);
/* Radiance RGBE pixels of float RGB [npix,3] -> [npix,4] bytes (Ward's float2rgbe, the encoder behind
 * cv2.imwrite("*.hdr"), scripts/inference/generate_hdr.py:27-30); negative components are stored as 0 */
int gmd_rgbe_encode(const float* rgb, uint8_t* out, int64_t npix, gmd_stream_t stream);
/* HOST function (no device work): run-length framing of RGBE scanlines for a Radiance .hdr file, the form OpenCV's encoder
 * behind cv2.imwrite("*.hdr") (generate_hdr.py:27-30) writes by default.  rgbe: host [H][W][4] bytes from gmd_rgbe_encode;
 * out: host buffer of at least gmd_rgbe_rle_bound(H, W) bytes; *out_bytes = bytes to write after the header.  Widths < 8 or
 * > 32767 come back flat, as the format prescribes. */
int64_t gmd_rgbe_rle_bound(int H, int W);
int gmd_rgbe_rle_encode(const uint8_t* rgbe, int H, int W, uint8_t* out, int64_t capacity, int64_t* out_bytes);
/* generate_hdr.py:244-245 (x*255).astype(uint8) */
int gmd_quantize_u8(const float* in, uint8_t* out, int64_t n, gmd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * latent-side fused step
 * ---------------------------------------------------------------------------------- */

/* One pass over a [B,4,h,w] fp32 latent: classifier-free-guidance combine
 * (stable_diffusion_dual_unet.py:1063-1065), optional per-sample guidance rescale factor
 * (:1067-1069, rescale_noise_cfg :71-94; `rescale_ratio` = std_text/std_cfg per sample from
 * gmd_cfg_std_ratio), x0 prediction (:1071-1075) and the PNDM/PLMS linear-multistep update
 * (diffusers PNDMScheduler.step_plms/_get_prev_sample, called at :1077 and :1093).
 *   mode 0: eps' = eps                        (first step)
 *   mode 1: eps' = (eps + e1)/2, sample = cur_sample   (PLMS counter==1 redo step)
 *   mode 2: eps' = (3 eps - e1)/2
 *   mode 3: eps' = (23 eps - 16 e1 + 5 e2)/12
 *   mode 4: eps' = (55 eps - 59 e1 + 37 e2 - 9 e3)/24
 *   x_prev = sample_coeff*sample - (alpha_delta*eps')/denom
 * eps_in: [2B,4,h,w] (uncond half first) when do_cfg else [B,4,h,w].
 * eps_out (the guided eps, appended to the scheduler history by the host), x_prev, x0 may alias nothing.
 * Arithmetic is float32 without FMA contraction, in torch's operation order: results are
 * bit-identical to the CPU reference given identical eps. */
int gmd_latent_step(const float* eps_in, const float* x, const float* cur_sample,
                    const float* e1, const float* e2, const float* e3,
                    int B, int64_t chw, int do_cfg, float guidance_scale,
                    const float* rescale_ratio, float guidance_rescale,
                    int mode, float sample_coeff, float alpha_delta, float denom,
                    float sqrt_alpha, float sqrt_one_minus_alpha,
                    float* eps_out, float* x_prev, float* x0, gmd_stream_t stream);

/* DPM-Solver++ multistep (dpmsolver++ / midpoint / epsilon, orders 1-2) -- the scheduler the reference swaps in for the
 * dual-UNet runs (scripts/inference/experiments/formal_improved.py:195) -- fused with the same CFG combine / rescale and
 * pipeline x0 as gmd_latent_step:
 *   m0 = (x - sigma_s0*eps)/alpha_s0;  x_prev = c_x*x - c_m*m0 [ - c_h*(inv_r0*(m0 - m1)) when order == 2 ]
 * c_x = sigma_t/sigma_s0, c_m = alpha_t*(exp(-h)-1), c_h = 0.5*c_m, inv_r0 = 1/r0: float32 scalars computed by the host
 * exactly as diffusers computes its 0-dim tensors.  m0_out is the x0 prediction the host keeps for the next step. */
int gmd_dpm_step(const float* eps_in, const float* x, const float* m1, int B, int64_t chw,
                 int do_cfg, float guidance_scale, const float* rescale_ratio, float guidance_rescale,
                 int order, float sigma_s0, float alpha_s0, float c_x, float c_m, float c_h, float inv_r0,
                 float sqrt_alpha, float sqrt_one_minus_alpha,
                 float* m0_out, float* x_prev, float* x0, gmd_stream_t stream);

/* DDPM ancestral step -- the scheduler the reference's Stage-3 CLI constructs (scripts/inference/generate_hdr.py:162,
 * used by the pipeline call at :212-218) -- fused with the same CFG combine / rescale and pipeline x0 as gmd_latent_step,
 * in the float32 operation order of diffusers' DDPMScheduler.step:
 *   p0 = (x - sched_sqrt_one_minus_alpha*eps)/sched_sqrt_alpha [clamp +-clip_range];  x_prev = x0_coeff*p0 + xt_coeff*x
 *   [ + noise_scale*noise when noise != NULL (t > 0) ].
 * `noise` is drawn by the host scheduler from the caller's generator, so a generator shared by the two schedulers of the
 * dual pipeline is consumed SDR first, GM second (stable_diffusion_dual_unet.py:1015, 1077, 1093). */
int gmd_ddpm_step(const float* eps_in, const float* x, const float* noise, int B, int64_t chw,
                  int do_cfg, float guidance_scale, const float* rescale_ratio, float guidance_rescale,
                  float sched_sqrt_alpha, float sched_sqrt_one_minus_alpha, int clip_sample, float clip_range,
                  float x0_coeff, float xt_coeff, float noise_scale, float sqrt_alpha, float sqrt_one_minus_alpha,
                  float* x_prev, float* x0, gmd_stream_t stream);

/* per-sample unbiased std of the text eps and of the guided eps -> ratio[b] = std_text/std_cfg
 * (rescale_noise_cfg, stable_diffusion_dual_unet.py:88-91) */
int gmd_cfg_std_ratio(const float* eps_in, int B, int64_t chw, float guidance_scale,
                      float* ratio, gmd_stream_t stream);

/* 8-channel concat + CFG duplicate + NCHW->NHWC + cast + zero channel padding in one pass
 * (stable_diffusion_gm.py:1045-1047 cat([sdr_latent, latents],1) then cat([.]*2);
 *  stable_diffusion_dual_unet.py:1045, 1080).  src0 [B,C0,HW] f32, src1 [B,C1,HW] f32 or NULL;
 * out [dup*B, HW, CP] of out_dtype with channels >= C0+C1 zeroed. */
int gmd_pack_unet_input(const float* src0, int C0, const float* src1, int C1, int B, int64_t HW,
                        int dup, void* out, int CP, int out_dtype, gmd_stream_t stream);
/* [B,HW,ld] (first C channels) of in_dtype -> [B,C,HW] float32 */
int gmd_unpack_nchw(const void* in, int in_dtype, int64_t ld, int B, int C, int64_t HW,
                    float* out, gmd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * UNet / VAE building blocks (arithmetic lives in the un-vendored
 * `diffusers` dependency: UNet2DConditionModel / AutoencoderKL called at
 * stable_diffusion_gm.py:1051,1094 and stable_diffusion_dual_unet.py:1052,1083)
 * ---------------------------------------------------------------------------------- */

/* DEBUG ONLY -- kernel-tuning hook of tools/ (bench_gemm.py, check_ring.py, bench_graph_ops.py); never called by the product
 * path and refused (GMD_ERR_UNSUPPORTED) unless the process has GMD_TUNING=1 in its environment.  Pins the tile (bm x bn), the
 * operand pipeline (pf: 9 = LDS-DMA, 1/2 = register staged, 1xx = ring variants) and the split-K factor of every later 16-bit
 * gmd_gemm_nt / gmd_conv3x3 launch of this process; 0 keeps the heuristic for that field, (0,0,0,0) restores it (always
 * allowed).  GMD_GEMM_FORCE="bm,bn,pf,ksplit" seeds the same override when the library is loaded, under the same GMD_TUNING
 * gate.  A forced plan whose kernel lacks the epilogue a launch asks for (fused GEGLU, column statistics) is refused at the
 * launch, never run. */
int gmd_gemm_plan_override(int bm, int bn, int pf, int ksplit);

/* Plan family of the CALLING THREAD's later 16-bit gmd_gemm_nt / gmd_conv3x3 / gmd_gemm_qkv_vt launches (and of the plan queries
 * gmd_gemm_colstats_plan / gmd_gemm_plan_info / gmd_gemm_qkv_vt_ok, which must agree with them): 0 (the default) = the plan that is
 * fastest when the launch has the chip to itself; 1 = the co-running family (256-row tiles, filled up with K slices: fewest L2 -> LDS
 * bytes per product) for launches that share the chip with a second stream's kernels -- what the dual-UNet pipeline selects around
 * its two overlapped forwards (stable_diffusion_dual_unet.py:1040-1093: the two UNet calls of one loop iteration).  Results of the
 * two families agree to float32 summation order (K slices).  Returns the previous family; any other argument only queries.  Thread-
 * local, so concurrent host threads do not see each other's choice.  GMD_PP=b / GMD_PP=1 in the environment pin 1 / 0 process-wide. */
int gmd_gemm_plan_family(int family);

/* Split-K launches of the 16-bit gmd_gemm_nt / gmd_conv3x3 with up to `max_slices` K slices reduce INSIDE the kernel (the last
 * slice of a tile adds the others' accumulator fragments in slice order and runs the fused epilogue: no partial-sum slabs, no
 * reduction launch); more slices, and launches whose consumer reads the slabs itself (gmd_conv3x3_groupnorm), keep the slab path.
 * Both paths add the same partial sums in the same order: results are bit-identical.  Default 4 (GMD_SPLITK_FIXUP=<n> in the
 * environment seeds it; 0 = slab path everywhere).  Returns the previous value; a negative argument only queries.  Process-wide:
 * change it only while no launch is in flight (tests, A/B measurements). */
int gmd_splitk_fixup_max(int max_slices);

/* C[b] = act(alpha * A[b] @ W[b]^T + bias + rowbias + residual).
 * A: [M,K] ld lda; W: [N,K] ld ldw (both K-contiguous); C: [M,N] ld ldc, out_dtype F32 or `dtype`.
 * bias: float32 [N] or NULL.  rowbias: float32 [ceil(M/rows_per_group), ldrb] (ldrb >= N; 0 means N) or NULL, added to rows
 * of group m/rows_per_group (ResnetBlock2D time-embedding add).  residual: `dtype` [M,N] ld ldr or NULL.
 * Requirements: K % 64 == 0 (BF16 / F16), K % 32 == 0 (F32S / F32SW), K % 4 == 0 (F32); lda, ldw multiples of 8 (16-bit) /
 * 4 (float32) elements (ldw of a pre-split W counts k's, as for the plain matrix);
 * base pointers 16-byte aligned.  nn.Linear / conv1x1 / attention score products.
 * workspace (optional, float32 scratch of workspace_bytes): lets launches that cannot fill the chip split K
 * (deterministic: partial sums are added in slice order, no floating-point atomics); with NULL / too small a workspace K is not split.
 * WORKSPACE CONTRACT (ABI v11; every entry point that takes a workspace): the last GMD_WS_TAIL_BYTES of the buffer are reserved for
 * the library's split-K arrival counters -- they must be ZERO when a buffer is first handed to the library (memset it once after
 * allocating it) and every launch leaves them zero; only workspace_bytes - GMD_WS_TAIL_BYTES are used for partial sums, and the plan
 * queries (gmd_gemm_colstats_plan, gmd_gemm_plan_info, gmd_gemm_qkv_vt_ok, gmd_gemm_out_split_ok, gmd_conv3x3_gn_fusable) take the
 * same workspace_bytes the launch will get.  One workspace serves ONE stream (or one captured graph) at a time: launches that may
 * run concurrently need workspaces of their own. */
int gmd_gemm_nt(const void* A, const void* W, void* C, int dtype, int out_dtype,
                int M, int N, int K, int64_t lda, int64_t ldw, int64_t ldc,
                int batch, int64_t strideA, int64_t strideW, int64_t strideC,
                const float* bias, const float* rowbias, int rows_per_group, int64_t ldrb,
                const void* residual, int64_t ldr, int64_t strideR,
                float alpha, int act, float* colstats, int colstats_bucket,
                void* workspace, int64_t workspace_bytes, gmd_stream_t stream);

/* Pre-split layout of a float32 weight matrix W [N,K] (row stride ldw, K % 32 == 0) for GMD_F32SW: `out` holds N*K*4 bytes,
 * per row n and block of 32 k's 64 bytes of float16 hi followed by 64 bytes of float16 lo, each as four 16-byte chunks
 * q = 0..3 holding k = {4q..4q+3, 16+4q..16+4q+3} of the block (the k's one MFMA lane group consumes).  For a convolution
 * weight [Cout, 9*Cin] (k = tap*Cin + c, Cin % 32 == 0) the same call applies with K = 9*Cin. */
int gmd_split_weights(const float* W, void* out, int64_t N, int64_t K, int64_t ldw, gmd_stream_t stream);

/* Column statistics for a following GroupNorm (diffusers GroupNorm over a ResnetBlock2D / Transformer2DModel input: the
 * statistics pass is folded into the epilogue of the launch that produces the tensor).  With `colstats` non-NULL,
 * gmd_gemm_nt / gmd_conv3x3 also write float32 colstats[M/64][N/colstats_bucket][2] = {sum, sum of squares} of the STORED
 * (rounded) outputs over each block of 64 rows and each bucket of `colstats_bucket` adjacent columns, in a fixed order.
 * Only launches that run the full-tile row epilogue of the 128-row ring kernels can do this (16-bit types, batch 1, no
 * split-K, M % 128 == 0, N a multiple of the 160- or 128-column tile, bucket dividing half a tile): this returns 1 when a
 * launch of these dimensions will, 0 otherwise (asking for colstats then fails with GMD_ERR_UNSUPPORTED).  For a
 * convolution pass M = B*Hout*Wout, N = Cout, K = 9*Cin. */
int gmd_gemm_colstats_plan(int dtype, int M, int N, int K, int batch, int64_t workspace_bytes, int bucket);

/* Which kernel a 16-bit gmd_gemm_nt / gmd_conv3x3 launch of these dimensions takes (for a convolution M = B*Hout*Wout, N = Cout,
 * K = 9*Cin; `geglu` = 1 for GMD_ACT_GEGLU launches): out4 = {tile rows, tile columns, kernel code, K slices}.  Kernel codes:
 * 0 = the LDS-DMA ring kernels (two 4-wave workgroups per CU; 64x64 tiles for under-filled launches), 283 = the ping-pong kernel
 * (one workgroup of 8 consumer + 4 loader waves on a 256-row tile, round 4), 244 = the loader / consumer kernel (4 consumer + 4
 * loader waves on a 128- or 64-row tile, for launches with about one tile per CU, round 4).  Pure host function: tests and measurement tools use
 * it to know what they exercise; nothing in the product path calls it.  Returns GMD_ERR_INVALID for other element types. */
int gmd_gemm_plan_info(int dtype, int M, int N, int K, int batch, int64_t workspace_bytes, int geglu, int* out4);

/* Fused Q|K|V projection of a self-attention (round 4, 16-bit types): C[M, vt_col0] (row stride ldc) = alpha * A[M,K] W[0:vt_col0, K]^T as
 * gmd_gemm_nt writes it, and the remaining N - vt_col0 columns (the V projection) TRANSPOSED, the way gmd_attention reads V:
 * Vt[sample][column - vt_col0][token] with row stride vt_ld, `vt_tokens` consecutive rows of A per sample.  One launch instead of the
 * projection plus a batched transposed GEMM per attention.  gmd_gemm_qkv_vt_ok: 1 when the launch's plan can do it (full tiles, the V
 * columns on a tile boundary, vt_tokens a multiple of 64); otherwise use gmd_gemm_nt twice.
 * reference: Attention.to_q / to_k / to_v of diffusers as called at gm_diffusion/pipelines/stable_diffusion_dual_unet.py:1052 */
int gmd_gemm_qkv_vt_ok(int dtype, int M, int N, int K, int vt_col0, int vt_tokens, int64_t workspace_bytes);
int gmd_gemm_qkv_vt(const void* A, const void* W, void* C, void* Vt, int dtype, int M, int N, int K, int64_t ldc, int vt_col0, int vt_tokens,
                    int64_t vt_ld, float alpha, void* workspace, int64_t workspace_bytes, gmd_stream_t stream);

/* 1 when a float32-split gmd_gemm_nt launch of these dimensions (batch 1) can take out_dtype = GMD_F32SA, i.e. store its [M, N] (GEGLU:
 * [M, N/2]) result pre-split for the contraction that follows: an unsplit launch of full 128-row tiles through the row epilogues. */
int gmd_gemm_out_split_ok(int M, int N, int K, int geglu, int64_t workspace_bytes);

/* Debug facility like gmd_gemm_plan_override (refused unless the process has GMD_TUNING=1): how stride-1 gmd_conv3x3 launches on
 * 256-row ping-pong tiles fetch their activations -- 2 (the default) = the tile's input patch resident in LDS, continuous consumers;
 * 1 = patch resident, ping-pong consumers; 0 = the per-tap implicit GEMM.  Results agree to rounding (the K order differs).  Seeded
 * from GMD_CONV_PATCH when the library is loaded. */
int gmd_conv_patch_override(int mode);

/* Measurement only: a one-thread kernel that writes the device's constant-rate 100 MHz counter into base[*row * stride + k] (device
 * memory; `row` may be NULL = row 0) at its place in `stream` -- also inside a captured HIP graph, where `row` (a device scalar the host
 * rewrites between replays) lets every replay fill its own row.  tools/timeline.py places such stamps at the block boundaries of both
 * UNet forwards to record the concurrent timeline of the two-stream pipeline. */
int gmd_stamp(uint64_t* base, const int* row, int stride, int k, gmd_stream_t stream);

/* GEGLU feed-forward of a BasicTransformerBlock in ONE launch (diffusers FeedForward: ff.net.0 = GEGLU(C -> 4C), ff.net.2 =
 * Linear(4C -> C); the reference reaches it through UNet2DConditionModel at stable_diffusion_dual_unet.py:1052, 1083):
 *   Y = (value * gelu_erf(gate)) @ W2^T + b2 + residual,  [value | gate] = X @ W1i^T + b1i
 * X, residual, Y: [M, C] of `dtype` (16-bit); W1i [8C, C] / b1i [8C]: ff.net.0.proj with value / gate rows interleaved in
 * 16-row groups (the layout of GMD_ACT_GEGLU); W2 [C, 4C], b2 [C].  The [M, 4C] GEGLU tensor stays on chip.  Instantiated
 * where a 128-row block's output fits the register file: gmd_ff_geglu_fused_supported() says whether (dtype, M, C) is
 * (C == 320, M % 128 == 0); elsewhere use gmd_gemm_nt(GMD_ACT_GEGLU) + gmd_gemm_nt. */
int gmd_ff_geglu_fused_supported(int dtype, int64_t M, int C);
int gmd_ff_geglu_fused(const void* X, const void* W1i, const float* b1i, const void* W2, const float* b2, const void* residual,
                       void* Y, int dtype, int64_t M, int C, gmd_stream_t stream);

/* 3x3 convolution, padding 1, as an implicit GEMM over channels-last data.
 * X: [B,Hin,Win,Cin]; Wt: [Cout, 9*Cin] with k = (ky*3+kx)*Cin + c; Y: [B,Hout,Wout,Cout].
 * stride 1 or 2 (Downsample2D: Hout = (Hin+2-3)/2+1); upsample=1 fuses nearest-2x
 * (Upsample2D: conv over the virtual 2Hin x 2Win image).  pad_mode 0: symmetric padding 1;
 * pad_mode 1: pad (0,1,0,1) then stride 2 (VAE encoder Downsample2D(padding=0)).
 * Epilogue as gmd_gemm_nt (rowbias is [B, ldrb], one row per sample; alpha scales the
 * accumulated sum before the bias: 1 for a plain convolution, 2^-s for a weight that was stored scaled by 2^s).  Cin % 64 == 0 (BF16 / F16), % 32 (F32S / F32SW), % 16 (F32). */
int gmd_conv3x3(const void* X, const void* Wt, void* Y, int dtype, int out_dtype,
                int B, int Hin, int Win, int Cin, int Cout, int stride, int upsample, int pad_mode,
                const float* bias, const float* rowbias, int64_t ldrb, const void* residual, float alpha,
                float* colstats, int colstats_bucket,
                void* workspace, int64_t workspace_bytes, gmd_stream_t stream);

/* conv3x3 whose output goes straight into a GroupNorm (+SiLU): ResnetBlock2D's conv1 -> (+ time embedding) -> norm2 -> SiLU
 * (diffusers ResnetBlock2D.forward; reached through UNet2DConditionModel at stable_diffusion_dual_unet.py:1052, 1083).  On the
 * 16x16 / 8x8 UNet levels the convolution runs split-K (float32 partial slabs in `workspace`); here the GroupNorm kernel sums
 * the slabs itself, applies the convolution's epilogue (alpha, bias, rowbias, residual), rounds to the activation type exactly
 * as the stored tensor would be rounded and normalises from registers: one launch instead of reduce + GroupNorm, the raw
 * tensor neither written nor re-read.  Ynorm = GroupNorm(conv(X)) [+ SiLU]; Yraw (NULL to skip) = conv(X) as gmd_conv3x3
 * would store it.  Results are bit-identical to gmd_conv3x3 followed by gmd_groupnorm_fused.
 * dtype: GMD_BF16 / GMD_F16 (tensors of that type) or GMD_F32S / GMD_F32SW (float32 tensors).  Only launches whose plan is
 * split-K and whose (sample, group) slice fits the register-resident GroupNorm fuse: gmd_conv3x3_gn_fusable() returns 1 for
 * exactly those (same arguments; `groups` = GroupNorm groups); for the others gmd_conv3x3_groupnorm returns
 * GMD_ERR_UNSUPPORTED and the caller issues gmd_conv3x3 + a GroupNorm entry point. */
int gmd_conv3x3_gn_fusable(int dtype, int B, int Hin, int Win, int Cin, int Cout, int stride, int upsample, int pad_mode,
                           int groups, int64_t workspace_bytes);
int gmd_conv3x3_groupnorm(const void* X, const void* Wt, void* Yraw, void* Ynorm, int dtype,
                          int B, int Hin, int Win, int Cin, int Cout, int stride, int upsample, int pad_mode,
                          const float* bias, const float* rowbias, int64_t ldrb, const void* residual, float alpha,
                          int groups, float eps, const float* gamma, const float* beta, int silu,
                          void* workspace, int64_t workspace_bytes, gmd_stream_t stream);

/* Flash-style attention, bf16 MFMA: O = softmax(scale * Q K^T) V per (batch, head).
 * Q: [B,Nq,*] head h at columns h*D..h*D+D, row stride ldq; K likewise (ldk);
 * Vt: V transposed, [B, H*D, ldvt] (keys contiguous, ldvt >= Nk, multiple of 8);
 * O: [B,Nq,H*D] row stride ldo.  D in {32,40,64,80,160}. */
int gmd_attention(const void* Q, const void* K, const void* Vt, void* O, int dtype,
                  int B, int H, int D, int Nq, int Nk,
                  int64_t ldq, int64_t ldk, int64_t ldvt, int64_t ldo,
                  int64_t strideQ, int64_t strideK, int64_t strideVt, int64_t strideO,
                  float scale, int causal, gmd_stream_t stream);
/* causal != 0 (needs Nq == Nk): query q attends keys 0..q only -- the CLIP text encoder's mask
 * (transformers CLIPTextModel, used through stable_diffusion_gm.py:398-439). */

/* row softmax: P[r, :cols] = softmax(scale * S[r, :cols]); S float32 ld lds, P out_dtype ld ldp;
 * columns cols..ldp-1 of P are zero-filled.  causal_nq > 0: row r belongs to query r % causal_nq and
 * columns beyond that query get probability 0. */
int gmd_softmax_rows(const float* S, int64_t lds, void* P, int out_dtype, int64_t ldp,
                     int64_t rows, int cols, float scale, int causal_nq, gmd_stream_t stream);

/* GroupNorm statistics over channels-last X [B,HW,C] -> per (b,c) affine
 * scale_shift[b][c] = {rstd*gamma[c], beta[c]-mean*rstd*gamma[c]}.
 * workspace: float32, at least B*nsplit*G*2 floats where nsplit = gmd_groupnorm_nsplit(HW). */
int gmd_groupnorm_nsplit(int64_t HW);
int gmd_groupnorm_stats(const void* X, int dtype, int B, int64_t HW, int C, int G, float eps,
                        const float* gamma, const float* beta, float* workspace,
                        float* scale_shift, gmd_stream_t stream);
/* Y = [silu](X*scale+shift) */
/* GroupNorm(+SiLU) in two launches for slabs too large for gmd_groupnorm_fused: partial sums per (sample, row split, group)
 * into `workspace` (same size as for gmd_groupnorm_stats), then an apply kernel whose workgroups fold the partials of their
 * sample themselves (deterministic order) -- no separate finalize launch, no scale_shift tensor. */
int gmd_groupnorm_split(const void* X, void* Y, int dtype, int B, int64_t HW, int C, int G, float eps,
                        const float* gamma, const float* beta, float* workspace, int silu, gmd_stream_t stream);
int gmd_groupnorm_apply(const void* X, void* Y, int dtype, int B, int64_t HW, int C,
                        const float* scale_shift, int silu, gmd_stream_t stream);
/* GroupNorm(+SiLU) in ONE pass over X, with the statistics its producer(s) left (`colstats` of gmd_gemm_nt / gmd_conv3x3,
 * one row block per 64 rows: HW % 64 == 0).  X may be the channel concatenation of two producers' outputs (UNet skip
 * connections): channels [0, Ca) use stats_a [B*HW/64][Ca/bucket][2], channels [Ca, C) stats_b [B*HW/64][(C-Ca)/bucket][2]
 * (NULL when Ca == C).  bucket must divide C/G, Ca and C-Ca. */
int gmd_groupnorm_colstats(const void* X, void* Y, int dtype, int B, int64_t HW, int C, int G, float eps,
                           const float* gamma, const float* beta, const float* stats_a, int Ca,
                           const float* stats_b, int bucket, int silu, gmd_stream_t stream);
/* GroupNorm(+SiLU) in ONE launch: one workgroup per (sample, group) over the group's [HW][C/G] slab (exact two-pass
 * variance, fixed reduction order; the slab is re-read from L2).  Returns GMD_ERR_UNSUPPORTED when the slab exceeds
 * 128 KiB: use gmd_groupnorm_stats + gmd_groupnorm_apply then. */
int gmd_groupnorm_fused(const void* X, void* Y, int dtype, int B, int64_t HW, int C, int G, float eps,
                        const float* gamma, const float* beta, int silu, gmd_stream_t stream);
/* LayerNorm over the last dim (C % 8 == 0, C <= 2048) */
int gmd_layernorm(const void* X, void* Y, int dtype, int64_t rows, int C,
                  const float* gamma, const float* beta, float eps, gmd_stream_t stream);
/* GEGLU: Y[r, f] = X[r, f] * gelu_erf(X[r, F + f]); X [rows, 2F], Y [rows, F] */
int gmd_geglu(const void* X, void* Y, int dtype, int64_t rows, int F, gmd_stream_t stream);
/* sinusoidal timestep embedding (diffusers get_timestep_embedding): out [B, dim] of `dtype`;
 * timestep read from DEVICE memory (t_dev, float32 scalar) so captured graphs can be replayed. */
int gmd_timestep_embedding(const float* t_dev, void* out, int dtype, int B, int dim,
                           int flip_sin_to_cos, float freq_shift, gmd_stream_t stream);
/* out[r, :Ca] = A[r], out[r, Ca:] = Bm[r]  (skip-connection concat, channels-last) */
int gmd_concat_channels(const void* A, int Ca, const void* Bm, int Cb, void* out, int dtype,
                        int64_t rows, gmd_stream_t stream);
/* out[r, :] = table[ids[r], :] + pos[r % T, :]  (token + position embedding of the CLIP text encoder,
 * stable_diffusion_gm.py:398-439); ids int32 in [0, vocab) -- out-of-range ids are an error the HOST must
 * exclude (checked there), rows = B*T. */
int gmd_embedding_lookup(const int32_t* ids, const void* table, const void* pos, void* out, int dtype,
                         int64_t rows, int T, int C, int vocab, gmd_stream_t stream);
/* out[0:bytes] = out[bytes:2*bytes] = in[0:bytes] (bytes % 16 == 0): the batch duplication of classifier-free guidance
 * (stable_diffusion_dual_unet.py:1045-1047 torch.cat([latents] * 2)) applied to an activation where the two conditionings
 * first differ (the CFG shared prefix of UNet2DConditionModel.forward_packed) */
int gmd_dup_batch(const void* in, void* out, int64_t bytes, gmd_stream_t stream);
/* elementwise cast between F32 and BF16 */
int gmd_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, gmd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GMD_HIP_H */
