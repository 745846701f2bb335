// Latent-side kernels of the denoising loop: CFG combine + x0 prediction + PNDM/PLMS update in one
// pass, the guidance-rescale std reduction, and the UNet input pack / output unpack (8-channel
// concat + CFG duplicate + NCHW<->channels-last + cast + channel padding).
//
// Built with -ffp-contract=off: float32 operations are issued in the same order as the torch
// expressions of the reference so the results are bit-identical to the CPU path given identical eps.
#include "gmd_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void latent_step_kernel(
    const float* __restrict__ eps_in, const float* __restrict__ x, const float* __restrict__ cur_sample,
    const float* __restrict__ e1, const float* __restrict__ e2, const float* __restrict__ e3, int B, int64_t chw,
    int do_cfg, float gs, const float* __restrict__ ratio, float gr, int mode, float sample_coeff,
    float alpha_delta, float denom, float sqrt_a, float sqrt_1ma, float* __restrict__ eps_out,
    float* __restrict__ x_prev, float* __restrict__ x0) {
    GMD_WG_TRACE_SCOPE(WGK_LATENT_STEP);
    const int64_t n = (int64_t)B * chw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float eps;
        if (do_cfg) {
            const float u = eps_in[i], t = eps_in[n + i];
            eps = u + gs * (t - u);  // dual.py:1065
            if (ratio) {             // rescale_noise_cfg, dual.py:91-93
                const float resc = eps * ratio[i / chw];
                eps = gr * resc + (1.0f - gr) * eps;
            }
        } else {
            eps = eps_in[i];
        }
        const float xt = x[i];
        if (eps_out) eps_out[i] = eps;
        if (x0) x0[i] = (xt - sqrt_1ma * eps) / sqrt_a;  // dual.py:1075
        float m, smp = xt;
        switch (mode) {  // diffusers PNDMScheduler.step_plms
            case 0: m = eps; break;
            case 1: m = (eps + e1[i]) / 2.0f; smp = cur_sample[i]; break;
            case 2: m = (3.0f * eps - e1[i]) / 2.0f; break;
            case 3: m = (23.0f * eps - 16.0f * e1[i] + 5.0f * e2[i]) / 12.0f; break;
            default: m = (1.0f / 24.0f) * (55.0f * eps - 59.0f * e1[i] + 37.0f * e2[i] - 9.0f * e3[i]); break;
        }
        x_prev[i] = sample_coeff * smp - alpha_delta * m / denom;  // _get_prev_sample
    }
}

// DPM-Solver++ (dpmsolver++ / midpoint / epsilon) multistep update fused with the CFG combine and both x0 forms, in the
// operation order of the torch expressions of diffusers' DPMSolverMultistepScheduler (float32, no FMA contraction):
//   m0     = (x - sigma_s0 * eps) / alpha_s0                                   convert_model_output
//   order1 : x_prev = c_x * x - c_m * m0                                       dpm_solver_first_order_update
//   order2 : x_prev = c_x * x - c_m * m0 - c_h * (inv_r0 * (m0 - m1))          multistep_dpm_solver_second_order_update
// with c_x = sigma_t/sigma_s0, c_m = alpha_t*(exp(-h)-1), c_h = 0.5*c_m computed by the host in float32.
__global__ __launch_bounds__(kThreads) void dpm_step_kernel(
    const float* __restrict__ eps_in, const float* __restrict__ x, const float* __restrict__ m1, int B, int64_t chw, int do_cfg,
    float gs, const float* __restrict__ ratio, float gr, int order, float sigma_s0, float alpha_s0, float c_x, float c_m, float c_h,
    float inv_r0, float sqrt_a, float sqrt_1ma, float* __restrict__ m0_out, float* __restrict__ x_prev, float* __restrict__ x0) {
    GMD_WG_TRACE_SCOPE(WGK_LATENT_STEP);
    const int64_t n = (int64_t)B * chw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float eps;
        if (do_cfg) {
            const float u = eps_in[i], t = eps_in[n + i];
            eps = u + gs * (t - u);  // dual.py:1065
            if (ratio) {             // rescale_noise_cfg, dual.py:91-93
                const float resc = eps * ratio[i / chw];
                eps = gr * resc + (1.0f - gr) * eps;
            }
        } else {
            eps = eps_in[i];
        }
        const float xt = x[i];
        if (x0) x0[i] = (xt - sqrt_1ma * eps) / sqrt_a;  // dual.py:1075 (the pipeline's own x0, from alphas_cumprod[t])
        const float m0 = (xt - sigma_s0 * eps) / alpha_s0;
        m0_out[i] = m0;
        float r = c_x * xt - c_m * m0;
        if (order == 2) r = r - c_h * (inv_r0 * (m0 - m1[i]));
        x_prev[i] = r;
    }
}

// DDPM ancestral step (the scheduler scripts/inference/generate_hdr.py:162 constructs) fused with the CFG combine and the
// pipeline's x0, in the operation order of the torch expressions of diffusers' DDPMScheduler.step (float32, no FMA):
//   x0    = (x - sqrt(1-a_t) * eps) / sqrt(a_t)            [clamped to +-clip_range when clip_sample]
//   prev  = x0_coeff * x0 + xt_coeff * x                    posterior mean, formula (7) of the DDPM paper
//   prev  = prev + noise_scale * noise                      t > 0 only (noise == nullptr at the last step)
// The noise tensor is drawn by the HOST scheduler with randn_tensor(generator) so that the shared generator of the dual
// pipeline is consumed in the reference's order (SDR step first, GM step second: stable_diffusion_dual_unet.py:1077, 1093).
__global__ __launch_bounds__(kThreads) void ddpm_step_kernel(
    const float* __restrict__ eps_in, const float* __restrict__ x, const float* __restrict__ noise, int B, int64_t chw, int do_cfg,
    float gs, const float* __restrict__ ratio, float gr, float sched_sqrt_a, float sched_sqrt_1ma, int clip, float clip_range,
    float x0_coeff, float xt_coeff, float noise_scale, float sqrt_a, float sqrt_1ma, float* __restrict__ x_prev,
    float* __restrict__ x0) {
    GMD_WG_TRACE_SCOPE(WGK_LATENT_STEP);
    const int64_t n = (int64_t)B * chw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float eps;
        if (do_cfg) {
            const float u = eps_in[i], t = eps_in[n + i];
            eps = u + gs * (t - u);  // dual.py:1065
            if (ratio) {             // rescale_noise_cfg, dual.py:91-93
                const float resc = eps * ratio[i / chw];
                eps = gr * resc + (1.0f - gr) * eps;
            }
        } else {
            eps = eps_in[i];
        }
        const float xt = x[i];
        if (x0) x0[i] = (xt - sqrt_1ma * eps) / sqrt_a;  // dual.py:1075 (never clipped)
        float p0 = (xt - sched_sqrt_1ma * eps) / sched_sqrt_a;
        if (clip) p0 = fminf(fmaxf(p0, -clip_range), clip_range);
        float r = x0_coeff * p0 + xt_coeff * xt;
        if (noise) r = r + noise_scale * noise[i];
        x_prev[i] = r;
    }
}

// one block per sample: unbiased std over chw of text eps and of the guided eps
__global__ __launch_bounds__(kThreads) void cfg_std_ratio_kernel(const float* __restrict__ eps_in, int B, int64_t chw,
                                                                 float gs, float* __restrict__ ratio) {
    GMD_WG_TRACE_SCOPE(WGK_CFG_RATIO);
    const int b = blockIdx.x;
    const int64_t n = (int64_t)B * chw;
    const float* u = eps_in + (int64_t)b * chw;
    const float* t = eps_in + n + (int64_t)b * chw;
    double st = 0, st2 = 0, sc = 0, sc2 = 0;
    for (int64_t i = threadIdx.x; i < chw; i += blockDim.x) {
        const float tv = t[i], uv = u[i];
        const float cv = uv + gs * (tv - uv);
        st += tv; st2 += (double)tv * tv; sc += cv; sc2 += (double)cv * cv;
    }
    __shared__ double red[4][kThreads];
    red[0][threadIdx.x] = st; red[1][threadIdx.x] = st2; red[2][threadIdx.x] = sc; red[3][threadIdx.x] = sc2;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double N = (double)chw;
        const double vt = (red[1][0] - red[0][0] * red[0][0] / N) / (N - 1.0);
        const double vc = (red[3][0] - red[2][0] * red[2][0] / N) / (N - 1.0);
        ratio[b] = (float)sqrt(vt) / (float)sqrt(vc);  // float32 std values, float32 division
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void pack_kernel(const float* __restrict__ s0, int C0, const float* __restrict__ s1,
                                                        int C1, int B, int64_t HW, int dup, T* __restrict__ out, int CP) {
    GMD_WG_TRACE_SCOPE(WGK_PACK);
    // one thread per (pixel, 8-channel group) of the padded output
    const int groups = CP / 8;
    const int64_t total = (int64_t)B * HW * groups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        const int64_t bp = i / groups;
        const int64_t b = bp / HW, p = bp - b * HW;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = g * 8 + j;
            float val = 0.0f;
            if (c < C0) val = s0[(b * C0 + c) * HW + p];
            else if (c < C0 + C1) val = s1[(b * C1 + (c - C0)) * HW + p];
            v[j] = val;
        }
        for (int d = 0; d < dup; ++d) {
            T* o = out + (((int64_t)d * B + b) * HW + p) * CP + g * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) Elem<T>::st(o + j, v[j]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void unpack_kernel(const T* __restrict__ in, int64_t ld, int B, int C, int64_t HW,
                                                          float* __restrict__ out) {
    GMD_WG_TRACE_SCOPE(WGK_UNPACK);
    const int64_t total = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % HW;
        const int64_t bc = i / HW;
        const int64_t b = bc / C, c = bc - b * C;
        out[i] = Elem<T>::ld(in + (b * HW + p) * ld + c);
    }
}

inline int grid_for(int64_t n) {
    int64_t g = (n + kThreads - 1) / kThreads;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" {

int gmd_latent_step(const float* eps_in, const float* x, const float* cur_sample, const float* e1, const float* e2,
                    const float* e3, int B, int64_t chw, int do_cfg, float guidance_scale, const float* rescale_ratio,
                    float guidance_rescale, int mode, float sample_coeff, float alpha_delta, float denom,
                    float sqrt_alpha, float sqrt_one_minus_alpha, float* eps_out, float* x_prev, float* x0,
                    gmd_stream_t stream) {
    GMD_REQUIRE(B >= 0 && chw >= 0, "gmd_latent_step: negative shape");
    GMD_REQUIRE(mode >= 0 && mode <= 4, "gmd_latent_step: mode %d not in 0..4", mode);
    if ((int64_t)B * chw == 0) return GMD_OK;
    GMD_REQUIRE(eps_in && x && x_prev, "gmd_latent_step: null pointer");
    GMD_REQUIRE(mode != 1 || (cur_sample && e1), "gmd_latent_step: mode 1 needs cur_sample and e1");
    GMD_REQUIRE(mode < 2 || e1, "gmd_latent_step: mode %d needs e1", mode);
    GMD_REQUIRE(mode < 3 || e2, "gmd_latent_step: mode %d needs e2", mode);
    GMD_REQUIRE(mode < 4 || e3, "gmd_latent_step: mode 4 needs e3");
    GMD_REQUIRE(denom != 0.0f && sqrt_alpha != 0.0f, "gmd_latent_step: zero denominator");
    latent_step_kernel<<<grid_for((int64_t)B * chw), kThreads, 0, (hipStream_t)stream>>>(
        eps_in, x, cur_sample, e1, e2, e3, B, chw, do_cfg, guidance_scale, do_cfg ? rescale_ratio : nullptr,
        guidance_rescale, mode, sample_coeff, alpha_delta, denom, sqrt_alpha, sqrt_one_minus_alpha, eps_out, x_prev, x0);
    GMD_CHECK_LAUNCH("gmd_latent_step");
    return GMD_OK;
}

int gmd_dpm_step(const float* eps_in, const float* x, const float* m1, int B, int64_t chw, int do_cfg, float guidance_scale,
                 const float* rescale_ratio, float guidance_rescale, int order, float sigma_s0, float alpha_s0, float c_x, float c_m,
                 float c_h, float inv_r0, float sqrt_alpha, float sqrt_one_minus_alpha, float* m0_out, float* x_prev, float* x0,
                 gmd_stream_t stream) {
    GMD_REQUIRE(B >= 0 && chw > 0, "gmd_dpm_step: bad shape B=%d chw=%lld", B, (long long)chw);
    if (B == 0) return GMD_OK;
    GMD_REQUIRE(eps_in && x && m0_out && x_prev, "gmd_dpm_step: null pointer");
    GMD_REQUIRE(order == 1 || order == 2, "gmd_dpm_step: order must be 1 or 2 (got %d)", order);
    GMD_REQUIRE(order == 1 || m1, "gmd_dpm_step: the second-order update needs the previous x0 prediction");
    GMD_REQUIRE(alpha_s0 != 0.0f && (x0 == nullptr || sqrt_alpha != 0.0f), "gmd_dpm_step: zero denominator");
    dpm_step_kernel<<<grid_for((int64_t)B * chw), kThreads, 0, (hipStream_t)stream>>>(
        eps_in, x, m1, B, chw, do_cfg, guidance_scale, do_cfg ? rescale_ratio : nullptr, guidance_rescale, order, sigma_s0, alpha_s0,
        c_x, c_m, c_h, inv_r0, sqrt_alpha, sqrt_one_minus_alpha, m0_out, x_prev, x0);
    GMD_CHECK_LAUNCH("gmd_dpm_step");
    return GMD_OK;
}

int gmd_ddpm_step(const float* eps_in, const float* x, const float* noise, int B, int64_t chw, int do_cfg, float guidance_scale,
                  const float* rescale_ratio, float guidance_rescale, float sched_sqrt_alpha, float sched_sqrt_one_minus_alpha,
                  int clip_sample, float clip_range, float x0_coeff, float xt_coeff, float noise_scale, float sqrt_alpha,
                  float sqrt_one_minus_alpha, float* x_prev, float* x0, gmd_stream_t stream) {
    GMD_REQUIRE(B >= 0 && chw > 0, "gmd_ddpm_step: bad shape B=%d chw=%lld", B, (long long)chw);
    if (B == 0) return GMD_OK;
    GMD_REQUIRE(eps_in && x && x_prev, "gmd_ddpm_step: null pointer");
    GMD_REQUIRE(sched_sqrt_alpha != 0.0f && (x0 == nullptr || sqrt_alpha != 0.0f), "gmd_ddpm_step: zero denominator");
    GMD_REQUIRE(!clip_sample || clip_range > 0.0f, "gmd_ddpm_step: clip_range must be positive");
    ddpm_step_kernel<<<grid_for((int64_t)B * chw), kThreads, 0, (hipStream_t)stream>>>(
        eps_in, x, noise, B, chw, do_cfg, guidance_scale, do_cfg ? rescale_ratio : nullptr, guidance_rescale, sched_sqrt_alpha,
        sched_sqrt_one_minus_alpha, clip_sample, clip_range, x0_coeff, xt_coeff, noise_scale, sqrt_alpha, sqrt_one_minus_alpha, x_prev, x0);
    GMD_CHECK_LAUNCH("gmd_ddpm_step");
    return GMD_OK;
}

int gmd_cfg_std_ratio(const float* eps_in, int B, int64_t chw, float guidance_scale, float* ratio, gmd_stream_t stream) {
    GMD_REQUIRE(B >= 0 && chw >= 2, "gmd_cfg_std_ratio: need at least 2 elements per sample");
    if (B == 0) return GMD_OK;
    GMD_REQUIRE(eps_in && ratio, "gmd_cfg_std_ratio: null pointer");
    cfg_std_ratio_kernel<<<B, kThreads, 0, (hipStream_t)stream>>>(eps_in, B, chw, guidance_scale, ratio);
    GMD_CHECK_LAUNCH("gmd_cfg_std_ratio");
    return GMD_OK;
}

int gmd_pack_unet_input(const float* src0, int C0, const float* src1, int C1, int B, int64_t HW, int dup, void* out,
                        int CP, int out_dtype, gmd_stream_t stream) {
    GMD_REQUIRE(B >= 0 && HW >= 0 && C0 > 0 && C1 >= 0, "gmd_pack_unet_input: bad shape");
    GMD_REQUIRE(CP % 8 == 0 && CP >= C0 + C1, "gmd_pack_unet_input: CP=%d must be a multiple of 8 and >= %d", CP, C0 + C1);
    GMD_REQUIRE(dup == 1 || dup == 2, "gmd_pack_unet_input: dup must be 1 or 2");
    GMD_REQUIRE(C1 == 0 || src1, "gmd_pack_unet_input: src1 is null");
    if ((int64_t)B * HW == 0) return GMD_OK;
    GMD_REQUIRE(src0 && out, "gmd_pack_unet_input: null pointer");
    const int64_t total = (int64_t)B * HW * (CP / 8);
    GMD_REQUIRE(gmd_known_dtype(out_dtype), "gmd_pack_unet_input: bad dtype %d", out_dtype);
    gmd_for_dtype(out_dtype, [&](auto tag) {
        using T = decltype(tag);
        pack_kernel<T><<<grid_for(total), kThreads, 0, (hipStream_t)stream>>>(src0, C0, src1, C1, B, HW, dup, (T*)out, CP);
    });
    GMD_CHECK_LAUNCH("gmd_pack_unet_input");
    return GMD_OK;
}

int gmd_unpack_nchw(const void* in, int in_dtype, int64_t ld, int B, int C, int64_t HW, float* out, gmd_stream_t stream) {
    GMD_REQUIRE(B >= 0 && C > 0 && HW >= 0 && ld >= C, "gmd_unpack_nchw: bad shape");
    if ((int64_t)B * HW == 0) return GMD_OK;
    GMD_REQUIRE(in && out, "gmd_unpack_nchw: null pointer");
    const int64_t total = (int64_t)B * C * HW;
    GMD_REQUIRE(gmd_known_dtype(in_dtype), "gmd_unpack_nchw: bad dtype %d", in_dtype);
    gmd_for_dtype(in_dtype, [&](auto tag) {
        using T = decltype(tag);
        unpack_kernel<T><<<grid_for(total), kThreads, 0, (hipStream_t)stream>>>((const T*)in, ld, B, C, HW, out);
    });
    GMD_CHECK_LAUNCH("gmd_unpack_nchw");
    return GMD_OK;
}

}  // extern "C"

GMD_WG_TRACE_SETTER(latent_step)
