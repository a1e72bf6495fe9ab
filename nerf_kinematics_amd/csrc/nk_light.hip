// Per-ray stages of render_rays for gfx950: ray generation, stratified depths, alpha
// compositing (forward and backward), hierarchical sampling, loss and the Adam update.
// All of these are HBM-bound scans over [rays, samples]: one 64-lane wavefront owns one ray,
// lanes read consecutive samples (coalesced), prefix products / sums run as wavefront scans.
#include "nk_common.h"

// ============================================================================================
// get_rays (+ ndc_rays): one thread per ray
// ============================================================================================
struct CamArgs { nk_camera cam; };

NK_D void ray_row(const nk_camera& cam, int64_t pix, float* __restrict__ out) {
    const float i = (float)(pix % cam.W), j = (float)(pix / cam.W);
    const float dx = (i - cam.cx) / cam.fx, dy = -(j - cam.cy) / cam.fy, dz = -1.0f;
    const float* m = cam.c2w;
    float d[3], o[3];
    for (int k = 0; k < 3; ++k) {
        d[k] = __fadd_rn(__fadd_rn(__fmul_rn(dx, m[4 * k + 0]), __fmul_rn(dy, m[4 * k + 1])), __fmul_rn(dz, m[4 * k + 2]));
        o[k] = m[4 * k + 3];
    }
    const float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float v[3] = {d[0] * inv, d[1] * inv, d[2] * inv};
    if (cam.ndc) {
        // shift the origin to the near plane, then warp (NeRF paper, appendix C)
        const float t = -(cam.ndc_near + o[2]) / d[2];
        for (int k = 0; k < 3; ++k) o[k] = o[k] + t * d[k];
        const float sx = -1.0f / ((float)cam.W / (2.0f * cam.fx)), sy = -1.0f / ((float)cam.H / (2.0f * cam.fy));
        const float o0 = sx * o[0] / o[2], o1 = sy * o[1] / o[2], o2 = 1.0f + 2.0f * cam.ndc_near / o[2];
        const float d0 = sx * (d[0] / d[2] - o[0] / o[2]), d1 = sy * (d[1] / d[2] - o[1] / o[2]);
        const float d2 = -2.0f * cam.ndc_near / o[2];
        o[0] = o0; o[1] = o1; o[2] = o2; d[0] = d0; d[1] = d1; d[2] = d2;
    }
    out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
    out[3] = d[0]; out[4] = d[1]; out[5] = d[2];
    out[6] = cam.near; out[7] = cam.far;
    out[8] = v[0]; out[9] = v[1]; out[10] = v[2];
}

__global__ void k_get_rays(CamArgs a, const int32_t* __restrict__ pixel_idx, int64_t first_pixel,
                           int64_t R, float* __restrict__ rb) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    ray_row(a.cam, pixel_idx ? (int64_t)pixel_idx[r] : first_pixel + r, rb + r * 11);
}

// Training batch of one view: R pixels drawn with replacement (pixel of ray r = floor(u32 * H*W /
// 2^32) with u32 = element counter + r of RNG stream NK_STREAM_PIXELS), their rays, and their
// colours from the 8-bit image (alpha composited onto white when asked, otherwise rgb as stored).
__global__ void k_sample_training_rays(CamArgs a, const uint8_t* __restrict__ image, int channels, int white,
                                       uint32_t seed, uint64_t counter, int64_t R, float* __restrict__ rb,
                                       float* __restrict__ target, int32_t* __restrict__ pixel_idx) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const uint64_t npix = (uint64_t)a.cam.H * (uint64_t)a.cam.W;
    const int64_t pix = (int64_t)(((uint64_t)nk_rand_u32(seed, NK_STREAM_PIXELS, counter + (uint64_t)r) * npix) >> 32);
    ray_row(a.cam, pix, rb + r * 11);
    const uint8_t* px = image + pix * channels;
    float c[3] = {(float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f};
    if (channels == 4 && white) {
        const float al = (float)px[3] / 255.0f;
        for (int k = 0; k < 3; ++k) c[k] = __fadd_rn(__fmul_rn(c[k], al), 1.0f - al);
    }
    target[r * 3 + 0] = c[0]; target[r * 3 + 1] = c[1]; target[r * 3 + 2] = c[2];
    if (pixel_idx) pixel_idx[r] = (int32_t)pix;
}

__global__ void k_sample_training_rays_dev(const nk_camera* __restrict__ cams, int n_views, const uint8_t* __restrict__ images,
                                           int64_t image_stride, int channels, int white,
                                           const nk_step_state* __restrict__ st, int64_t R, float* __restrict__ rb,
                                           float* __restrict__ target, int32_t* __restrict__ pixel_idx) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int view = st->view;
    if (view < 0 || view >= n_views) return;        // a step state that was never set: draw nothing rather than read outside the tables
    const nk_camera cam = cams[view];
    const uint64_t npix = (uint64_t)cam.H * (uint64_t)cam.W;
    const int64_t pix = (int64_t)(((uint64_t)nk_rand_u32(st->pixel_seed, NK_STREAM_PIXELS, st->pixel_counter + (uint64_t)r) * npix) >> 32);
    ray_row(cam, pix, rb + r * 11);
    const uint8_t* px = images + (int64_t)view * image_stride + pix * channels;
    float c[3] = {(float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f};
    if (channels == 4 && white) {
        const float al = (float)px[3] / 255.0f;
        for (int k = 0; k < 3; ++k) c[k] = __fadd_rn(__fmul_rn(c[k], al), 1.0f - al);
    }
    target[r * 3 + 0] = c[0]; target[r * 3 + 1] = c[1]; target[r * 3 + 2] = c[2];
    if (pixel_idx) pixel_idx[r] = (int32_t)pix;
}

extern "C" int nk_sample_training_rays_dev(const nk_camera* cams, int32_t n_views, const uint8_t* images,
                                           int64_t image_stride, int32_t channels, int32_t white_background,
                                           const nk_step_state* state, int64_t R, float* ray_batch, float* target,
                                           int32_t* pixel_idx, void* stream) {
    if (!cams || !images || !state || !ray_batch || !target || R <= 0 || n_views <= 0 || image_stride <= 0)
        NK_FAIL(NK_ERR_ARG, "nk_sample_training_rays_dev: null pointer or non-positive size");
    if (channels != 3 && channels != 4)
        NK_FAIL(NK_ERR_ARG, "nk_sample_training_rays_dev: %d channels (3 or 4)", channels);
    hipLaunchKernelGGL(k_sample_training_rays_dev, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, nk_stream(stream),
                       cams, (int)n_views, images, image_stride, channels, white_background, state, R, ray_batch, target, pixel_idx);
    NK_CHECK_LAUNCH("k_sample_training_rays_dev");
    return NK_OK;
}

extern "C" int nk_get_rays(const nk_camera* cam, const int32_t* pixel_idx, int64_t first_pixel,
                           int64_t R, float* ray_batch, void* stream) {
    if (!cam || !ray_batch || R <= 0) NK_FAIL(NK_ERR_ARG, "nk_get_rays: null pointer or R <= 0");
    if (cam->H <= 0 || cam->W <= 0 || cam->fx == 0.0f || cam->fy == 0.0f)
        NK_FAIL(NK_ERR_ARG, "nk_get_rays: bad camera (H=%d W=%d fx=%g fy=%g)", cam->H, cam->W, cam->fx, cam->fy);
    if (!pixel_idx && (first_pixel < 0 || first_pixel + R > (int64_t)cam->H * cam->W))
        NK_FAIL(NK_ERR_ARG, "nk_get_rays: pixel range [%lld, %lld) outside the image", (long long)first_pixel, (long long)(first_pixel + R));
    CamArgs a; a.cam = *cam;
    hipLaunchKernelGGL(k_get_rays, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, nk_stream(stream),
                       a, pixel_idx, first_pixel, R, ray_batch);
    NK_CHECK_LAUNCH("k_get_rays");
    return NK_OK;
}

extern "C" int nk_sample_training_rays(const nk_camera* cam, const uint8_t* image, int32_t channels,
                                       int32_t white_background, uint32_t seed, uint64_t counter, int64_t R,
                                       float* ray_batch, float* target, int32_t* pixel_idx, void* stream) {
    if (!cam || !image || !ray_batch || !target || R <= 0)
        NK_FAIL(NK_ERR_ARG, "nk_sample_training_rays: null pointer or R <= 0");
    if (cam->H <= 0 || cam->W <= 0 || cam->fx == 0.0f || cam->fy == 0.0f)
        NK_FAIL(NK_ERR_ARG, "nk_sample_training_rays: bad camera (H=%d W=%d fx=%g fy=%g)", cam->H, cam->W, cam->fx, cam->fy);
    if (channels != 3 && channels != 4)
        NK_FAIL(NK_ERR_ARG, "nk_sample_training_rays: %d channels (3 or 4)", channels);
    CamArgs a; a.cam = *cam;
    hipLaunchKernelGGL(k_sample_training_rays, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, nk_stream(stream),
                       a, image, channels, white_background, seed, counter, R, ray_batch, target, pixel_idx);
    NK_CHECK_LAUNCH("k_sample_training_rays");
    return NK_OK;
}

// ============================================================================================
// stratified depths: one thread per (ray, sample)
// ============================================================================================
__global__ void k_stratified_z(const float* __restrict__ rb, int64_t R, int Nc, int perturb, int lindisp,
                               uint32_t seed, const nk_step_state* __restrict__ st, float* __restrict__ z) {
    if (st) seed = st->render_seed;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * Nc) return;
    const int64_t r = e / Nc;
    const int i = (int)(e - r * Nc);
    const float near = rb[r * 11 + 6], far = rb[r * 11 + 7];
    auto zat = [&](int k) -> float {
        const float t = nk_linspace01(k, Nc);
        if (!lindisp) return __fadd_rn(__fmul_rn(near, 1.0f - t), __fmul_rn(far, t));
        return 1.0f / __fadd_rn(__fmul_rn(1.0f / near, 1.0f - t), __fmul_rn(1.0f / far, t));
    };
    float zi = zat(i);
    if (perturb) {
        const float lower = (i == 0) ? zi : 0.5f * (zi + zat(i - 1));
        const float upper = (i == Nc - 1) ? zi : 0.5f * (zat(i + 1) + zi);
        const float u = nk_rand_uniform(seed, NK_STREAM_STRATIFIED, (uint64_t)e);
        zi = __fadd_rn(lower, __fmul_rn(upper - lower, u));
    }
    z[e] = zi;
}

extern "C" int nk_stratified_z(const float* ray_batch, int64_t R, const nk_render_opts* o, float* z, void* stream) {
    if (!ray_batch || !o || !z || R <= 0) NK_FAIL(NK_ERR_ARG, "nk_stratified_z: null pointer or R <= 0");
    if (o->num_coarse < 2) NK_FAIL(NK_ERR_ARG, "nk_stratified_z: num_coarse=%d < 2", o->num_coarse);
    const int64_t total = R * o->num_coarse;
    hipLaunchKernelGGL(k_stratified_z, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nk_stream(stream),
                       ray_batch, R, o->num_coarse, o->perturb, o->lindisp, o->seed, o->state, z);
    NK_CHECK_LAUNCH("k_stratified_z");
    return NK_OK;
}

// ============================================================================================
// raw2outputs: one wavefront per ray, lane l owns samples [l*M, (l+1)*M)
// ============================================================================================
NK_D float wave_sum(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int M>
__global__ __launch_bounds__(256) void k_composite_fwd(
    const float4* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rb, int64_t R, int n,
    float noise_std, uint32_t seed, const nk_step_state* __restrict__ st, int noise_stream, int white,
    float* __restrict__ rgb, float* __restrict__ disp, float* __restrict__ accm, float* __restrict__ depth,
    float* __restrict__ weights) {
    if (st) seed = st->render_seed;
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= R) return;   // wave-uniform
    const float* rr = rb + ray * 11;
    const float dn = sqrtf(rr[3] * rr[3] + rr[4] * rr[4] + rr[5] * rr[5]);
    const int64_t base = ray * n;
    float al[M], zz[M], cr[M], cg[M], cb[M];
    float prod = 1.0f;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int j = lane * M + m;
        al[m] = 0.0f; zz[m] = 0.0f; cr[m] = cg[m] = cb[m] = 0.0f;
        if (j < n) {
            const float4 w4 = raw[base + j];
            const float zj = z[base + j];
            const float dist = ((j + 1 < n) ? (z[base + j + 1] - zj) : 1e10f) * dn;
            float s = w4.w;
            if (noise_std > 0.0f) s += noise_std * nk_rand_normal(seed, noise_stream, (uint64_t)(base + j));
            al[m] = 1.0f - expf(-fmaxf(s, 0.0f) * dist);
            zz[m] = zj;
            cr[m] = 1.0f / (1.0f + expf(-w4.x)); cg[m] = 1.0f / (1.0f + expf(-w4.y)); cb[m] = 1.0f / (1.0f + expf(-w4.z));
            prod *= (1.0f - al[m] + 1e-10f);
        }
    }
    // exclusive product scan over lanes
    float incl = prod;
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(incl, off, 64);
        if (lane >= off) incl *= t;
    }
    float T = __shfl_up(incl, 1, 64);
    if (lane == 0) T = 1.0f;
    float sr = 0, sg = 0, sb = 0, sd = 0, sa = 0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int j = lane * M + m;
        const float w = al[m] * T;
        T *= (1.0f - al[m] + 1e-10f);
        if (j < n) {
            sr += w * cr[m]; sg += w * cg[m]; sb += w * cb[m]; sd += w * zz[m]; sa += w;
            if (weights) weights[base + j] = w;
        }
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sd = wave_sum(sd); sa = wave_sum(sa);
    if (lane == 0) {
        if (white) { sr += 1.0f - sa; sg += 1.0f - sa; sb += 1.0f - sa; }
        rgb[ray * 3 + 0] = sr; rgb[ray * 3 + 1] = sg; rgb[ray * 3 + 2] = sb;
        if (depth) depth[ray] = sd;
        if (accm) accm[ray] = sa;
        if (disp) {
            const float q = sd / sa;            // NaN for an empty ray, as in the reference
            disp[ray] = (q != q) ? q : 1.0f / fmaxf(1e-10f, q);
        }
    }
}

static int composite_m(int n) {
    const int m = (n + 63) / 64;
    if (m <= 1) return 1; if (m <= 2) return 2; if (m <= 3) return 3; if (m <= 4) return 4; if (m <= 8) return 8;
    return -1;
}

extern "C" int nk_raw2outputs(const float* raw, const float* z, const float* ray_batch, int64_t R, int32_t n,
                              const nk_render_opts* o, int32_t noise_stream, float* rgb, float* disp, float* acc,
                              float* depth, float* weights, void* stream) {
    if (!raw || !z || !ray_batch || !o || !rgb || R <= 0 || n <= 0)
        NK_FAIL(NK_ERR_ARG, "nk_raw2outputs: null pointer or non-positive size");
    const int M = composite_m(n);
    if (M < 0) NK_FAIL(NK_ERR_UNSUPPORTED, "nk_raw2outputs: %d samples per ray (max 512)", n);
    const dim3 grid((unsigned)((R + 3) / 4)), block(256);
#define NK_LAUNCH_CF(MM) hipLaunchKernelGGL(k_composite_fwd<MM>, grid, block, 0, nk_stream(stream), \
        (const float4*)raw, z, ray_batch, R, (int)n, o->radiance_field_noise_std, o->seed, o->state, (int)noise_stream, \
        (int)o->white_background, rgb, disp, acc, depth, weights)
    switch (M) { case 1: NK_LAUNCH_CF(1); break; case 2: NK_LAUNCH_CF(2); break; case 3: NK_LAUNCH_CF(3); break;
                 case 4: NK_LAUNCH_CF(4); break; default: NK_LAUNCH_CF(8); break; }
#undef NK_LAUNCH_CF
    NK_CHECK_LAUNCH("k_composite_fwd");
    return NK_OK;
}

// backward: d raw from d rgb_map (+ optional d acc_map, d depth_map)
template <int M>
__global__ __launch_bounds__(256) void k_composite_bwd(
    const float4* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rb, int64_t R, int n,
    float noise_std, uint32_t seed, const nk_step_state* __restrict__ st, int noise_stream, int white,
    const float* __restrict__ d_rgb, const float* __restrict__ d_acc, const float* __restrict__ d_depth,
    float4* __restrict__ d_raw) {
    if (st) seed = st->render_seed;
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= R) return;
    const float* rr = rb + ray * 11;
    const float dn = sqrtf(rr[3] * rr[3] + rr[4] * rr[4] + rr[5] * rr[5]);
    const int64_t base = ray * n;
    const float gr = d_rgb ? d_rgb[ray * 3 + 0] : 0.0f, gg = d_rgb ? d_rgb[ray * 3 + 1] : 0.0f,
                gb = d_rgb ? d_rgb[ray * 3 + 2] : 0.0f;
    float ga = d_acc ? d_acc[ray] : 0.0f;
    const float gd = d_depth ? d_depth[ray] : 0.0f;
    if (white) ga -= gr + gg + gb;
    float al[M], zz[M], cr[M], cg[M], cb[M], ds[M], pos[M];
    float prod = 1.0f;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int j = lane * M + m;
        al[m] = 0.0f; zz[m] = 0.0f; cr[m] = cg[m] = cb[m] = 0.0f; ds[m] = 0.0f; pos[m] = 0.0f;
        if (j < n) {
            const float4 w4 = raw[base + j];
            const float zj = z[base + j];
            const float dist = ((j + 1 < n) ? (z[base + j + 1] - zj) : 1e10f) * dn;
            float s = w4.w;
            if (noise_std > 0.0f) s += noise_std * nk_rand_normal(seed, noise_stream, (uint64_t)(base + j));
            al[m] = 1.0f - expf(-fmaxf(s, 0.0f) * dist);
            zz[m] = zj; ds[m] = dist; pos[m] = s > 0.0f ? 1.0f : 0.0f;
            cr[m] = 1.0f / (1.0f + expf(-w4.x)); cg[m] = 1.0f / (1.0f + expf(-w4.y)); cb[m] = 1.0f / (1.0f + expf(-w4.z));
            prod *= (1.0f - al[m] + 1e-10f);
        }
    }
    float incl = prod;
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(incl, off, 64);
        if (lane >= off) incl *= t;
    }
    float T0 = __shfl_up(incl, 1, 64);
    if (lane == 0) T0 = 1.0f;
    // g_i = dL/dw_i, local sums of g_i w_i, then an exclusive suffix sum over lanes
    float Tm[M], gw[M], g[M];
    float T = T0, loc = 0.0f;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        Tm[m] = T;
        const float w = al[m] * T;
        T *= (1.0f - al[m] + 1e-10f);
        g[m] = gr * cr[m] + gg * cg[m] + gb * cb[m] + gd * zz[m] + ga;
        gw[m] = g[m] * w;
        loc += gw[m];
    }
    float suf = loc;
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_down(suf, off, 64);
        if (lane + off < 64) suf += t;
    }
    float S = __shfl_down(suf, 1, 64);
    if (lane == 63) S = 0.0f;
#pragma unroll
    for (int m = M - 1; m >= 0; --m) {
        const int j = lane * M + m;
        if (j < n) {
            const float w = al[m] * Tm[m];
            const float dalpha = g[m] * Tm[m] - S / (1.0f - al[m] + 1e-10f);
            float4 o;
            o.x = w * gr * cr[m] * (1.0f - cr[m]);
            o.y = w * gg * cg[m] * (1.0f - cg[m]);
            o.z = w * gb * cb[m] * (1.0f - cb[m]);
            o.w = dalpha * ds[m] * (1.0f - al[m]) * pos[m];
            d_raw[base + j] = o;
        }
        S += gw[m];
    }
}

extern "C" int nk_raw2outputs_bwd(const float* raw, const float* z, const float* ray_batch, int64_t R, int32_t n,
                                  const nk_render_opts* o, int32_t noise_stream, const float* d_rgb,
                                  const float* d_acc, const float* d_depth, float* d_raw, void* stream) {
    if (!raw || !z || !ray_batch || !o || !d_raw || R <= 0 || n <= 0)
        NK_FAIL(NK_ERR_ARG, "nk_raw2outputs_bwd: null pointer or non-positive size");
    const int M = composite_m(n);
    if (M < 0) NK_FAIL(NK_ERR_UNSUPPORTED, "nk_raw2outputs_bwd: %d samples per ray (max 512)", n);
    const dim3 grid((unsigned)((R + 3) / 4)), block(256);
#define NK_LAUNCH_CB(MM) hipLaunchKernelGGL(k_composite_bwd<MM>, grid, block, 0, nk_stream(stream), \
        (const float4*)raw, z, ray_batch, R, (int)n, o->radiance_field_noise_std, o->seed, o->state, (int)noise_stream, \
        (int)o->white_background, d_rgb, d_acc, d_depth, (float4*)d_raw)
    switch (M) { case 1: NK_LAUNCH_CB(1); break; case 2: NK_LAUNCH_CB(2); break; case 3: NK_LAUNCH_CB(3); break;
                 case 4: NK_LAUNCH_CB(4); break; default: NK_LAUNCH_CB(8); break; }
#undef NK_LAUNCH_CB
    NK_CHECK_LAUNCH("k_composite_bwd");
    return NK_OK;
}

// ============================================================================================
// sample_pdf + sort(cat[z, z_samples]): one wavefront per ray, CDF and bins staged in LDS
// ============================================================================================
enum { PDF_MAX_NC = 256, PDF_MAX_N = 512 };

// A wavefront works on its own ray and its own LDS rows: its LDS operations execute in issue
// order, so between phases it needs no workgroup barrier, only a point the compiler does not
// move memory operations across.
NK_D void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ascending bitonic sort of the first n (a power of two) floats of one wavefront's LDS row
NK_D void wave_bitonic(float* v, int n, int lane) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < n; i += 64) {
                const int l = i ^ j;
                if (l > i) {
                    const float a = v[i], b = v[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { v[i] = b; v[l] = a; }
                }
            }
            wave_sync();
        }
    }
}

// The depths of the coarse pass are ascending already, so only the Nf new samples are sorted
// (bitonic, NPF = Nf rounded up to a power of two, padded with +inf) and the two ascending lists
// are merged by rank: a coarse depth goes to (its index + samples below it), a sample to (its
// index + coarse depths not above it).  The result is the sorted multiset torch.sort returns.
__global__ __launch_bounds__(256) void k_sample_pdf(
    const float* __restrict__ zc, const float* __restrict__ wts, int64_t R, int Nc, int Nf, int NPF, int NPA, int perturb,
    uint32_t seed, const nk_step_state* __restrict__ st, float* __restrict__ z_all, float* __restrict__ z_samples) {
    if (st) seed = st->render_seed;
    __shared__ float s_cdf[4][PDF_MAX_NC];
    __shared__ float s_bins[4][PDF_MAX_NC];
    __shared__ float s_zc[4][PDF_MAX_NC];
    __shared__ float s_z[4][PDF_MAX_N];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
    if (ray >= R) return;                               // (no workgroup barrier below)
    const float* zr = zc + ray * Nc;
    const float* wr = wts + ray * Nc;
    float* cdf = s_cdf[wv]; float* bins = s_bins[wv]; float* zs = s_z[wv]; float* zcl = s_zc[wv];
    const int B = Nc - 1;          // bins = midpoints; interior weights = B - 1
    const int NW = B - 1;
    // midpoints, and the sum of (w + 1e-5) over the interior weights
    float loc = 0.0f;
    for (int i = lane; i < B; i += 64) bins[i] = 0.5f * (zr[i + 1] + zr[i]);
    const int CH = (NW + 63) / 64;                      // consecutive weights per lane
    float wl[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int i = lane * CH + m;
        wl[m] = (m < CH && i < NW) ? wr[1 + i] + 1e-5f : 0.0f;
        loc += wl[m];
    }
    const float total = wave_sum(loc);
    // inclusive cumsum of pdf = w / total
    float run = 0.0f, inc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) { run += wl[m] / total; inc[m] = run; }
    float scan = run;
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(scan, off, 64);
        if (lane >= off) scan += t;
    }
    float excl = __shfl_up(scan, 1, 64);
    if (lane == 0) { excl = 0.0f; cdf[0] = 0.0f; }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int i = lane * CH + m;
        if (m < CH && i < NW) cdf[i + 1] = excl + inc[m];
    }
    for (int i = lane; i < Nc; i += 64) zcl[i] = zr[i];
    for (int i = Nf + lane; i < NPF; i += 64) zs[i] = __builtin_inff();
    wave_sync();
    // inverse CDF
    for (int k = lane; k < Nf; k += 64) {
        const float u = perturb ? nk_rand_uniform(seed, NK_STREAM_PDF, (uint64_t)(ray * Nf + k)) : nk_linspace01(k, Nf);
        int lo = 0, hi = B;                              // first index with cdf[idx] > u
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] > u) hi = mid; else lo = mid + 1; }
        const int below = max(lo - 1, 0), above = min(lo, B - 1);
        const float cb = cdf[below], ca = cdf[above];
        float denom = ca - cb;
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (u - cb) / denom;
        const float bb = bins[below], ba = bins[above];
        const float s = __fadd_rn(bb, __fmul_rn(t, ba - bb));
        zs[k] = s;
        if (z_samples) z_samples[ray * Nf + k] = s;
    }
    wave_sync();
    // The merge by rank needs ascending coarse depths (what nk_stratified_z and the reference's depth
    // sampling produce).  Any other input -- unsorted, or holding a NaN (every comparison with it is
    // false) -- takes the reference's own route: sort the concatenation of both lists.
    bool asc = true;
    for (int i = lane; i + 1 < Nc; i += 64) asc = asc && (zcl[i] <= zcl[i + 1]);
    const bool merge = __builtin_amdgcn_ballot_w64(!asc) == 0;      // wave-uniform
    float* out = z_all + ray * (Nc + Nf);
    if (!merge) {
        for (int i = lane; i < Nc; i += 64) zs[Nf + i] = zcl[i];
        for (int i = Nc + Nf + lane; i < NPA; i += 64) zs[i] = __builtin_inff();
        wave_sync();
        wave_bitonic(zs, NPA, lane);
        for (int i = lane; i < Nc + Nf; i += 64) out[i] = zs[i];
        return;
    }
    // bitonic sort of the NPF sample slots, ascending
    wave_bitonic(zs, NPF, lane);
    // merge by rank
    for (int i = lane; i < Nc; i += 64) {
        const float v = zcl[i];
        int lo = 0, hi = Nf;                             // samples strictly below v
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (zs[mid] < v) lo = mid + 1; else hi = mid; }
        out[i + lo] = v;
    }
    for (int k = lane; k < Nf; k += 64) {
        const float v = zs[k];
        int lo = 0, hi = Nc;                             // coarse depths not above v
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (zcl[mid] <= v) lo = mid + 1; else hi = mid; }
        out[k + lo] = v;
    }
}

extern "C" int nk_sample_pdf(const float* z_coarse, const float* weights, int64_t R, const nk_render_opts* o,
                             float* z_all, float* z_samples, void* stream) {
    if (!z_coarse || !weights || !o || !z_all || R <= 0) NK_FAIL(NK_ERR_ARG, "nk_sample_pdf: null pointer or R <= 0");
    const int Nc = o->num_coarse, Nf = o->num_fine;
    if (Nc < 3 || Nc > PDF_MAX_NC || Nf < 1 || Nc + Nf > PDF_MAX_N)
        NK_FAIL(NK_ERR_UNSUPPORTED, "nk_sample_pdf: num_coarse=%d num_fine=%d outside 3..%d / sum<=%d", Nc, Nf, PDF_MAX_NC, PDF_MAX_N);
    int NPF = 2, NPA = 2;             // sample count and sample + coarse count, rounded up to powers of two
    while (NPF < Nf) NPF <<= 1;
    while (NPA < Nc + Nf) NPA <<= 1;
    hipLaunchKernelGGL(k_sample_pdf, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, nk_stream(stream),
                       z_coarse, weights, R, Nc, Nf, NPF, NPA, (int)o->perturb, o->seed, o->state, z_all, z_samples);
    NK_CHECK_LAUNCH("k_sample_pdf");
    return NK_OK;
}

// ============================================================================================
// occupancy skipping (inference experiment; no counterpart in the reference, which evaluates the
// fine network on every sample)
// ============================================================================================
// One thread per fine sample, 32 consecutive samples = one group of the fine network's launch.  A
// sample is "near something" when one of the four coarse samples around its depth (two on either
// side) has compositing weight >= eps; a group with no such sample gets zero rows in raw_f and is
// left out of the list.  The list is filled through a counter (one update per block of 32 groups), so
// the order of its blocks varies from run to run; the rendered values do not depend on it (every
// listed group is evaluated on its own).
__global__ __launch_bounds__(1024) void k_mark_live_groups(const float* __restrict__ zc, const float* __restrict__ wc,
                                                           const float* __restrict__ z_all, int64_t P, int Nc, int N, float eps,
                                                           int32_t* __restrict__ live, int32_t* __restrict__ n_live,
                                                           float* __restrict__ raw_f) {
    __shared__ int flag[32];        // the block's 32 groups
    __shared__ int base;
    const int64_t pt = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    bool near = false;
    if (pt < P) {
        const int64_t ray = pt / N;
        const float* z = zc + ray * Nc;
        const float* w = wc + ray * Nc;
        const float v = z_all[pt];
        int lo = 0, hi = Nc;                             // coarse depths not above v
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (z[mid] <= v) lo = mid + 1; else hi = mid; }
        // coarse samples lo-2 .. lo+1 (v lies between lo-1 and lo)
        for (int k = lo - 2; k <= lo + 1; ++k)
            if (k >= 0 && k < Nc && w[k] >= eps) near = true;
    }
    const uint64_t vote = __ballot(near);
    const int g = threadIdx.x >> 5;                     // group inside the block
    const bool group_live = ((vote >> (32 * (g & 1))) & 0xFFFFFFFFull) != 0;
    if ((threadIdx.x & 31) == 0) flag[g] = group_live ? 1 : 0;
    __syncthreads();
    // one counter update per block (a counter update per group was 1.3 ms per 65 536 rays: they all
    // hit one address); the block's live groups take consecutive slots in ascending order
    if (threadIdx.x == 0) {
        int n = 0;
        for (int k = 0; k < 32; ++k) n += flag[k];
        base = n ? atomicAdd(n_live, n) : 0;
    }
    __syncthreads();
    if (group_live) {
        if ((threadIdx.x & 31) == 0) {
            int before = 0;
            for (int k = 0; k < g; ++k) before += flag[k];
            live[base + before] = (int32_t)(pt >> 5);
        }
    } else if (pt < P) {
        *(float4*)(raw_f + pt * 4) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

int nk_mark_live_groups(const float* z_coarse, const float* w_coarse, const float* z_all, int64_t R, int32_t Nc,
                        int32_t N, float eps, int32_t* live, int32_t* n_live, float* raw_f, void* stream) {
    if (!z_coarse || !w_coarse || !z_all || !live || !n_live || !raw_f || R <= 0 || Nc < 2 || N <= Nc)
        NK_FAIL(NK_ERR_ARG, "nk_mark_live_groups: null pointer or bad size");
    const int64_t P = R * (int64_t)N;
    NK_CHECK_HIP(hipMemsetAsync(n_live, 0, 4, nk_stream(stream)));
    hipLaunchKernelGGL(k_mark_live_groups, dim3((unsigned)((P + 1023) / 1024)), dim3(1024), 0, nk_stream(stream), z_coarse,
                       w_coarse, z_all, P, (int)Nc, (int)N, eps, live, n_live, raw_f);
    NK_CHECK_LAUNCH("k_mark_live_groups");
    return NK_OK;
}

// ============================================================================================
// loss and optimiser
// ============================================================================================
__global__ __launch_bounds__(256) void k_mse_loss(const float* __restrict__ rc, const float* __restrict__ rf,
                                                  const float* __restrict__ tgt, int64_t n3, float gscale,
                                                  float* __restrict__ loss, float* __restrict__ dc, float* __restrict__ df) {
    __shared__ float part[4];
    float acc = 0.0f;
    const float inv = 1.0f / (float)n3;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n3; e += (int64_t)gridDim.x * blockDim.x) {
        const float t = tgt[e];
        if (rc) { const float d = rc[e] - t; acc += d * d; if (dc) dc[e] = 2.0f * d * inv * gscale; }
        if (rf) { const float d = rf[e] - t; acc += d * d; if (df) df[e] = 2.0f * d * inv * gscale; }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && loss) atomicAdd(loss, (part[0] + part[1] + part[2] + part[3]) * inv);
}

extern "C" int nk_mse_loss(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t R,
                           float grad_scale, float* loss, float* d_rgb_coarse, float* d_rgb_fine, void* stream) {
    if (!target || R <= 0 || (!rgb_coarse && !rgb_fine)) NK_FAIL(NK_ERR_ARG, "nk_mse_loss: null pointer or R <= 0");
    const int64_t n3 = R * 3;
    const unsigned blocks = (unsigned)((n3 + 255) / 256 < 1024 ? (n3 + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_mse_loss, dim3(blocks), dim3(256), 0, nk_stream(stream), rgb_coarse, rgb_fine, target, n3,
                       grad_scale, loss, d_rgb_coarse, d_rgb_fine);
    NK_CHECK_LAUNCH("k_mse_loss");
    return NK_OK;
}

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int64_t n, float omb1, float b2, float omb2,
                                              float eps, float step_size, float bc2_sqrt, float gscale,
                                              const nk_step_state* __restrict__ st) {
    if (st) { step_size = st->adam_step_size; bc2_sqrt = st->adam_bc2_sqrt; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = m[i] + omb1 * (gi - m[i]);                   // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = __fadd_rn(__fmul_rn(v[i], b2), __fmul_rn(omb2, __fmul_rn(gi, gi)));   // mul_(b2).addcmul_(g, g, 1 - b2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        m[i] = mi; v[i] = vi;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

extern "C" int nk_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                            int64_t step, double lr, double beta1, double beta2, double eps, float grad_scale,
                            void* stream) {
    if (!params || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step < 1)
        NK_FAIL(NK_ERR_ARG, "nk_adam_step: null pointer, n <= 0 or step < 1");
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, nk_stream(stream), params, grad, exp_avg, exp_avg_sq, n,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)(lr / bc1),
                       (float)sqrt(bc2), grad_scale, (const nk_step_state*)nullptr);
    NK_CHECK_LAUNCH("k_adam");
    return NK_OK;
}

__global__ void k_set_step_state(nk_step_state v, nk_step_state* __restrict__ dst) { *dst = v; }

extern "C" int nk_set_step_state(const nk_step_state* state_host, nk_step_state* state_dev, void* stream) {
    if (!state_host || !state_dev) NK_FAIL(NK_ERR_ARG, "nk_set_step_state: null pointer");
    hipLaunchKernelGGL(k_set_step_state, dim3(1), dim3(1), 0, nk_stream(stream), *state_host, state_dev);
    NK_CHECK_LAUNCH("k_set_step_state");
    return NK_OK;
}

extern "C" int nk_adam_step_dev(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                const nk_step_state* state, double beta1, double beta2, double eps, float grad_scale,
                                void* stream) {
    if (!params || !grad || !exp_avg || !exp_avg_sq || !state || n <= 0)
        NK_FAIL(NK_ERR_ARG, "nk_adam_step_dev: null pointer or n <= 0");
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, nk_stream(stream), params, grad, exp_avg, exp_avg_sq, n,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, 0.0f, 1.0f, grad_scale, state);
    NK_CHECK_LAUNCH("k_adam");
    return NK_OK;
}
