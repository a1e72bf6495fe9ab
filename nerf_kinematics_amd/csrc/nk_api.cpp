// render_rays forward / backward: the stage kernels chained on one stream over a caller-owned
// workspace.  Mirrors the body of nerf-pytorch's render_rays (coarse pass, sample_pdf, fine pass)
// with the knobs of the nerf.train / nerf.validation blocks of the reference experiment files
// (/root/reference/trained_models/nerf-pytorch/lego-lowres3/config.yml:46-63).
#include <math.h>
#include <string.h>
#include <mutex>
#include "nk_common.h"

// Scratch of nk_render_rays_fwd_noalloc: device memory that is part of the loaded code object, one
// copy per GPU at its own address.  The name is only a host-side handle; host code uses the
// address hipGetSymbolAddress returns for the current device (static_scratch()).
static __device__ __align__(256) unsigned char scratch[NK_NOALLOC_SCRATCH_BYTES];

namespace {

struct WsLayout {
    int64_t z_c, raw_c, w_c, rgb_c_tmp, z_all, raw_f, live, act_c, act_f, d_raw, dz, dz_c, total;
};

// nk_render_opts.skip_empty_eps applies to inference calls with a fine pass
inline bool skips_empty(const nk_render_opts* o, int training) { return !training && o->num_fine > 0 && o->skip_empty_eps > 0.0f; }

// NK_ACT_FP8_BF8: the power of two dZ is multiplied by before it is rounded to e5m2.  d(loss)/d(rgb)
// of a mean-squared error over R rays is at most 2/(3R) per component, so with 2^(ceil(log2 R) + 6)
// the largest dZ of the output layer is about 43 and the chain has a factor 1 300 of headroom below
// the e5m2 maximum (57 344; conversions saturate there), while e5m2's 32 binades reach down to
// gradients 2^-30 of that size.
inline float dz_scale_for(int64_t R) { return ldexpf(1.0f, (int)ceil(log2((double)R)) + 6); }

constexpr int64_t NK_PAIR_GEMM_MAX_POINTS = 1 << 20;

inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

int layout(const nk_arch* ac, const nk_arch* af, const nk_render_opts* o, int64_t R, int training, WsLayout* L) {
    if (!ac || !o || R <= 0) NK_FAIL(NK_ERR_ARG, "render workspace: null pointer or R <= 0");
    const int Nc = o->num_coarse, Nf = o->num_fine, N = Nc + Nf;
    if (Nc < 2 || Nf < 0) NK_FAIL(NK_ERR_ARG, "render workspace: num_coarse=%d num_fine=%d", Nc, Nf);
    if (Nf > 0 && !af) NK_FAIL(NK_ERR_ARG, "render workspace: num_fine > 0 needs a fine network");
    const int fmt = training ? o->act_format : NK_ACT_BF16;
    if (fmt < NK_ACT_BF16 || fmt > NK_ACT_FP8_BF8) NK_FAIL(NK_ERR_ARG, "render workspace: unknown act_format %d", fmt);
    if (!(o->skip_empty_eps >= 0.0f)) NK_FAIL(NK_ERR_ARG, "render workspace: skip_empty_eps must be >= 0");
    int64_t off = 0;
    auto take = [&](int64_t bytes) { int64_t at = off; off = align256(off + bytes); return at; };
    L->z_c = take(R * Nc * 4);
    L->raw_c = take(R * Nc * 16);
    L->w_c = take(R * Nc * 4);
    L->rgb_c_tmp = take(R * 12);
    L->z_all = L->raw_f = L->live = L->act_c = L->act_f = L->d_raw = L->dz = L->dz_c = -1;
    if (Nf > 0) { L->z_all = take(R * N * 4); L->raw_f = take(R * N * 16); }
    // skip_empty_eps: a counter (first 256 bytes) and the list of the fine groups to evaluate
    if (skips_empty(o, training)) L->live = take(256 + nk_groups(R * (int64_t)N) * 4);
    if (training) {
        int64_t b = nk_act_bytes_fmt(ac, R, Nc, fmt);
        if (b < 0) return (int)b;
        L->act_c = take(b);
        int64_t dzb = nk_dz_bytes_fmt(ac, R, Nc, fmt);
        if (dzb < 0) return (int)dzb;
        if (Nf > 0) {
            b = nk_act_bytes_fmt(af, R, N, fmt);
            if (b < 0) return (int)b;
            L->act_f = take(b);
            const int64_t d2 = nk_dz_bytes_fmt(af, R, N, fmt);
            if (d2 < 0) return (int)d2;
            if (d2 > dzb) dzb = d2;
        }
        L->d_raw = take(R * (Nf > 0 ? N : Nc) * 16);
        L->dz = take(dzb);
        // Small batches: the coarse network gets a dZ region of its own, so that the weight-gradient
        // GEMMs of both networks can run in one shared launch after both chains (one network's
        // products alone leave most of the chip idle).  Large batches reuse one region in turn.
        if (Nf > 0 && R * (int64_t)N <= NK_PAIR_GEMM_MAX_POINTS) L->dz_c = take(nk_dz_bytes_fmt(ac, R, Nc, fmt));
    }
    L->total = off;
    return NK_OK;
}

// Profiling hook: events of the calling thread, recorded at the stage boundaries of the next
// nk_render_rays_fwd / _bwd calls (see nk_set_stage_events in the header).
thread_local hipEvent_t g_stage_events[NK_STAGE_EVENT_COUNT];
thread_local int g_stage_event_count = 0;

int mark(int idx, void* stream) {
    if (idx >= g_stage_event_count || !g_stage_events[idx]) return NK_OK;
    NK_CHECK_HIP(hipEventRecord(g_stage_events[idx], nk_stream(stream)));
    return NK_OK;
}

enum { NK_MAX_DEVICES = 64 };
std::mutex g_scratch_lock;
void* g_scratch_ptr[NK_MAX_DEVICES];       // per device, filled on the first call with that device current

// Address of `scratch` on the current device (looked up once per device), or nullptr with the
// error string set.
void* static_scratch() {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { nk_set_error("hipGetDevice failed: %s", hipGetErrorString(e)); return nullptr; }
    if (dev < 0 || dev >= NK_MAX_DEVICES) { nk_set_error("device %d outside 0..%d", dev, NK_MAX_DEVICES - 1); return nullptr; }
    std::lock_guard<std::mutex> guard(g_scratch_lock);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            nk_set_error("hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
            return nullptr;
        }
        g_scratch_ptr[dev] = p;
    }
    return g_scratch_ptr[dev];
}

}  // namespace

extern "C" int nk_set_stage_events(void** events, int count) {
    if (count < 0 || count > NK_STAGE_EVENT_COUNT || (count > 0 && !events))
        NK_FAIL(NK_ERR_ARG, "nk_set_stage_events: count %d outside 0..%d, or null array", count, (int)NK_STAGE_EVENT_COUNT);
    for (int i = 0; i < count; ++i) g_stage_events[i] = (hipEvent_t)events[i];
    g_stage_event_count = count;
    return NK_OK;
}

extern "C" int64_t nk_render_workspace_bytes(const nk_arch* coarse, const nk_arch* fine, const nk_render_opts* opts,
                                             int64_t R, int training) {
    WsLayout L;
    int rc = layout(coarse, fine, opts, R, training, &L);
    return rc < 0 ? rc : L.total;
}

extern "C" int nk_render_rays_fwd(const nk_arch* ac, const void* pc, const nk_arch* af, const void* pf,
                                  const float* ray_batch, int64_t R, const nk_render_opts* o, int training,
                                  void* workspace, int64_t workspace_bytes, const nk_render_out* out, void* stream) {
    if (!ac || !pc || !ray_batch || !o || !workspace || !out) NK_FAIL(NK_ERR_ARG, "nk_render_rays_fwd: null pointer");
    if (((uintptr_t)workspace & 255) != 0) NK_FAIL(NK_ERR_ARG, "nk_render_rays_fwd: workspace must be 256-byte aligned");
    WsLayout L;
    int rc = layout(ac, af, o, R, training, &L);
    if (rc < 0) return rc;
    if (workspace_bytes < L.total)
        NK_FAIL(NK_ERR_WORKSPACE, "nk_render_rays_fwd: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)L.total);
    const int Nc = o->num_coarse, Nf = o->num_fine, N = Nc + Nf;
    if (Nf > 0 && (!pf || !out->rgb_fine)) NK_FAIL(NK_ERR_ARG, "nk_render_rays_fwd: num_fine > 0 needs packed_fine and rgb_fine");
    if (Nf == 0 && !out->rgb_coarse) NK_FAIL(NK_ERR_ARG, "nk_render_rays_fwd: coarse-only render needs rgb_coarse");
    char* ws = (char*)workspace;
    float* z_c = (float*)(ws + L.z_c); float* raw_c = (float*)(ws + L.raw_c); float* w_c = (float*)(ws + L.w_c);
    float* rgb_c = out->rgb_coarse ? out->rgb_coarse : (float*)(ws + L.rgb_c_tmp);

    if ((rc = nk_stratified_z(ray_batch, R, o, z_c, stream)) < 0) return rc;
    if ((rc = mark(NK_EV_FWD_COARSE_BEGIN, stream)) < 0) return rc;
    const int fmt = training ? o->act_format : NK_ACT_BF16;
    if ((rc = nk_mlp_forward_fmt(ac, pc, ray_batch, z_c, R, Nc, raw_c, training ? ws + L.act_c : nullptr, fmt, nullptr, nullptr,
                                 stream)) < 0) return rc;
    if ((rc = mark(NK_EV_FWD_COARSE_END, stream)) < 0) return rc;
    if ((rc = nk_raw2outputs(raw_c, z_c, ray_batch, R, Nc, o, NK_STREAM_NOISE_COARSE, rgb_c, out->disp_coarse,
                             out->acc_coarse, out->depth_coarse, w_c, stream)) < 0) return rc;
    if (Nf > 0) {
        float* z_all = (float*)(ws + L.z_all); float* raw_f = (float*)(ws + L.raw_f);
        if ((rc = nk_sample_pdf(z_c, w_c, R, o, z_all, nullptr, stream)) < 0) return rc;
        const int32_t* live = nullptr;
        const int32_t* n_live = nullptr;
        if (L.live >= 0) {
            // groups of 32 fine samples near which the coarse pass saw nothing get zero-density rows
            // here and are left out of the fine network's launch
            n_live = (const int32_t*)(ws + L.live); live = (const int32_t*)(ws + L.live + 256);
            if ((rc = nk_mark_live_groups(z_c, w_c, z_all, R, Nc, N, o->skip_empty_eps, (int32_t*)live, (int32_t*)n_live, raw_f,
                                          stream)) < 0) return rc;
        }
        if ((rc = mark(NK_EV_FWD_FINE_BEGIN, stream)) < 0) return rc;
        if ((rc = nk_mlp_forward_fmt(af, pf, ray_batch, z_all, R, N, raw_f, training ? ws + L.act_f : nullptr, fmt, live, n_live,
                                     stream)) < 0) return rc;
        if ((rc = mark(NK_EV_FWD_FINE_END, stream)) < 0) return rc;
        if ((rc = nk_raw2outputs(raw_f, z_all, ray_batch, R, N, o, NK_STREAM_NOISE_FINE, out->rgb_fine, out->disp_fine,
                                 out->acc_fine, out->depth_fine, nullptr, stream)) < 0) return rc;
    }
    return NK_OK;
}

extern "C" int nk_render_rays_bwd(const nk_arch* ac, const void* pc, const nk_arch* af, const void* pf,
                                  const float* ray_batch, int64_t R, const nk_render_opts* o, void* workspace,
                                  int64_t workspace_bytes, const float* d_rgb_fine, const float* d_rgb_coarse,
                                  float* grad_coarse, float* grad_fine, void* stream) {
    if (!ac || !pc || !ray_batch || !o || !workspace) NK_FAIL(NK_ERR_ARG, "nk_render_rays_bwd: null pointer");
    WsLayout L;
    int rc = layout(ac, af, o, R, 1, &L);
    if (rc < 0) return rc;
    if (workspace_bytes < L.total)
        NK_FAIL(NK_ERR_WORKSPACE, "nk_render_rays_bwd: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)L.total);
    const int Nc = o->num_coarse, Nf = o->num_fine, N = Nc + Nf;
    char* ws = (char*)workspace;
    float* d_raw = (float*)(ws + L.d_raw);
    hipStream_t st = nk_stream(stream);
    const bool fine = Nf > 0 && d_rgb_fine, coarse = d_rgb_coarse != nullptr;
    if (fine && (!pf || !grad_fine)) NK_FAIL(NK_ERR_ARG, "nk_render_rays_bwd: fine gradient requested without packed_fine / grad_fine");
    if (coarse && !grad_coarse) NK_FAIL(NK_ERR_ARG, "nk_render_rays_bwd: coarse gradient requested without grad_coarse");
    const float dz_scale = o->act_format == NK_ACT_FP8_BF8 ? dz_scale_for(R) : 1.0f;
    const NkBwdJob jf = {af, pf, ws + L.act_f, d_raw, R, N, ws + L.dz, grad_fine, o->act_format, dz_scale};
    const NkBwdJob jc = {ac, pc, ws + L.act_c, d_raw, R, Nc, ws + (L.dz_c >= 0 ? L.dz_c : L.dz), grad_coarse, o->act_format, dz_scale};
    const bool together = fine && coarse && L.dz_c >= 0;
    if (fine) {
        if ((rc = nk_raw2outputs_bwd((float*)(ws + L.raw_f), (float*)(ws + L.z_all), ray_batch, R, N, o,
                                     NK_STREAM_NOISE_FINE, d_rgb_fine, nullptr, nullptr, d_raw, stream)) < 0) return rc;
        if ((rc = mark(NK_EV_BWD_FINE_CHAIN_BEGIN, stream)) < 0) return rc;
        if ((rc = nk_mlp_backward_chain(jf, st)) < 0) return rc;
        if ((rc = mark(NK_EV_BWD_FINE_CHAIN_END, stream)) < 0) return rc;
        if (!together && (rc = nk_mlp_backward_gemms(&jf, 1, st)) < 0) return rc;
        if ((rc = mark(NK_EV_BWD_FINE_GEMMS_END, stream)) < 0) return rc;
    }
    if (coarse) {
        if ((rc = nk_raw2outputs_bwd((float*)(ws + L.raw_c), (float*)(ws + L.z_c), ray_batch, R, Nc, o,
                                     NK_STREAM_NOISE_COARSE, d_rgb_coarse, nullptr, nullptr, d_raw, stream)) < 0) return rc;
        if ((rc = mark(NK_EV_BWD_COARSE_CHAIN_BEGIN, stream)) < 0) return rc;
        if ((rc = nk_mlp_backward_chain(jc, st)) < 0) return rc;
        if ((rc = mark(NK_EV_BWD_COARSE_CHAIN_END, stream)) < 0) return rc;
        if (!together && (rc = nk_mlp_backward_gemms(&jc, 1, st)) < 0) return rc;
    }
    if (together) {
        const NkBwdJob both[2] = {jf, jc};
        if ((rc = nk_mlp_backward_gemms(both, 2, st)) < 0) return rc;
    }
    return mark(NK_EV_BWD_END, stream);
}

extern "C" int nk_render_rays_fwd_noalloc(const nk_arch* ac, const void* pc, const nk_arch* af, const void* pf,
                                          const float* ray_batch, int64_t R, const nk_render_opts* o,
                                          const nk_render_out* out, void* stream) {
    WsLayout L;
    int rc = layout(ac, af, o, R, 0, &L);
    if (rc < 0) return rc;
    if (L.total > (int64_t)NK_NOALLOC_SCRATCH_BYTES)
        NK_FAIL(NK_ERR_WORKSPACE, "nk_render_rays_fwd_noalloc: %lld rays x (%d + %d) samples need %lld bytes of scratch, "
                "the built-in buffer holds %lld; render in smaller pieces or use nk_render_rays_fwd",
                (long long)R, o->num_coarse, o->num_fine, (long long)L.total, (long long)NK_NOALLOC_SCRATCH_BYTES);
    void* ws = static_scratch();
    if (!ws) return NK_ERR_HIP;
    // Every region of the inference layout is written in full by the stage that produces it before
    // a later stage reads it, so nothing of the previous call's contents has to be cleared.
    return nk_render_rays_fwd(ac, pc, af, pf, ray_batch, R, o, 0, ws, L.total, out, stream);
}
