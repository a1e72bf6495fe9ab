// NeRF MLP backward for gfx950: a fused backward chain (dL/d activation through every layer,
// activations masked and kept in registers exactly as in the forward kernel, transposed weight
// stream through LDS) that writes each layer's dZ as bf16, and a split-K MFMA GEMM that turns
// (dZ, saved layer input) pairs into weight and bias gradients.
//
// Saved tensors all use one layout, [group of 32 points][feature octet][point][8 features] bf16,
// which is what a wavefront's converted accumulators store with 8-byte pieces and what
// ds_read_b64_tr_b16 turns into MFMA operands whose k index runs over points.
#include "nk_common.h"
#include "nk_mfma.h"

// --------------------------------------------------------------------------------------------
// backward chain
// --------------------------------------------------------------------------------------------
struct BwdArgs {
    NkPlan plan;
    const char* packed_t;   // transposed weight stream
    const char* act;        // saved activations
    const float* d_raw;     // [P,4]
    char* dz;               // dZ regions
    int64_t P, G;
};

#include "nk_bwd_chain.inc"


// --------------------------------------------------------------------------------------------
// dW GEMM: C[m, n] += sum_p A[p, m] * B[p, n]  (A = dZ region, B = saved layer-input region)
// --------------------------------------------------------------------------------------------
enum { NK_COL_IDENT = 0, NK_COL_PEX = 1, NK_COL_PED = 2 };

struct GemmArgs {
    const char* A; const char* B;   // regions [G][MT*4][32][8], [G][NT*4][32][8] bf16
    int MT, NT;                     // 32-wide tiles of A features / B features
    int WM, WN;                     // wave grid (WM*WN <= 8); wave tile = (MT/WM) x (NT/WN) tiles
    int64_t n_stages;               // G (a stage = 1 group = 32 points)
    int64_t stages_per_wg;
    float* grad;                    // flat gradient
    int w_off, ld;                  // destination matrix origin and leading dimension
    int row_first, row_count, row_dst0;
    int colmode, col_base, col_count, nfreq;
    int b_off;                      // bias gradient origin, or -1
};

// Transposed LDS reads are issued from inline asm: hipcc drains vmcnt before a
// ds_read_b64_tr_b16 INTRINSIC that follows an LDS-DMA load (it does not for plain LDS loads),
// which would serialise every stage's loads with its MFMAs.  The price is that the waits for the
// read results are written by hand too (tr_wait).
struct TrFrag { u32x2 lo, hi; };

// issue the two 8-byte transposed reads of one MFMA operand: 32 "rows" = features 32*tile..,
// k index = point 16*ks..; `piece` is the LDS byte address of the [octet][point][8] image
NK_D void tr_issue(TrFrag& f, uint32_t piece, int tile, int ks, int lane) {
    const int g4 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int o = 4 * tile + 2 * (g4 & 1) + (pp >> 1);
    const int sw = (o & 3) << 2;
    const int c0 = 16 * ks + 8 * (g4 >> 1) + q;
    const uint32_t base = piece + o * 512 + (pp & 1) * 8;
    const uint32_t a0 = base + ((c0) ^ sw) * 16, a1 = base + ((c0 + 4) ^ sw) * 16;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a1));
}
template <int N>
NK_D void tr_wait(TrFrag (&f)[N]) {           // results of every read issued so far have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i].lo), "+v"(f[i].hi));
}
NK_D bf16x8 tr_value(const TrFrag& f) {
    const u32x4 u = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return __builtin_bit_cast(bf16x8, u);
}

// A stage is one group of 32 points: A piece (MT*2 KiB) then B piece (NT*2 KiB).  Stages go
// through a ring of NK_GEMM_RING LDS slots with three stages in flight, because one CU needs
// ~50 KB of loads outstanding at all times to keep its share of HBM bandwidth busy.  Order per
// stage s: counted wait (stage s landed; the loads of s+1, s+2 stay in flight) -> barrier (also
// retires every wave's reads of slot (s-1) % RING) -> issue stage s+3 into that slot -> MFMAs.
enum { NK_GEMM_RING = 4 };

template <int N>
NK_D void gemm_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(512, 1) void k_dw_gemm(GemmArgs a) {
    constexpr int TM = MT / WM, TN = NT / WN;       // wave tile, in 32x32 tiles
    static_assert(TM * WM == MT && TN * WN == NT && WM * WN <= 8, "wave grid must tile the output");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int stage_bytes = (MT + NT) * 2048;       // one group's A piece then B piece
    constexpr int blocks_per_stage = (MT + NT) * 2;
    constexpr int min_pieces = blocks_per_stage / 8;    // every wave issues at least this many loads per stage
    const bool active = wave < WM * WN;
    const int wm = active ? wave / WN : 0, wn = active ? wave % WN : 0;
    const int64_t s0 = (int64_t)blockIdx.x * a.stages_per_wg;
    int64_t s1 = s0 + a.stages_per_wg;
    if (s1 > a.n_stages) s1 = a.n_stages;
    if (s0 >= s1) return;

    auto issue = [&](int64_t stage) {
        char* dst = lds + (int)(stage % NK_GEMM_RING) * stage_bytes;
#pragma unroll
        for (int i = 0; i < (blocks_per_stage + 7) / 8; ++i) {      // compile-time trip count: the compiler
            const int b = wave + 8 * i;                             // would drain vmcnt after a real loop
            if (b >= blocks_per_stage) break;
            const bool isA = b < MT * 2;
            const int pb = isA ? b : b - MT * 2;            // 1-KiB block inside the piece
            const char* src = isA ? a.A + stage * MT * 2048 : a.B + stage * NT * 2048;
            const int o = 2 * pb + (lane >> 5), cpos = lane & 31;
            const int csrc = cpos ^ ((o & 3) << 2);           // swizzle on the source side
            __builtin_amdgcn_global_load_lds((const NK_GLB void*)(src + (o * 32 + csrc) * 16),
                                             (NK_LDS void*)(dst + b * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float bsum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) bsum[i] = 0.0f;
    bf16x2 ones; ones[0] = (__bf16)1.0f; ones[1] = (__bf16)1.0f;
    const bool want_bias = a.b_off >= 0;

    const uint32_t lds0 = (uint32_t)(uintptr_t)(NK_LDS char*)lds;
    for (int64_t s = s0; s < s0 + 3 && s < s1; ++s) issue(s);
    for (int64_t s = s0; s < s1; ++s) {
        // loads issued after stage s: those of the stages s+1, s+2 that exist
        const int64_t left = s1 - 1 - s;
        if (left >= 2) gemm_wait_barrier<2 * min_pieces>();
        else if (left == 1) gemm_wait_barrier<min_pieces>();
        else gemm_wait_barrier<0>();
        if (s + 3 < s1) issue(s + 3);
        if (active) {
            const uint32_t pa = lds0 + (uint32_t)(s % NK_GEMM_RING) * stage_bytes;
            const uint32_t pb = pa + MT * 2048;
            TrFrag fa[2][TM], fb[2][TN];        // both k-steps of the stage: the second is read under the first's MFMAs
#pragma unroll
            for (int i = 0; i < TM; ++i) tr_issue(fa[0][i], pa, wm * TM + i, 0, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) tr_issue(fb[0][j], pb, wn * TN + j, 0, lane);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                tr_wait(fa[ks]); tr_wait(fb[ks]);
                if (ks == 0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) tr_issue(fa[1][i], pa, wm * TM + i, 1, lane);
#pragma unroll
                    for (int j = 0; j < TN; ++j) tr_issue(fb[1][j], pb, wn * TN + j, 1, lane);
                }
                bf16x8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = tr_value(fa[ks][i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = tr_value(fb[ks][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                // bias gradient: the WN waves of a row share it, each takes its k-steps
                if (want_bias && ((2 * (int)(s & 1) + ks) % WN) == wn) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            bf16x2 pr; pr[0] = af[i][2 * w]; pr[1] = af[i][2 * w + 1];
                            bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, bsum[i], false);
                        }
                    }
                }
            }
        }
    }
    if (!active) return;
    // epilogue: atomically add this K-slice's tiles into the flat gradient
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = 32 * (wn * TN + j) + r;
        int dcol = -1;
        if (a.colmode == NK_COL_IDENT) dcol = n < a.col_count ? a.col_base + n : -1;
        else {
            const int s = n >> 4, hh = (n >> 3) & 1, jj = n & 7;
            const int cc = a.colmode == NK_COL_PEX ? (s < 4 ? nk_pe_xyz_col(8 * s + jj, hh, a.nfreq) : -1)
                                                   : (s < 2 ? nk_pe_dir_col(8 * s + jj, hh, a.nfreq) : -1);
            dcol = cc >= 0 ? a.col_base + cc : -1;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = 32 * (wm * TM + i) + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int mr = m - a.row_first;
                if (dcol >= 0 && mr >= 0 && mr < a.row_count)
                    atomicAdd(a.grad + a.w_off + (int64_t)(a.row_dst0 + mr) * a.ld + dcol, acc[i][j][e]);
            }
        }
    }
    if (want_bias) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float tot = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            const int m = 32 * (wm * TM + i) + r, mr = m - a.row_first;
            if (h == 0 && mr >= 0 && mr < a.row_count) atomicAdd(a.grad + a.b_off + a.row_dst0 + mr, tot);
        }
    }
}

static int launch_gemm(const NkPlan& p, const char* A, int MA, const char* B, int NB, int64_t G, float* grad,
                       int w_off, int ld, int row_first, int row_count, int row_dst0, int colmode, int col_base,
                       int col_count, int nfreq, int b_off, hipStream_t stream) {
    GemmArgs g;
    g.A = A; g.B = B; g.MT = MA / 32; g.NT = NB / 32;
    g.n_stages = G;
    // Split-K width: one workgroup per CU for large batches; fewer when a workgroup would see less
    // than ~48 stages, because every workgroup pays the same atomic epilogue (one output matrix).
    static const int env_wgs = []() { const char* e = getenv("NK_GEMM_WGS"); return e ? atoi(e) : 0; }();
    int64_t target_wgs = env_wgs > 0 ? env_wgs : g.n_stages / 48;
    if (env_wgs <= 0) target_wgs = target_wgs < 64 ? 64 : (target_wgs > 256 ? 256 : target_wgs);
    g.stages_per_wg = (g.n_stages + target_wgs - 1) / target_wgs;
    if (g.stages_per_wg < 1) g.stages_per_wg = 1;
    g.grad = grad; g.w_off = w_off; g.ld = ld; g.row_first = row_first; g.row_count = row_count; g.row_dst0 = row_dst0;
    g.colmode = colmode; g.col_base = col_base; g.col_count = col_count; g.nfreq = nfreq; g.b_off = b_off;
    const unsigned nwg = (unsigned)((g.n_stages + g.stages_per_wg - 1) / g.stages_per_wg);
    const int lds_bytes = NK_GEMM_RING * (g.MT + g.NT) * 2048;
    const dim3 grid(nwg), block(512);
    // (A tiles, B tiles) -> wave grid; every shape the supported networks produce
#define NK_GEMM_CASE(MT_, NT_, WM_, WN_) \
    if (g.MT == MT_ && g.NT == NT_) { \
        g.WM = WM_; g.WN = WN_; \
        NK_CHECK_HIP(hipFuncSetAttribute((const void*)k_dw_gemm<MT_, NT_, WM_, WN_>, \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); \
        hipLaunchKernelGGL((k_dw_gemm<MT_, NT_, WM_, WN_>), grid, block, lds_bytes, stream, g); \
    } else
    NK_GEMM_CASE(8, 8, 2, 4) NK_GEMM_CASE(8, 2, 4, 2) NK_GEMM_CASE(4, 8, 2, 4) NK_GEMM_CASE(4, 4, 2, 4)
    NK_GEMM_CASE(1, 4, 1, 4) NK_GEMM_CASE(1, 8, 1, 8) NK_GEMM_CASE(4, 1, 4, 1)
    NK_GEMM_CASE(4, 2, 4, 2) NK_GEMM_CASE(2, 4, 2, 4) NK_GEMM_CASE(1, 2, 1, 2) NK_GEMM_CASE(2, 1, 2, 1)
    NK_FAIL(NK_ERR_UNSUPPORTED, "dW GEMM: no kernel for %d x %d tiles", g.MT, g.NT);
#undef NK_GEMM_CASE
    NK_CHECK_LAUNCH("k_dw_gemm");
    (void)p;
    return NK_OK;
}

extern "C" int64_t nk_dz_bytes(const nk_arch* arch, int64_t R, int32_t n) {
    NkPlan p;
    int rc = nk_build_plan(arch, &p);
    if (rc < 0) return rc;
    if (R <= 0 || n <= 0) { nk_set_error("nk_dz_bytes: non-positive size"); return NK_ERR_ARG; }
    return nk_dz_offset(p, nk_dz_count(p), nk_groups(R * (int64_t)n));
}

extern "C" int nk_mlp_backward(const nk_arch* arch, const void* packed, const void* act_save, const float* d_raw,
                               int64_t R, int32_t n, void* dz_scratch, float* grad, void* stream) {
    if (!packed || !act_save || !d_raw || !dz_scratch || !grad || R <= 0 || n <= 0)
        NK_FAIL(NK_ERR_ARG, "nk_mlp_backward: null pointer or non-positive size");
    if (((uintptr_t)packed & 15) || ((uintptr_t)act_save & 15) || ((uintptr_t)dz_scratch & 15) || ((uintptr_t)d_raw & 15))
        NK_FAIL(NK_ERR_ARG, "nk_mlp_backward: buffers must be 16-byte aligned");
    BwdArgs a;
    int rc = nk_build_plan(arch, &a.plan);
    if (rc < 0) return rc;
    const NkPlan& p = a.plan;
    a.packed_t = (const char*)packed + (int64_t)p.fwd_blocks * 1024;
    a.act = (const char*)act_save; a.d_raw = d_raw; a.dz = (char*)dz_scratch;
    a.P = R * (int64_t)n; a.G = nk_groups(a.P);
    hipStream_t st = nk_stream(stream);
    const int nw = nk_waves_per_wg();
    const dim3 grid((unsigned)(a.G / nw)), block(nw * 64);
    if (p.W == 256) {
        if (nw == 8) hipLaunchKernelGGL((k_mlp_bwd<256, 128, 8>), grid, block, 3 * (16 + 1) * 1024, st, a);
        else hipLaunchKernelGGL((k_mlp_bwd<256, 128, 4>), grid, block, 3 * (16 + 1) * 1024, st, a);
    } else {
        if (nw == 8) hipLaunchKernelGGL((k_mlp_bwd<128, 64, 8>), grid, block, 3 * (8 + 1) * 1024, st, a);
        else hipLaunchKernelGGL((k_mlp_bwd<128, 64, 4>), grid, block, 3 * (8 + 1) * 1024, st, a);
    }
    NK_CHECK_LAUNCH("k_mlp_bwd");

    // weight / bias gradients, one GEMM per (dZ, layer input) pair
    const int D = p.D, W = p.W, WD = p.WD;
    const int64_t G = a.G;
    auto act = [&](int idx) { return a.act + nk_act_offset(p, idx, G); };
    auto dz = [&](int idx) { return (const char*)a.dz + nk_dz_offset(p, idx, G); };
    const NkLayer& lrgb = p.layer[D + 3];
    const NkLayer& lviews = p.layer[D + 2];
    const NkLayer& lfeat = p.layer[D + 1];
    const NkLayer& lalpha = p.layer[D];
    // rgb: rows 0..2 of dOut against g
    rc = launch_gemm(p, dz(0), 32, act(3 + D), WD, G, grad, lrgb.w_off, lrgb.in_dim, 0, 3, 0, NK_COL_IDENT, 0, WD, 0,
                     lrgb.b_off, st);
    if (rc < 0) return rc;
    // alpha: row 3 of dOut against h_{D-1}
    rc = launch_gemm(p, dz(0), 32, act(2 + D - 1), W, G, grad, lalpha.w_off, lalpha.in_dim, 3, 1, 0, NK_COL_IDENT, 0, W,
                     0, lalpha.b_off, st);
    if (rc < 0) return rc;
    // views: dZ_views against cat[feat, gamma(d)]
    rc = launch_gemm(p, dz(1), WD, act(2 + D), W, G, grad, lviews.w_off, lviews.in_dim, 0, WD, 0, NK_COL_IDENT, 0, W, 0,
                     lviews.b_off, st);
    if (rc < 0) return rc;
    rc = launch_gemm(p, dz(1), WD, act(1), 32, G, grad, lviews.w_off, lviews.in_dim, 0, WD, 0, NK_COL_PED, W, 32, p.Ld,
                     -1, st);
    if (rc < 0) return rc;
    // feat: dZ_feat against h_{D-1}
    rc = launch_gemm(p, dz(2), W, act(2 + D - 1), W, G, grad, lfeat.w_off, lfeat.in_dim, 0, W, 0, NK_COL_IDENT, 0, W, 0,
                     lfeat.b_off, st);
    if (rc < 0) return rc;
    // trunk
    for (int L = D - 1; L >= 0; --L) {
        const NkLayer& l = p.layer[L];
        const char* dzl = dz(3 + D - 1 - L);
        if (L == 0) {
            rc = launch_gemm(p, dzl, W, act(0), 64, G, grad, l.w_off, l.in_dim, 0, W, 0, NK_COL_PEX, 0, 64, p.Lx, l.b_off, st);
        } else if (p.skip >= 0 && L == p.skip + 1) {
            rc = launch_gemm(p, dzl, W, act(0), 64, G, grad, l.w_off, l.in_dim, 0, W, 0, NK_COL_PEX, 0, 64, p.Lx, -1, st);
            if (rc < 0) return rc;
            rc = launch_gemm(p, dzl, W, act(2 + L - 1), W, G, grad, l.w_off, l.in_dim, 0, W, 0, NK_COL_IDENT,
                             3 + 6 * p.Lx, W, 0, l.b_off, st);
        } else {
            rc = launch_gemm(p, dzl, W, act(2 + L - 1), W, G, grad, l.w_off, l.in_dim, 0, W, 0, NK_COL_IDENT, 0, W, 0,
                             l.b_off, st);
        }
        if (rc < 0) return rc;
    }
    return NK_OK;
}
