// Device helpers shared by the MLP kernels: MFMA operand fragments, bf16 packing, the
// positional-encoding fragments, the LDS weight stream and the activation tile store.
#pragma once
#include "nk_common.h"

union Frag { u32x4 u; bf16x8 b; };

NK_D uint32_t pack_bf16(float lo, float hi) {
    bf16x2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// sin / cos of x * 2^l with a two-constant reduction to one revolution, then v_sin / v_cos
// (which take revolutions).  Absolute error ~1e-6, far below the bf16 operand rounding.
NK_D void sincos_rev(float x, float& s, float& c) {
    const float C_HI = 0.15915494f, C_LO = 3.0994494e-9f;      // 1/(2 pi) split
    const float n = rintf(x * C_HI);
    float r = fmaf(x, C_HI, -n);
    r = fmaf(x, C_LO, r);
    s = __builtin_amdgcn_sinf(r);
    c = __builtin_amdgcn_cosf(r);
}

// Two fp32 values -> two OCP e4m3 bytes in the low (hi = false) or high half of `old`; rounds to
// nearest even, and with MODE.FP16_OVFL set (nk_fp8_saturate) values beyond +-448 give +-448.
NK_D uint32_t pk_fp8(float a, float b, uint32_t old, bool hi) {
    return (uint32_t)(hi ? __builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, true)
                         : __builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, false));
}
// ... and OCP e5m2 (saturates at +-57344 under the same mode bit)
NK_D uint32_t pk_bf8(float a, float b, uint32_t old, bool hi) {
    return (uint32_t)(hi ? __builtin_amdgcn_cvt_pk_bf8_f32(a, b, (int)old, true)
                         : __builtin_amdgcn_cvt_pk_bf8_f32(a, b, (int)old, false));
}
// MODE.FP16_OVFL = 1 for the rest of the wave's life: 8-bit float conversions saturate instead of
// producing NaN (e4m3) or infinity (e5m2).  hwreg(MODE, offset 23, width 1).
NK_D void nk_fp8_saturate() { __builtin_amdgcn_s_setreg(1 | (23 << 6), 1); }

// gamma(x) as 4 B-operand fragments (k-slots per nk_pe_xyz_col); p8 (optional): the same 32 slots
// as e4m3 bytes, dwords 2s, 2s+1 = k-step s
NK_D void pe_xyz_frags(const NkPlan& p, float px, float py, float pz, int h, Frag (&f)[4], uint32_t* p8 = nullptr) {
    float v[32];
    const float base = h ? 32.0f : 1.0f;
#pragma unroll
    for (int fi = 0; fi < 5; ++fi) {
        const float sc = nk_enc_freq(p.lin_x, p.step_x, p.end_x, p.Lx, 5 * h + fi, base * (float)(1 << fi));
        sincos_rev(px * sc, v[6 * fi + 0], v[6 * fi + 3]);
        sincos_rev(py * sc, v[6 * fi + 1], v[6 * fi + 4]);
        sincos_rev(pz * sc, v[6 * fi + 2], v[6 * fi + 5]);
    }
    v[30] = h ? pz : px;
    v[31] = h ? 0.0f : py;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int w = 0; w < 4; ++w) f[s].u[w] = pack_bf16(v[8 * s + 2 * w], v[8 * s + 2 * w + 1]);
    if (p8) {
#pragma unroll
        for (int d = 0; d < 8; ++d) p8[d] = pk_fp8(v[4 * d + 2], v[4 * d + 3], pk_fp8(v[4 * d], v[4 * d + 1], 0, false), true);
    }
}

// gamma(d) as 2 fragments (k-slots per nk_pe_dir_col); p8 (optional): e4m3 bytes, dwords 2s, 2s+1 = k-step s
NK_D void pe_dir_frags(const NkPlan& p, float dx, float dy, float dz, int h, Frag (&f)[2], uint32_t* p8 = nullptr) {
    float v[16];
    const float base = h ? 4.0f : 1.0f;
#pragma unroll
    for (int fi = 0; fi < 2; ++fi) {
        const float sc = nk_enc_freq(p.lin_d, p.step_d, p.end_d, p.Ld, 2 * h + fi, base * (float)(1 << fi));
        sincos_rev(dx * sc, v[6 * fi + 0], v[6 * fi + 3]);
        sincos_rev(dy * sc, v[6 * fi + 1], v[6 * fi + 4]);
        sincos_rev(dz * sc, v[6 * fi + 2], v[6 * fi + 5]);
    }
    v[12] = h ? dz : dx;
    v[13] = h ? 0.0f : dy;
    v[14] = 0.0f; v[15] = 0.0f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int w = 0; w < 4; ++w) f[s].u[w] = pack_bf16(v[8 * s + 2 * w], v[8 * s + 2 * w + 1]);
    if (p8) {
#pragma unroll
        for (int d = 0; d < 4; ++d) p8[d] = pk_fp8(v[4 * d + 2], v[4 * d + 3], pk_fp8(v[4 * d], v[4 * d + 1], 0, false), true);
    }
}

// The weight stream: stage i sits in LDS ring slot i % 3 and is loaded (LDS-DMA, 16 B per lane)
// two stages before it is consumed.  All NW waves issue the loads (block b by wave b % NW) and
// all pass one barrier per stage, which both publishes the landed stage and retires the reads of
// the slot that the next issue overwrites.
template <int NW>
struct WStream {
    const char* g;          // next byte of the stream to load
    char* ring;             // LDS base
    int slot_bytes;
    int cur;                // ring slot holding the stage being consumed
    int nxt;                // ring slot the next issue fills
    int wave, lane;
    bool spread_ok = false; // NK_DMA_SPREAD: set by the kernel once the ring is primed
    // NK_DMA_SPREAD (forward chain): the pieces of a stage's stream load are issued one every third
    // k-step from inside the tile's MFMA loop instead of as a burst after the barrier, where all
    // waves of the CU queue for the address path at once (-3 % inference, -1 % with saves).
#ifndef NK_DMA_SPREAD
#define NK_DMA_SPREAD 1
#endif
    int pend_nb;            // NK_DMA_SPREAD: blocks of the load announced by issue_next, issued piece by piece
    // piece i of the announced load: block wave + i * NW (issued from inside the tile's MFMA loop)
    NK_D void piece(int i) {
        const int b = wave + i * NW;
        if (b < pend_nb) {
            char* dst = ring + nxt * slot_bytes;
            const uint32_t voff = lane * 16;
            const uint32_t l = (uint32_t)(uintptr_t)(NK_LDS char*)(dst + b * 1024);
            const char* sb = g + (int64_t)b * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(l), "v"(voff), "s"(sb) : "memory");
        }
    }
    NK_D void finish() {    // after the last piece
        g += (int64_t)pend_nb * 1024;
        nxt = nxt == 2 ? 0 : nxt + 1;
        pend_nb = 0;
    }
    NK_D void issue_next(int nb) {       // nb = 1-KiB blocks of the stage after next (0: none left)
        if (NK_DMA_SPREAD && spread_ok) { pend_nb = nb; return; }
        char* dst = ring + nxt * slot_bytes;
#ifndef NK_DMA_ASM
#define NK_DMA_ASM 1
#endif
#if NK_DMA_ASM
        // Issued from asm: hipcc books a global_load ... lds as a pending FLAT operation, and while
        // one is pending (it never sees the hand-written vmcnt waits) it turns every counted
        // lgkmcnt wait of the tile's LDS reads into lgkmcnt(0).
        const uint32_t voff = lane * 16;
        for (int b = wave; b < nb; b += NW) {
            const uint32_t l = (uint32_t)(uintptr_t)(NK_LDS char*)(dst + b * 1024);
            const char* sb = g + (int64_t)b * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(l), "v"(voff), "s"(sb) : "memory");     // (m0 is reserved: hipcc sets it itself before each use)
        }
#else
        for (int b = wave; b < nb; b += NW)
            __builtin_amdgcn_global_load_lds((const NK_GLB void*)(g + (int64_t)b * 1024 + lane * 16),
                                             (NK_LDS void*)(dst + b * 1024), 16, 0, 0);
#endif
        g += (int64_t)nb * 1024;
        nxt = nxt == 2 ? 0 : nxt + 1;
    }
    NK_D void start(int nb0, int nb1) {      // stages 0 and 1 loaded and visible
        cur = 0; nxt = 0;
        issue_next(nb0);
        issue_next(nb1);
        __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0)
        __syncthreads();
    }
    NK_D void advance() { cur = cur == 2 ? 0 : cur + 1; }
    // End of stage: wait until at most N vector-memory operations are outstanding, then barrier.
    // vmcnt retires in issue order, so with N = (operations issued after the NEXT stage's load:
    // the stores of the previous and of this stage, and this wave's share of the load issued in
    // this stage for the stage after next) the next stage has landed.  Passing a smaller N than
    // the true count only waits longer; a larger one would race.
    template <int N>
    NK_D void flip_keep() {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
        advance();
    }
    // Same, naming the tile's accumulator: its MFMAs are register-only instructions that hipcc
    // otherwise sinks below the barrier, stretching the live range of 16 weight fragments.
    template <int N>
    NK_D void flip_keep(f32x16& acc) {
        asm volatile("s_waitcnt vmcnt(%1) lgkmcnt(0)\n\ts_barrier" : "+v"(acc) : "n"(N) : "memory");
        advance();
    }
    NK_D const char* stage() const { return ring + cur * slot_bytes; }
};

// bias fragment of a stage (block index nk): 16 floats for this lane-half
NK_D f32x16 lds_bias(const char* stage, int nk, int h) {
    const f32x4* b = (const f32x4*)(stage + nk * 1024 + h * 64);
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = b[q];
        v[4 * q + 0] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
    }
    return v;
}

// One 32-row output tile: NK0 k-steps against fragments b0[], then NK1 against b1[].  The A
// fragments come from LDS with an explicit 8-deep read-ahead so the MFMA chain never waits on a
// just-issued ds_read.
template <int NK0, int NK1, int NPRE = 0, typename WS = void>   // NPRE: other LDS reads issued just before (the bias fragment)
NK_D f32x16 tile_run(f32x16 acc, const char* stage, int lane, const Frag* b0, const Frag* b1, WS* ws = nullptr) {
    constexpr int NK = NK0 + NK1;
#ifndef NK_LOOKAHEAD
#define NK_LOOKAHEAD 8
#endif
    constexpr int LA = NK < NK_LOOKAHEAD ? NK : NK_LOOKAHEAD;
    const char* base = stage + lane * 16;
    bf16x8 a[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) a[i] = *(const bf16x8*)(base + i * 1024);
    // pin the software pipeline: LA reads up front, then one MFMA / one read alternately
    // (mask 0x100 = DS read, 0x008 = MFMA); without this the scheduler sinks each read to its use
    __builtin_amdgcn_sched_group_barrier(0x100, LA + NPRE, 0);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const bf16x8 b = ks < NK0 ? b0[ks < NK0 ? ks : 0].b : b1[ks >= NK0 ? ks - NK0 : 0].b;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % LA], b, acc, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (!__is_same(WS, void)) {       // one piece of the announced stream load every third k-step
            if (NK_DMA_SPREAD && ks % 3 == 1 && ks / 3 < 6) ws->piece(ks / 3);
        }
        if (ks + LA < NK) {
            a[ks % LA] = *(const bf16x8*)(base + (ks + LA) * 1024);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
    if constexpr (!__is_same(WS, void)) {
        if (NK_DMA_SPREAD) {
#pragma unroll
            for (int i = (NK + 1) / 3; i < 6; ++i) ws->piece(i);    // pieces a short tile had no k-step for
            ws->finish();
        }
    }
    return acc;
}

// Streaming stores for saved activations / dZ: written once, read once by a later kernel, and at
// 5 KB per point they would otherwise push the weight stream out of L2.
#ifndef NK_NT_STORES
#define NK_NT_STORES 1
#endif
NK_D void st_nt(u32x2* p, u32x2 v) {
#if NK_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
NK_D void st_nt(u32x4* p, u32x4 v) {
#if NK_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// Pin two fragments: everything that computes them is emitted before this point (keeps the
// mask / convert of a finished tile ahead of the next stream load instead of being sunk below it).
NK_D void pin_frags(Frag& f0, Frag& f1) {
    asm volatile("" : "+v"(f0.u), "+v"(f1.u) : : "memory");
}

// store one bf16 activation tile: octets 4t..4t+3 of region `dst` ([G][NO][32][8] bf16)
NK_D void save_tile(char* dst, int64_t grp, int NO, int t, int c, int h, const Frag& f0, const Frag& f1) {
    // features 8q+4h+{0..3} of tile t are dwords (2q&3.., ) of fragment q>>1
    char* base = dst + ((grp * NO + 4 * t) * 32 + c) * 16 + 8 * h;
    st_nt((u32x2*)(base + 0 * 512), u32x2{f0.u[0], f0.u[1]});
    st_nt((u32x2*)(base + 1 * 512), u32x2{f0.u[2], f0.u[3]});
    st_nt((u32x2*)(base + 2 * 512), u32x2{f1.u[0], f1.u[1]});
    st_nt((u32x2*)(base + 3 * 512), u32x2{f1.u[2], f1.u[3]});
}

NK_D void acc_to_frags(const f32x16& a, bool relu, Frag& f0, Frag& f1);
NK_D void acc_to_frags8(const f32x16& a, bool relu, Frag& f0, Frag& f1, Frag& s8);

// max(x, 0) as ONE instruction, v_max_i32 on the bit pattern (negative floats are negative
// integers; -0 and NaNs with the sign bit set give +0, other NaNs pass).  fmaxf() costs two:
// hipcc first canonicalises x with v_max_f32 x, x, also behind __builtin_amdgcn_fmed3f; inline asm
// is no way out either, the hazard recogniser does not cover an asm operand read straight after
// the MFMA that wrote it.
NK_D float relu1(float x) {
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}

// Accumulator tile -> bf16 fragments (ReLU when `relu`), shifting the SIGN bit of each
// pre-activation into `mw`, one v_alignbit per register (first register ends up highest).
// The forward kernel stores ~mw for a ReLU layer (all ones otherwise) as the layer's
// pass-gradient word: a gradient passes where the pre-activation's sign bit is clear.  That is
// relu'(x) everywhere except at x == +0 exactly, where torch blocks the gradient.
// S8: also the tile as sixteen e4m3 bytes in s8 (byte e = accumulator register e, rounded once
// from the fp32 value after the ReLU) -- what the forward kernel saves under NK_ACT_FP8*.
template <bool MASK, bool S8 = false>
NK_D void acc_to_frags_mask(const f32x16& a, bool relu, Frag& f0, Frag& f1, uint32_t& mw, Frag* s8 = nullptr) {
    if (!MASK) { acc_to_frags(a, relu, f0, f1); return; }
    // (never __builtin_bit_cast a vector ELEMENT: hipcc reads the wrong lane of it)
    // registers are visited in index order 0..15 so that bit 31-k of the word belongs to register k
    uint32_t m = mw;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float x = a[i];
        m = __builtin_amdgcn_alignbit(m, (uint32_t)__float_as_int(x), 31);     // (m << 1) | sign(x)
    }
    asm volatile("" : "+v"(m));     // computed here: otherwise it is sunk to the layer's end and
    mw = m;                         // keeps every accumulator tile of the layer alive (spills)
    if constexpr (S8) acc_to_frags8(a, relu, f0, f1, *s8);
    else acc_to_frags(a, relu, f0, f1);
}

// ReLU of two packed bf16 values in one instruction: v_pk_max_i16 against 0 on the bit patterns
// (a negative bf16, -0 included, is a negative 16-bit integer).  Rounding to bf16 keeps the sign, so
// this equals rounding max(x, 0) bit for bit, at half the instructions of a ReLU per fp32 value.
// Measured: slower when each max directly follows its convert (inference 3.06 -> 3.23 ms), and
// within 0.3 % of the fp32 form (19.19 vs 19.25 ms per training step) with all converts issued
// first, so it stays off; kept as a switch.
#ifndef NK_RELU_PACKED
#define NK_RELU_PACKED 0
#endif
typedef __attribute__((ext_vector_type(2))) short s16x2;
NK_D uint32_t relu_pk_bf16(uint32_t p) {
    s16x2 v = __builtin_bit_cast(s16x2, p);
    v = __builtin_elementwise_max(v, s16x2{0, 0});
    return __builtin_bit_cast(uint32_t, v);
}

NK_D void acc_to_frags(const f32x16& a, bool relu, Frag& f0, Frag& f1) {
#if NK_RELU_PACKED
    // all eight converts first, then the eight packed maxima: no instruction waits on its neighbour
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        f0.u[w] = pack_bf16(a[2 * w], a[2 * w + 1]);
        f1.u[w] = pack_bf16(a[8 + 2 * w], a[8 + 2 * w + 1]);
    }
    if (relu) {
        asm volatile("" : "+v"(f0.u), "+v"(f1.u));
#pragma unroll
        for (int w = 0; w < 4; ++w) { f0.u[w] = relu_pk_bf16(f0.u[w]); f1.u[w] = relu_pk_bf16(f1.u[w]); }
    }
#else
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        float x0 = a[2 * w], x1 = a[2 * w + 1], y0 = a[8 + 2 * w], y1 = a[8 + 2 * w + 1];
        if (relu) { x0 = relu1(x0); x1 = relu1(x1); y0 = relu1(y0); y1 = relu1(y1); }
        f0.u[w] = pack_bf16(x0, x1);
        f1.u[w] = pack_bf16(y0, y1);
    }
#endif
}

NK_D void acc_to_frags8(const f32x16& a, bool relu, Frag& f0, Frag& f1, Frag& s8) {
    // four registers at a time: two bf16 pairs (dwords of f0 / f1) and one dword of e4m3 bytes
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        float x0 = a[4 * w], x1 = a[4 * w + 1], x2 = a[4 * w + 2], x3 = a[4 * w + 3];
        if (relu) { x0 = relu1(x0); x1 = relu1(x1); x2 = relu1(x2); x3 = relu1(x3); }
        Frag& f = w < 2 ? f0 : f1;
        f.u[2 * (w & 1)] = pack_bf16(x0, x1);
        f.u[2 * (w & 1) + 1] = pack_bf16(x2, x3);
        uint32_t d = pk_fp8(x2, x3, pk_fp8(x0, x1, 0, false), true);
        // computed here: left to itself hipcc holds all sixteen ReLU results back and converts them
        // late, which costs the forward kernel 45 spilled registers
        asm volatile("" : "+v"(d), "+v"(f.u[2 * (w & 1)]), "+v"(f.u[2 * (w & 1) + 1]));
        s8.u[w] = d;
    }
}
