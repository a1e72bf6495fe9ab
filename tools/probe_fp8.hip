// Probe of the gfx950 8-bit building blocks the fp8 activation experiment relies on (dev tool):
//   1. which LDS bytes ds_read_b64_tr_b8 delivers to which lane,
//   2. rounding / overflow behaviour of v_cvt_pk_fp8_f32 and v_cvt_pk_bf8_f32 (OCP e4m3 / e5m2),
//   3. the k order of the operands of v_mfma_f32_32x32x16_fp8_fp8 / _bf8_fp8.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_fp8.hip -o tools/exp/probe_fp8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k_tr(uint16_t* out) {       // out[lane][8] = LDS offset each received byte came from
    __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
    const int lane = threadIdx.x;
    uint16_t res[8];
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = lane; i < 4096; i += 64) lds[i] = pass ? (i >> 8) : (i & 255);
        __syncthreads();
        const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + lane * 32;
        u32x2 v;
        asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        for (int j = 0; j < 8; ++j) {
            const uint32_t b = (v[j >> 2] >> (8 * (j & 3))) & 255;
            res[j] = pass ? (uint16_t)(res[j] | (b << 8)) : (uint16_t)b;
        }
        __syncthreads();
    }
    for (int j = 0; j < 8; ++j) out[lane * 8 + j] = res[j];
}

__global__ void k_cvt(const float* in, int n, uint32_t* fp8, uint32_t* bf8, float* back8, float* backb, int ovfl) {
    if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);      // MODE bit 23 (FP16_OVFL), width 1
    const int i = threadIdx.x;
    if (i >= n) return;
    const int p8 = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], in[i], 0, false);
    const int pb = __builtin_amdgcn_cvt_pk_bf8_f32(in[i], in[i], 0, false);
    fp8[i] = p8 & 255; bf8[i] = pb & 255;
    back8[i] = __builtin_amdgcn_cvt_f32_fp8(p8, 0);
    backb[i] = __builtin_amdgcn_cvt_f32_bf8(pb, 0);
}

// C[m][n] = sum_k A[m][k] B[k][n], operands handed over as (lane r = m or n, half kh): bytes j = k 8kh + j
__global__ void k_mfma(const unsigned char* A /*[32][16]*/, const unsigned char* B /*[16][32]*/, float* C, int a_bf8) {
    const int lane = threadIdx.x, r = lane & 31, kh = lane >> 5;
    uint64_t a = 0, b = 0;
    for (int j = 0; j < 8; ++j) {
        a |= (uint64_t)A[r * 16 + 8 * kh + j] << (8 * j);
        b |= (uint64_t)B[(8 * kh + j) * 32 + r] << (8 * j);
    }
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (a_bf8) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8((long)a, (long)b, acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8((long)a, (long)b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * kh) * 32 + r] = acc[i];
}

static float dec(unsigned v, int mant, int bias) {      // OCP e4m3fn / e5m2 byte -> float
    const int s = v >> 7, e = (v >> mant) & ((1 << (7 - mant)) - 1), m = v & ((1 << mant) - 1);
    float x = e ? ldexpf(1.0f + m / (float)(1 << mant), e - bias) : ldexpf(m / (float)(1 << mant), 1 - bias);
    return s ? -x : x;
}

int main() {
    uint16_t* d_out; hipMalloc(&d_out, 64 * 8 * 2);
    k_tr<<<1, 64>>>(d_out);
    uint16_t h[512]; hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b8: lane -> (source lane : byte) of its 8 bytes, addresses lane*32\n");
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 8; ++j) printf(" %2d:%d", h[l * 8 + j] / 32, h[l * 8 + j] % 32);
        printf("\n");
    }
    const float vals[] = {0.0f, 0.3f, 1.0f, 1.0625f, 1.1f, 0.0019f, 0.001f, 447.0f, 448.0f, 464.0f, 500.0f, 1e6f, -1.0f, -500.0f,
                          57344.0f, 60000.0f, 70000.0f, 1e-5f, 1.5e-5f, 7.6e-6f, 6e-5f, 3.0e-8f};
    const int n = sizeof(vals) / sizeof(vals[0]);
    float *d_in, *d_b8, *d_bb; uint32_t *d_f8, *d_bf;
    hipMalloc(&d_in, n * 4); hipMalloc(&d_b8, n * 4); hipMalloc(&d_bb, n * 4); hipMalloc(&d_f8, n * 4); hipMalloc(&d_bf, n * 4);
    hipMemcpy(d_in, vals, n * 4, hipMemcpyHostToDevice);
    for (int ov = 0; ov < 2; ++ov) {
        k_cvt<<<1, 64>>>(d_in, n, d_f8, d_bf, d_b8, d_bb, ov);
        uint32_t f8[64], bf[64]; float b8[64], bb[64];
        hipMemcpy(f8, d_f8, n * 4, hipMemcpyDeviceToHost); hipMemcpy(bf, d_bf, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(b8, d_b8, n * 4, hipMemcpyDeviceToHost); hipMemcpy(bb, d_bb, n * 4, hipMemcpyDeviceToHost);
        printf("conversions, FP16_OVFL=%d\n", ov);
        for (int i = 0; i < n; ++i)
            printf("  %12g -> e4m3 0x%02x = %g (hw back %g)   e5m2 0x%02x = %g (hw back %g)\n", vals[i], f8[i], dec(f8[i], 3, 7),
                   b8[i], bf[i], dec(bf[i], 2, 15), bb[i]);
    }
    // MFMA operand order
    unsigned char A[512], B[512]; float Af[512], Bf[512], C[1024], Cref[1024];
    for (int a_bf8 = 0; a_bf8 < 2; ++a_bf8) {
        srand(1);
        for (int i = 0; i < 512; ++i) {
            A[i] = (rand() % 0x30) + 0x20; B[i] = (rand() % 0x30) + 0x20;
            if (rand() & 1) A[i] |= 0x80;
            Af[i] = a_bf8 ? dec(A[i], 2, 15) : dec(A[i], 3, 7); Bf[i] = dec(B[i], 3, 7);
        }
        for (int m = 0; m < 32; ++m) for (int nn = 0; nn < 32; ++nn) {
            double s = 0; for (int k = 0; k < 16; ++k) s += (double)Af[m * 16 + k] * Bf[k * 32 + nn];
            Cref[m * 32 + nn] = (float)s;
        }
        unsigned char *dA, *dB; float* dC;
        hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 4096);
        hipMemcpy(dA, A, 512, hipMemcpyHostToDevice); hipMemcpy(dB, B, 512, hipMemcpyHostToDevice);
        k_mfma<<<1, 64>>>(dA, dB, dC, a_bf8);
        hipMemcpy(C, dC, 4096, hipMemcpyDeviceToHost);
        double err = 0, mag = 0;
        for (int i = 0; i < 1024; ++i) { err = fmax(err, fabs(C[i] - Cref[i])); mag = fmax(mag, fabs(Cref[i])); }
        printf("mfma 32x32x16 %s_fp8: max |C - ref| = %g (max |ref| %g)\n", a_bf8 ? "bf8" : "fp8", err, mag);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("done: %s\n", hipGetErrorString(e));
    return 0;
}
