// HBM bandwidth ceilings for the access patterns the MLP kernels use (dev tool, not shipped):
//   w8   : 8-byte stores, a wave writes 4 x 512 B contiguous pieces 512 B apart (activation tile store)
//   w16  : 16-byte stores, a wave writes 2 x 1 KiB contiguous pieces
//   r16  : global_load_lds 16 B per lane, 1 KiB per wave instruction (GEMM / weight stream)
//   rv16 : plain 16-byte vector loads
// build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o gpurun_out/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// each wave handles `per_wave` bytes in 2 KiB tiles
__global__ __launch_bounds__(256) void k_w8(char* dst, int64_t tiles_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    char* base = dst + w * tiles_per_wave * 2048 + (lane & 31) * 16 + 8 * (lane >> 5);
    const u32x2 v = {(uint32_t)lane, (uint32_t)w};
    for (int64_t t = 0; t < tiles_per_wave; ++t) {
        char* p = base + t * 2048;
        *(u32x2*)(p + 0) = v; *(u32x2*)(p + 512) = v; *(u32x2*)(p + 1024) = v; *(u32x2*)(p + 1536) = v;
    }
}
__global__ __launch_bounds__(256) void k_w16(char* dst, int64_t tiles_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    char* base = dst + w * tiles_per_wave * 2048 + lane * 16;
    const u32x4 v = {(uint32_t)lane, (uint32_t)w, 1u, 2u};
    for (int64_t t = 0; t < tiles_per_wave; ++t) {
        char* p = base + t * 2048;
        *(u32x4*)(p + 0) = v; *(u32x4*)(p + 1024) = v;
    }
}
// strided like the real kernel: a wave's tile t of region r lives at r*region_stride + w*tile_bytes*NT + t*2048
__global__ __launch_bounds__(256) void k_w8_regions(char* dst, int regions, int64_t region_stride, int tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    const u32x2 v = {(uint32_t)lane, (uint32_t)w};
    for (int r = 0; r < regions; ++r) {
        char* base = dst + r * region_stride + w * tiles * 2048 + (lane & 31) * 16 + 8 * (lane >> 5);
        for (int t = 0; t < tiles; ++t) {
            char* p = base + t * 2048;
            *(u32x2*)(p + 0) = v; *(u32x2*)(p + 512) = v; *(u32x2*)(p + 1024) = v; *(u32x2*)(p + 1536) = v;
        }
    }
}
__global__ __launch_bounds__(512) void k_r16(const char* src, int64_t kib_per_wave, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * 8 + wave;
    const char* base = src + w * kib_per_wave * 1024 + lane * 16;
    char* l = lds + wave * 8192;
    for (int64_t t = 0; t < kib_per_wave; t += 8) {
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (t + j) * 1024),
                                             (__attribute__((address_space(3))) void*)(l + j * 1024), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0);
    }
    if (sink && lane == 0 && w == 0) sink[0] = *(uint32_t*)l;
}
__global__ __launch_bounds__(256) void k_rv16(const char* src, int64_t kib_per_wave, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    const char* base = src + w * kib_per_wave * 1024 + lane * 16;
    u32x4 acc = {0, 0, 0, 0};
    for (int64_t t = 0; t < kib_per_wave; t += 4) {
        u32x4 a = *(const u32x4*)(base + t * 1024), b = *(const u32x4*)(base + (t + 1) * 1024);
        u32x4 c = *(const u32x4*)(base + (t + 2) * 1024), d = *(const u32x4*)(base + (t + 3) * 1024);
        acc ^= a ^ b ^ c ^ d;
    }
    if (sink && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[1] = 1;
}

// bare MFMA loop: 4 independent 32x32x16 bf16 chains per wave, operands in registers
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 1.2345f) out[0] = 1.0f;
}

template <typename F>
static void run(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.3f ms  %7.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
}

int main() {
    const int64_t bytes = 16LL << 30;
    char* buf; uint32_t* sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, bytes));
    for (int wgs : {2048, 8192, 32768}) {
        const int64_t waves = (int64_t)wgs * 4, tiles = bytes / 2048 / waves;
        char nm[64];
        snprintf(nm, 64, "w8   %d wgs", wgs);
        run(nm, (double)waves * tiles * 2048, [&] { hipLaunchKernelGGL(k_w8, dim3(wgs), dim3(256), 0, 0, buf, tiles); });
        snprintf(nm, 64, "w16  %d wgs", wgs);
        run(nm, (double)waves * tiles * 2048, [&] { hipLaunchKernelGGL(k_w16, dim3(wgs), dim3(256), 0, 0, buf, tiles); });
        snprintf(nm, 64, "rv16 %d wgs", wgs);
        run(nm, (double)waves * tiles * 2048, [&] { hipLaunchKernelGGL(k_rv16, dim3(wgs), dim3(256), 0, 0, buf, tiles * 2, sink); });
    }
    {   // the activation-save pattern: 24576 WGs x 4 waves, 11 regions of 8 tiles each
        const int wgs = 24576, regions = 10, tiles = 8;
        const int64_t stride = (int64_t)wgs * 4 * tiles * 2048;
        run("w8 regions (fwd save shape)", (double)regions * stride,
            [&] { hipLaunchKernelGGL(k_w8_regions, dim3(wgs), dim3(256), 0, 0, buf, regions, stride, tiles); });
    }
    for (int wgs : {256, 512, 1024}) {
        const int64_t waves = (int64_t)wgs * 8, kib = bytes / 1024 / waves / 8 * 8;
        char nm[64];
        snprintf(nm, 64, "r16 lds %d wgs", wgs);
        run(nm, (double)waves * kib * 1024, [&] { hipLaunchKernelGGL(k_r16, dim3(wgs), dim3(512), 65536, 0, buf, kib, sink); });
    }
    for (int wpc : {4, 8}) {     // waves per CU: 1 or 2 per SIMD
        const int iters = 20000, wgs = 256 * wpc / 4;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(256), 0, 0, (float*)sink, 100); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(256), 0, 0, (float*)sink, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = (double)wgs * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
        printf("mfma bf16 32x32x16, %d waves/CU  %8.3f ms  %7.1f TFLOP/s\n", wpc, ms, flop / (ms * 1e-3) / 1e12);
    }
    return 0;
}
