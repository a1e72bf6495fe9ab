// Timing probe (dev tool): the MFMA chain's tile loop at equal FLOPs on the two bf16 MFMA shapes.
//
// One "stage" = a 32-row output tile of a 256-wide layer for the 32 points a wave owns: 16 A
// fragments of 1 KiB read from LDS (ds_read_b128, 8-deep read-ahead as in the chain kernels), B
// operands resident in registers (64 VGPRs), one barrier per stage.
//   form 32: 16 x v_mfma_f32_32x32x16_bf16 (one 32x32 accumulator tile)
//   form 16: 32 x v_mfma_f32_16x16x32_bf16 (two row tiles x two 16-point column tiles; every A
//            fragment feeds both column tiles, so LDS bytes per FLOP are the same)
// 4 waves per workgroup, 2 workgroups per CU, no global traffic inside the loop: what differs is the
// instruction mix and what the chip clocks under it.  Results are summed into a sink so nothing is
// optimised away; they are not meaningful numbers.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o tools/exp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int FORM>
__global__ __launch_bounds__(256, 2) void k_probe(const bf16x8* __restrict__ w, float* __restrict__ sink, int stages) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // 16 KiB of "weights" per ring slot, two slots
    for (int i = tid; i < 2 * 1024; i += 256) ((bf16x8*)lds)[i] = w[i];
    __syncthreads();
    bf16x8 b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) b[k] = w[(lane + 64 * k + blockIdx.x) & 2047];
    float total = 0.0f;
    for (int s = 0; s < stages; ++s) {
        const char* base = lds + (s & 1) * 16384 + lane * 16;
        bf16x8 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8*)(base + i * 1024);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        if (FORM == 32) {
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % 8], b[ks], acc, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (ks + 8 < 16) {
                    a[ks % 8] = *(const bf16x8*)(base + (ks + 8) * 1024);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) total += acc[i];
        } else {
            // row tile r (0, 1), k-step ks (0..7): fragment 8 r + ks; column tile c uses B operand b[2 ks + c]
            f32x4 acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                const int r = f >> 3, ks = f & 7;
                acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[f % 8], b[2 * ks], acc[r][0], 0, 0, 0);
                acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[f % 8], b[2 * ks + 1], acc[r][1], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (f + 8 < 16) {
                    a[f % 8] = *(const bf16x8*)(base + (f + 8) * 1024);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) total += acc[r][c][i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (total == 123.456f) sink[0] = total;
}

int main(int argc, char** argv) {
    const int stages = argc > 1 ? atoi(argv[1]) : 4000;
    bf16x8* w; float* sink;
    hipMalloc(&w, 2048 * 16); hipMalloc(&sink, 4);
    // random bf16 operands in (-1, 1): zeros toggle nothing and the chip then clocks far higher than under real data
    {
        unsigned short* h = (unsigned short*)malloc(2048 * 16);
        srand(1);
        for (int i = 0; i < 2048 * 8; ++i) {
            const float x = (float)rand() / RAND_MAX * 2.0f - 1.0f;
            unsigned u; memcpy(&u, &x, 4);
            h[i] = (unsigned short)(u >> 16);
        }
        hipMemcpy(w, h, 2048 * 16, hipMemcpyHostToDevice);
        free(h);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = 512 * 8;        // 16 rounds of 2 workgroups on each of 256 CUs
    for (int form : {32, 16, 32, 16}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (form == 32) hipLaunchKernelGGL(k_probe<32>, dim3(wgs), dim3(256), 32768, 0, w, sink, stages);
            else hipLaunchKernelGGL(k_probe<16>, dim3(wgs), dim3(256), 32768, 0, w, sink, stages);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)wgs * 4 * stages * 16 * 32768.0;
            if (rep) printf("form %2d: %8.3f ms  %7.1f TFLOP/s\n", form, ms, flop / ms / 1e9);
        }
    }
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
