// What the chip sustains on v_mfma_f32_32x32x16_bf16 in the shape the split fp32 contraction uses it: six MFMAs per k-step on three accumulators,
// operands (a) held in registers, (b) re-read from LDS (one ds_read_b128 per MFMA, software-pipelined one step ahead), at 1 / 2 / 4 wavefronts
// per SIMD on every CU.  Prints TFLOP/s (bf16 MFMA FLOPs) and the cycles per MFMA per SIMD at 2.4 GHz.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_bf16_rate.hip -o tools/bin/mfma_bf16_rate && tools/bin/mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS = 2: as 1, with a workgroup barrier every three iterations (36 MFMAs: the stage of conv3x3_patch_kernel) and the first fragments of
// the next stage read only after it -- what a stage boundary costs a wavefront that has no partner on its SIMD.
template <int LDS>
__global__ __launch_bounds__(256) void loop(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short panel[6][64][40];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 6 * 64 * 40; i += blockDim.x) (&panel[0][0][0])[i] = (unsigned short)(0x3c00 + (i & 127));
    __syncthreads();
    f32x16 acc = {0}, mid = {0}, low = {0};
    const unsigned short* base = &panel[0][lane & 31][(lane >> 5) * 8];
    bf16x8 f[2][6];
#pragma unroll
    for (int q = 0; q < 6; ++q) f[0][q] = *reinterpret_cast<const bf16x8*>(base + q * 64 * 40);
#pragma unroll
    for (int q = 0; q < 6; ++q) f[1][q] = f[0][q];
    for (int it = 0; it < iters; ++it) {
        if (LDS == 2 && it % 3 == 0) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 6; ++q) f[0][q] = *reinterpret_cast<const bf16x8*>(base + q * 64 * 40 + (it & 1) * 16);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (LDS) {
#pragma unroll
                for (int q = 0; q < 6; ++q) f[s ^ 1][q] = *reinterpret_cast<const bf16x8*>(base + q * 64 * 40 + ((it + s) & 1) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8* a = &f[s][0];
            const bf16x8* b = &f[s][3];
            low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], low, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], mid, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
            low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], low, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], mid, 0, 0, 0);
            low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], low, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    acc += mid + low;
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}

template <int LDS>
static void run(int waves_per_simd, int iters) {
    float* out;
    hipMalloc(&out, 4);
    const int blocks = 256 * waves_per_simd;                 // 256 threads = one wavefront per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop<LDS>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = (double)blocks * 4 * iters * 12;
        const double tf = mfmas * 32768.0 / (ms * 1e-3) / 1e12;
        if (rep == 2)
            printf("%-28s %d wavefront(s) per SIMD: %8.3f ms  %8.1f TFLOP/s  %6.1f cycles per MFMA per SIMD at 2.4 GHz\n", LDS == 2 ? "LDS + barrier per 36 MFMAs" : LDS ? "operands re-read from LDS" : "operands in registers",
                   waves_per_simd, ms, tf, ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * 12));
    }
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) run<0>(w, 20000 / w);
    for (int w : {1, 2, 4}) run<1>(w, 20000 / w);
    for (int w : {1, 2, 4}) run<2>(w, 19998 / w / 3 * 3);
    return 0;
}
