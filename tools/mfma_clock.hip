// Bare fp32 MFMA loop on every SIMD of the chip: cycles per v_mfma_f32_32x32x2_f32 (s_memtime = shader clock) and the shader clock the chip
// holds under that load (against the 100 MHz wall clock), for 1..4 wavefronts per SIMD, with and without ds_read_b128 operand traffic.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_clock.hip -o gpurun_out/mfma_clock ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool LDS>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float* out, unsigned long long* cyc, unsigned long long* wall) {
    __shared__ __attribute__((aligned(16))) float sh[2][64][36];
    for (int e = threadIdx.x; e < 2 * 64 * 36; e += 256) (&sh[0][0][0])[e] = 1.0f + (e & 7) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc = {0};
    float a = 1.0f + lane * 0.001f, b = 0.5f;
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if constexpr (LDS) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(&sh[0][(wave & 1) * 32 + (lane & 31)][ks * 8 + (lane >> 5) * 4]);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(&sh[1][(wave >> 1) * 32 + (lane & 31)][ks * 8 + (lane >> 5) * 4]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
}

template <bool LDS>
void run(int wgs_per_cu, int iters) {
    const int blocks = 256 * wgs_per_cu;
    float* out; unsigned long long *cyc, *wall;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8); hipMalloc(&wall, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<LDS><<<blocks, 256>>>(iters / 8, out, cyc, wall);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<LDS><<<blocks, 256>>>(iters, out, cyc, wall);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(blocks), hw(blocks);
    hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(hw.data(), wall, blocks * 8, hipMemcpyDeviceToHost);
    double c = 0, w = 0; for (int i = 0; i < blocks; ++i) { c += hc[i]; w += hw[i]; }
    c /= blocks; w /= blocks;
    const double flop = (double)blocks * 4 * iters * 16 * 4096.0;
    printf("%s waves/SIMD %d: %.3f ms  %.1f TF/s   cycles per MFMA per SIMD %.1f   shader clock %.3f GHz (s_memtime %.0f ticks in %.0f x 10 ns)\n",
           LDS ? "lds+mfma" : "mfma    ", wgs_per_cu, ms, flop / ms / 1e9, c / (16.0 * iters * wgs_per_cu), c / (w * 10.0), c, w);
    hipFree(out); hipFree(cyc); hipFree(wall);
}

int main() {
    for (int w = 1; w <= 4; ++w) run<false>(w, 4000);
    for (int w = 1; w <= 4; ++w) run<true>(w, 4000);
    for (int w = 1; w <= 4; ++w) run<false>(w, 40000);
    return 0;
}
