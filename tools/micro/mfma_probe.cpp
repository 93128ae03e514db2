// Microbenchmark: back-to-back v_mfma_f32_32x32x16_f16 on every CU with 1, 4, 8, 16 waves per CU: cycles per MFMA, the shader
// clock the chip sustains meanwhile (s_memtime against the 100-MHz s_memrealtime) and the resulting dense TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/micro/mfma_probe.cpp && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NSETS>
__global__ void probe(unsigned long long* out, int iters, float* sink, const _Float16* src) {
    f32x16 acc[4] = {};
    f16x8 a[NSETS], b[NSETS];
    for (int s = 0; s < NSETS; ++s)
        for (int i = 0; i < 8; ++i) { a[s][i] = src[(threadIdx.x * 8 + i + s * 17) & 1023]; b[s][i] = src[(threadIdx.x * 8 + i + s * 29 + 5) & 1023]; }
    __syncthreads();
    const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k % NSETS], b[(k / 4) % NSETS], acc[k & 3], 0, 0, 0);
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 123.456f) sink[0] = s;
    // the SIMD serves its oldest wave first: time the LAST wave of the workgroup, not wave 0
    if ((threadIdx.x & 63) == 0) { atomicMax(&out[2 * blockIdx.x], m1 - m0); atomicMax(&out[2 * blockIdx.x + 1], r1 - r0); }
}
int main() {
    unsigned long long* out; float* sink; _Float16* src;
    (void)hipMalloc(&out, 4096 * 16); (void)hipMalloc(&sink, 4); (void)hipMalloc(&src, 2048);
    std::vector<_Float16> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (_Float16)((i % 37) * 0.03f - 0.5f);
    (void)hipMemcpy(src, h.data(), 2048, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int nsets : {1, 4})
        for (int threads : {64, 256, 512, 1024}) {
            for (int grid : {256}) {
                (void)hipMemset(out, 0, 4096 * 16);
                if (nsets == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(threads), 0, 0, out, iters, sink, src);
                else hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(threads), 0, 0, out, iters, sink, src);
                (void)hipDeviceSynchronize();
                std::vector<unsigned long long> o(2 * grid);
                (void)hipMemcpy(o.data(), out, 16 * grid, hipMemcpyDeviceToHost);
                double m = 0, r = 0; for (int i = 0; i < grid; ++i) m += o[2 * i], r += o[2 * i + 1];
                m /= grid; r /= grid;
                const double waves = threads / 64.0, mf = (double)iters * 16;
                std::printf("operand sets %d  %4d threads/WG x %d WGs: %.2f ms, %.0f MHz, cycles per MFMA per wave %.1f, per CU-wide MFMA slot (cycles*4/(waves*mfma)) %.1f, %.0f TFLOP/s\n",
                            nsets, threads, grid, r / 1e5, m / r * 100, m / mf, m * 4 / (waves * mf), grid * waves * mf * 32768 / (r / 1e8) / 1e12);
            }
        }
    return 0;
}
