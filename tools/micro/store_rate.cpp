// Microbenchmark: global store rate per CU for 16-B-per-lane stores (1 KB per wave instruction), by number of CUs storing.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_rate tools/micro/store_rate.cpp && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ROWS>   // ROWS = rows of 256 B per wave instruction (1: 1 KB contiguous per wave; 4: in_proj epilogue shape)
__global__ __launch_bounds__(512) void store_kernel(uint4* out, int iters, size_t row_stride16, unsigned long long* cyc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint4 v = make_uint4(tid, blockIdx.x, 0, 0);
    // wave w stores rows [w*32, w*32+32) of a [256 rows][row_stride] tile in 8 instructions of 4 rows x 256 B (ROWS = 4),
    // or 1 KB contiguous per instruction (ROWS = 1)
    for (int it = 0; it < iters; ++it) {
        uint4* base = out + ((size_t)blockIdx.x * iters + it) * (ROWS == 4 ? 16 : 4096);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v.z = it + i;
            if (ROWS == 4) base[(size_t)(wave * 32 + i * 4 + (lane >> 4)) * row_stride16 + (lane & 15)] = v;
            else base[(wave * 8 + i) * 64 + lane] = v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int iters = 64;
    const size_t row_stride16 = 8256 * 2 / 16;   // z rows: Lp = 8256 fp16
    size_t bytes = (size_t)2048 * iters * 65536 + (size_t)256 * row_stride16 * 16 + (1 << 20);
    uint4* out;
    unsigned long long* cyc;
    hipMalloc(&out, bytes);
    hipMalloc(&cyc, 4096 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rows : {1, 4})
        for (int grid : {1, 8, 32, 64, 128, 256, 512}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (rows == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(512), 0, 0, out, iters, row_stride16, cyc);
                else hipLaunchKernelGGL(store_kernel<4>, dim3(grid), dim3(512), 0, 0, out, iters, row_stride16, cyc);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(grid);
            hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
            double mean = 0;
            for (auto c : h) mean += c;
            mean /= grid;
            const double per_wg = (double)iters * 64 * 1024;   // bytes per workgroup
            std::printf("rows/instr %d  wgs %4d  %.3f ms  %.1f GB/s total  mean ticks/wg %.0f  -> %.1f B/tick/CU, %.0f ticks per store instr (64 per iter per CU)\n",
                        rows, grid, ms, per_wg * grid / ms / 1e6, mean, per_wg / mean, mean / (iters * 64));
        }
    return 0;
}
