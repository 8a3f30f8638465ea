// What does a random gather of 64 bytes cost on gfx950: a 64-byte sector or a 128-byte line?  (The localise kernel
// gathers two flank windows of <= 64 bytes per read out of 128-byte lines; MI355X_MICROARCH.md calls other access widths
// than wide streaming reads uncalibrated.)  Three kernels over a 3 GB table, one lane per random unit:
//   unit64    4 x 16 B at a 64-byte aligned random offset
//   in_line   4 x 16 B at a 16-byte aligned random offset inside a random 128-byte line (today's layout)
//   line128   8 x 16 B, a whole random 128-byte line
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_granularity gather_granularity.hip ; run under rocprofv3 --pmc FETCH_SIZE
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}

template <int MODE>
__global__ void __launch_bounds__(256) gather(const uint4* __restrict__ table, uint64_t n_lines, uint32_t* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t h = mix(i * 0x9e3779b97f4a7c15ull + 12345);
    const uint64_t line = h % n_lines;
    uint64_t first;                      // in uint4 units
    int cnt;
    if (MODE == 0) { first = line * 8 + ((h >> 40) & 1) * 4; cnt = 4; }
    else if (MODE == 1) { first = line * 8 + ((h >> 40) & 3) + ((h >> 42) & 1); cnt = 4; }      // offsets 0..4: stays in the line
    else { first = line * 8; cnt = 8; }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (k < cnt) { const uint4 v = table[first + k]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    out[i] = acc;
}

int main() {
    const uint64_t bytes = 3ull << 30, n_lines = bytes / 128, n = 20000000;
    uint4* table; uint32_t* out;
    if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&out, n * 4) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
    hipMemset(table, 1, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* names[3] = {"unit64 ", "in_line", "line128"};
    for (int rep = 0; rep < 3; rep++)
        for (int m = 0; m < 3; m++) {
            hipEventRecord(a);
            const dim3 g((unsigned)((n + 255) / 256)), t(256);
            if (m == 0) hipLaunchKernelGGL(gather<0>, g, t, 0, 0, table, n_lines, out, n);
            else if (m == 1) hipLaunchKernelGGL(gather<1>, g, t, 0, 0, table, n_lines, out, n);
            else hipLaunchKernelGGL(gather<2>, g, t, 0, 0, table, n_lines, out, n);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            if (hipGetLastError() != hipSuccess) { fprintf(stderr, "launch failed\n"); return 1; }
            printf("rep %d %s: %.3f ms, %.2f G gathers/s, %.0f GB/s of bytes used\n", rep, names[m], ms, n / ms / 1e6, n * (m == 2 ? 128.0 : 64.0) / ms / 1e6);
        }
    hipFree(table); hipFree(out);
    return 0;
}
