// Micro-benchmark (tools, not product): HBM copy rate of a [rows][T] f32 slab as a function of the contiguous run one wave-instruction
// touches per row.  Pattern "dword": a wave instruction reads 2 rows x 128 B (lanes 0-31 one row, 32-63 another) -- the x2 up-sampler's
// fragment loads; "dwordx4 x 4 rows": 4 rows x 256 B; "dwordx4 x 1 row": 1 KB of one row (the conv_post stream kernel's form).
// Each wave copies 16 rows x 64 columns per step (the up-sampler's chunk), writes the same shape elsewhere.
//   hipcc --offload-arch=gfx950 -O3 tools/run_granularity.hip -o /tmp/run_gran && /tmp/run_gran
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(512) void copy_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int T, int n_units, int n_cb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int u = blockIdx.x * 8 + wave; u < n_units; u += gridDim.x * 8) {
        const int rb = u / n_cb, cb = u - rb * n_cb;            // 16-row block, column block
        const size_t base = (size_t)rb * 16 * T;
        if (MODE == 0) {                                        // 64 columns: dword per lane, 2 rows x 128 B per instruction, 2 column halves
            const int col = cb * 64 + (lane & 31), half = lane >> 5;
            float v[2][8];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[ti][j] = in[base + (size_t)(half * 8 + j) * T + col + ti * 32];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int j = 0; j < 8; ++j) out[base + (size_t)(half * 8 + j) * T + col + ti * 32] = v[ti][j];
        } else if (MODE == 1) {                                 // 64 columns: dwordx4 per lane, 4 rows x 256 B per instruction
            const int col = cb * 64 + (lane & 15) * 4, r = lane >> 4;
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *(const float4*)(in + base + (size_t)(k * 4 + r) * T + col);
#pragma unroll
            for (int k = 0; k < 4; ++k) *(float4*)(out + base + (size_t)(k * 4 + r) * T + col) = v[k];
        } else {                                                // 256 columns x 4 rows per unit-quarter: dwordx4 per lane, 1 row x 1 KB per instruction
            const int q = cb & 3, cbb = cb >> 2;                // the same bytes per unit: 4 rows x 256 columns
            const int col = cbb * 256 + lane * 4;
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *(const float4*)(in + base + (size_t)(q * 4 + k) * T + col);
#pragma unroll
            for (int k = 0; k < 4; ++k) *(float4*)(out + base + (size_t)(q * 4 + k) * T + col) = v[k];
        }
    }
}

int main() {
    const int rows = 2048, T = 132608;                          // 32 items x 64 channels, T a multiple of 256: 1.087 GB
    const size_t n = (size_t)rows * T;
    float *in, *out;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4);
    hipMemset(in, 1, n * 4); hipMemset(out, 0, n * 4);
    const int n_cb = T / 64, n_units = rows / 16 * n_cb;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"dword, 2 rows x 128 B per instruction", "dwordx4, 4 rows x 256 B per instruction", "dwordx4, 1 row x 1 KB per instruction"};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 3; ++m) {
            hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) {
                if (m == 0) copy_kernel<0><<<256, 512>>>(in, out, rows, T, n_units, n_cb);
                if (m == 1) copy_kernel<1><<<256, 512>>>(in, out, rows, T, n_units, n_cb);
                if (m == 2) copy_kernel<2><<<256, 512>>>(in, out, rows, T, n_units, n_cb);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("%-44s %.3f ms  %.2f TB/s (read + write)\n", names[m], ms, 2.0 * n * 4 / ms / 1e9);
        }
    return 0;
}
