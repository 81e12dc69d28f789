// Calibration: sustained bf16 MFMA rate of register-resident loops (no LDS, no global traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE> __global__ __launch_bounds__(512, 2) void k(const float* seed, float* out, int iters) {
    const int lane = threadIdx.x;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)seed[(lane * 37 + i * 8 + j) & 1023]; b[i][j] = (__bf16)seed[(lane * 91 + i * 8 + j + 5) & 1023]; }
    if constexpr (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        float s = 0; for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 1], acc[i & 3], 0, 0, 0);
        float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}
int main(int argc, char** argv) {
    float *seed, *out; hipMalloc(&seed, 4096); hipMalloc(&out, 4 * 1024 * 1024);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = argc > 1 ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape : {16, 32}) for (int threads : {256, 512}) {
        const int iters = 20000, blocks = 256;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (shape == 16) k<16><<<blocks, threads>>>(seed, out, iters); else k<32><<<blocks, threads>>>(seed, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * (threads / 64) * iters * (shape == 16 ? 16 * 16384.0 : 8 * 32768.0);
        printf("shape %dx%d waves/CU %d %s: %.1f TFLOP/s (%.2f ms)\n", shape, shape, threads / 64, argc > 1 ? "zeros" : "random", flops / ms / 1e9, ms);
    }
    return 0;
}
