// N3: reference-clip ingest on the device.  Restates, after WAV decoding, the arithmetic of
// AudioProcessor.load_audio (reference core/audio_processor.py:16-44): resample to the model rate, remove the DC
// offset, scale the peak to 29491 (90 % of full scale), truncate to int16.
//
//   resample : polyphase FIR  y[n] = sum_i x[i] * h[(n + skip) * down - i * up]   (f64 accumulate, f32 out)
//              h = the host-designed Kaiser low-pass (same design as the host mirror, core/audio_processor.py::_resample)
//   normalise: per clip  mean (f64 sum -> f32), peak = max |x - mean| (f32), out = int16((x - mean) * f32(29491 / peak))
// All HBM-bound streaming passes over a few hundred KB per clip; clips are batched on grid.y.
#include "vv_common.h"
#include "vv_kernels.h"

namespace {

__global__ __launch_bounds__(256) void resample_poly_kernel(const float* __restrict__ x, int n_in, const double* __restrict__ h,
                                                            int n_taps, int up, int down, int skip, float* __restrict__ y, int n_out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= n_out) return;
    const long long pos = (long long)(n + skip) * down;            // index into the zero-stuffed signal
    // taps k = pos - i*up in [0, n_taps): i in [ceil((pos - n_taps + 1)/up), floor(pos/up)]
    long long i_hi = pos / up;
    long long lo_num = pos - n_taps + 1;
    long long i_lo = lo_num <= 0 ? 0 : (lo_num + up - 1) / up;
    if (i_hi > n_in - 1) i_hi = n_in - 1;
    double acc = 0.0;
    for (long long i = i_lo; i <= i_hi; ++i) acc = fma((double)x[i], h[pos - i * up], acc);
    y[n] = (float)acc;
}

// stats[clip] = {sum (f64), peak bits (as f64 slot reused: low 32 bits hold the float bits of max |x - mean|)}
__global__ __launch_bounds__(256) void clip_sum_kernel(const float* __restrict__ x, const long long* __restrict__ off, double* __restrict__ stats) {
    const int clip = blockIdx.y;
    const long long a = off[clip], b = off[clip + 1];
    double s = 0.0;
    for (long long i = a + blockIdx.x * 256 + threadIdx.x; i < b; i += (long long)gridDim.x * 256) s += (double)x[i];
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(stats + 2 * clip, part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void clip_peak_kernel(const float* __restrict__ x, const long long* __restrict__ off, double* __restrict__ stats) {
    const int clip = blockIdx.y;
    const long long a = off[clip], b = off[clip + 1];
    if (b <= a) return;
    const float mean = (float)(stats[2 * clip] / (double)(b - a));
    float p = 0.f;
    for (long long i = a + blockIdx.x * 256 + threadIdx.x; i < b; i += (long long)gridDim.x * 256) p = fmaxf(p, fabsf(x[i] - mean));
    for (int o = 32; o; o >>= 1) p = fmaxf(p, __shfl_xor(p, o));
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = p;
    __syncthreads();
    if (threadIdx.x == 0) {
        p = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        atomicMax((unsigned*)(stats + 2 * clip + 1), __float_as_uint(p));     // non-negative floats order like their bit patterns
    }
}

__global__ __launch_bounds__(256) void clip_scale_kernel(const float* __restrict__ x, const long long* __restrict__ off,
                                                         const double* __restrict__ stats, int16_t* __restrict__ out) {
    const int clip = blockIdx.y;
    const long long a = off[clip], b = off[clip + 1];
    if (b <= a) return;
    const float mean = (float)(stats[2 * clip] / (double)(b - a));
    const float peak = __uint_as_float(*(const unsigned*)(stats + 2 * clip + 1));
    const float scale = peak > 0.f ? (float)(29491.0 / (double)peak) : 1.f;
    for (long long i = a + blockIdx.x * 256 + threadIdx.x; i < b; i += (long long)gridDim.x * 256)
        out[i] = (int16_t)(int)((x[i] - mean) * scale);                        // C truncation, like ndarray.astype(int16)
}

}  // namespace

int vvk_resample_poly(const float* x, int n_in, const double* h, int n_taps, int up, int down, int skip, float* y, int n_out,
                      hipStream_t st, const char** err) {
    if (n_in <= 0 || n_out <= 0 || n_taps <= 0 || up <= 0 || down <= 0 || skip < 0) { *err = "resample: bad shape"; return -22; }
    resample_poly_kernel<<<(n_out + 255) / 256, 256, 0, st>>>(x, n_in, h, n_taps, up, down, skip, y, n_out);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}

int vvk_normalize_clips(const float* x, const long long* off, int n_clips, long long max_len, double* stats, int16_t* out,
                        hipStream_t st, const char** err) {
    if (n_clips <= 0 || max_len <= 0) { *err = "normalize_clips: empty"; return -22; }
    if (hipMemsetAsync(stats, 0, sizeof(double) * 2 * n_clips, st) != hipSuccess) { *err = "normalize_clips: memset"; return -5; }
    long long bx = (max_len + 256 * 8 - 1) / (256 * 8);
    if (bx > 256) bx = 256;
    dim3 grid((unsigned)bx, n_clips);
    clip_sum_kernel<<<grid, 256, 0, st>>>(x, off, stats);
    clip_peak_kernel<<<grid, 256, 0, st>>>(x, off, stats);
    clip_scale_kernel<<<grid, 256, 0, st>>>(x, off, stats, out);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}
