// K1: reference-audio front end: int16 PCM -> centered, reflect-padded frames -> Hann window ->
// 1024-point DFT magnitude -> mel filterbank -> log(clamp 1e-5).   One workgroup per frame.
// The frame, the twiddle table and the magnitudes live in LDS; audio is read once (coalesced
// int16), the 513x100 filterbank comes from L2.  Direct DFT (exact twiddle table, fp32
// accumulation): 1 MFLOP-class per frame and run once per utterance, so clarity wins over an FFT.
#include "vv_common.h"
#include "vv_kernels.h"

namespace {

__global__ __launch_bounds__(256) void mel_kernel(const int16_t* __restrict__ audio, int ld_audio,
                                                  const int* __restrict__ audio_len, const float* __restrict__ window,
                                                  const float* __restrict__ tw_cos, const float* __restrict__ tw_sin,
                                                  const float* __restrict__ fb /*[n_fft/2+1][n_mel]*/, float* __restrict__ mel,
                                                  int F_max, int n_fft, int hop, int n_mel) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* fr = (float*)smem;            // [n_fft]
    float* tc = fr + n_fft;              // [n_fft]
    float* ts = tc + n_fft;              // [n_fft]
    float* mag = ts + n_fft;             // [n_fft/2 + 1]
    const int b = blockIdx.y, f = blockIdx.x;
    const int S = max(audio_len[b], 1);       // an empty clip reads the row's first (padding) sample, never a[-1]
    const int frames = audio_len[b] / hop + 1;
    const int nb = n_fft / 2 + 1;
    if (f >= frames) {                   // frames beyond this clip: defined zeros (never read as reference)
        for (int m = threadIdx.x; m < n_mel; m += 256) mel[((size_t)b * F_max + f) * n_mel + m] = 0.f;
        return;
    }
    const int16_t* a = audio + (size_t)b * ld_audio;
    const int half = n_fft / 2;
    for (int n = threadIdx.x; n < n_fft; n += 256) {
        int pos = f * hop + n - half;
        if (pos < 0) pos = -pos;                       // reflect (no edge repeat), torch.stft center=True
        if (pos >= S) pos = 2 * (S - 1) - pos;
        pos = min(max(pos, 0), S - 1);
        fr[n] = (float)a[pos] * (1.0f / 32768.0f) * window[n];
        tc[n] = tw_cos[n];
        ts[n] = tw_sin[n];
    }
    __syncthreads();
    const int mask = n_fft - 1;
    for (int k = threadIdx.x; k < nb; k += 256) {
        float re = 0.f, im = 0.f;
        int idx = 0;
        for (int n = 0; n < n_fft; ++n) {
            const float x = fr[n];
            re = fmaf(x, tc[idx], re);
            im = fmaf(x, ts[idx], im);
            idx = (idx + k) & mask;
        }
        mag[k] = sqrtf(re * re + im * im);
    }
    __syncthreads();
    for (int m = threadIdx.x; m < n_mel; m += 256) {
        float acc = 0.f;
        for (int k = 0; k < nb; ++k) acc = fmaf(fb[(size_t)k * n_mel + m], mag[k], acc);
        mel[((size_t)b * F_max + f) * n_mel + m] = logf(fmaxf(acc, 1e-5f));
    }
}

}  // namespace

int vvk_mel(const int16_t* audio, int ld_audio, const int* audio_len, const float* window, const float* tw_cos, const float* tw_sin,
            const float* fb, float* mel, int B, int F_max, int n_fft, int hop, int n_mel, hipStream_t st, const char** err) {
    if (B <= 0 || F_max <= 0) { *err = "mel: empty"; return -22; }
    if (n_fft & (n_fft - 1) || n_fft > 4096) { *err = "mel: n_fft must be a power of two <= 4096"; return -22; }
    const size_t lds = (size_t)(3 * n_fft + n_fft / 2 + 1) * sizeof(float);
    dim3 grid(F_max, B);
    mel_kernel<<<grid, 256, lds, st>>>(audio, ld_audio, audio_len, window, tw_cos, tw_sin, fb, mel, F_max, n_fft, hop, n_mel);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}
