// N3 + a8: reference-clip ingest on the device.  Restates, after RIFF parsing on the host (no arithmetic there),
// the arithmetic of AudioProcessor.load_audio (reference core/audio_processor.py:15-44):
//   AudioSegment.set_channels(1)      -> audioop.tomono(data, width, 0.5, 0.5)  = (l + r) >> 1      (pydub audio_segment.py)
//   AudioSegment.set_frame_rate(sr)   -> audioop.ratecv(data, width, 1, src, sr, None): linear interpolation in float64 on the
//                                        samples shifted to 32 bit, truncated, shifted back (CPython Modules/audioop.c)
//   np.array(samples, float32) ; x - np.mean(x) ; peak = max |.| ; * f32(29491 / peak) ; astype(int16)        (:25,28-44)
// BIT-EXACT against stdlib audioop + numpy (tests/test_ingest_gpu.py, tests/golden/ingest_golden.npz), including numpy's
// float32 summation order in np.mean: add.reduce walks the array in buffers of 8192 elements, sums each buffer pairwise
// (leaves of <= 128 elements on 8 interleaved accumulators, split at (n / 2) & ~7) and accumulates the buffer sums in order.
//
//   ingest_pcm : one thread per OUTPUT sample: two input frames -> mono -> interpolate -> float32             (HBM stream)
//   chunk sums : one workgroup per 8192-sample buffer; 8 lanes per leaf = numpy's 8 accumulators; tree combined through LDS
//   mean / peak / scale : streaming passes, clips batched on grid.y
// The polyphase FIR resampler of rounds 2-4 (vv_resample_poly) stays as an explicit opt-in; it is NOT the reference's arithmetic.
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

// ---------------------------------------------------------------- audioop.tomono + audioop.ratecv, one thread per output sample
// desc[clip] = {byte offset of the clip's interleaved PCM, width (1 | 2 | 4 bytes, signed), channels, n_frames, I, O, out offset, n_out}
// (I, O) = (src, dst) rates / gcd; I == O means "same rate": pydub skips ratecv.
template <typename T>
__device__ __forceinline__ long long mono_frame(const T* __restrict__ p, long long f, int ch) {
    const T* q = p + f * ch;
    if (ch == 1) return (long long)q[0];
    if (ch == 2) return ((long long)q[0] + (long long)q[1]) >> 1;                  // floor(l * 0.5 + r * 0.5)
    long long s = 0;
    for (int c = 0; c < ch; ++c) {                                                   // pydub: converted[i] += sample // channels
        long long v = (long long)q[c], d = v / ch;
        s += (v % ch != 0 && v < 0) ? d - 1 : d;
    }
    return s;
}

template <typename T>
__device__ __forceinline__ float ingest_one(const T* __restrict__ p, int ch, long long n_frames, long long I, long long O, long long m) {
    const long long last = n_frames - 1;                                             // every frame index is clamped: a wrong n_out cannot read past the clip
    if (I == O) return (float)mono_frame(p, m < last ? m : last, ch);
    const int shift = 32 - 8 * (int)sizeof(T);
    const long long to32 = 1LL << shift;                                             // GETSAMPLE32: the sample at the top of a 32-bit word
    long long n1 = (m * I + O - 1) / O;                                              // the frame ratecv calls cur_i
    const long long d = n1 * O - m * I;                                              // in [0, O)
    if (n1 > last) n1 = last;
    const double cur = (double)(mono_frame(p, n1, ch) * to32);
    const double prev = n1 > 0 ? (double)(mono_frame(p, n1 - 1, ch) * to32) : 0.0;   // ratecv starts with prev_i = cur_i = 0
    // cur_o = (int)((prev_i * d + cur_i * (outrate - d)) / outrate), every operation rounded separately (no fma contraction)
    const double v = __ddiv_rn(__dadd_rn(__dmul_rn(prev, (double)d), __dmul_rn(cur, (double)(O - d))), (double)O);
    return (float)(T)((long long)v >> shift);
}

__global__ __launch_bounds__(256) void ingest_pcm_kernel(const unsigned char* __restrict__ pcm, const long long* __restrict__ desc,
                                                         float* __restrict__ out) {
    const long long* d = desc + 8 * blockIdx.y;
    const long long n_out = d[7];
    const int width = (int)d[1], ch = (int)d[2];
    const long long n_frames = d[3], I = d[4], O = d[5];
    float* y = out + d[6];
    const unsigned char* p = pcm + d[0];
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < n_out; m += (long long)gridDim.x * 256) {
        float v;
        if (width == 2) v = ingest_one((const int16_t*)p, ch, n_frames, I, O, m);
        else if (width == 4) v = ingest_one((const int32_t*)p, ch, n_frames, I, O, m);
        else v = ingest_one((const signed char*)p, ch, n_frames, I, O, m);
        y[m] = v;
    }
}

// ---------------------------------------------------------------- np.mean(float32 array): numpy's summation order
constexpr int NP_BUF = 8192;          // np.getbufsize(): add.reduce hands the inner loop one buffer at a time
constexpr int NP_LEAF = 128;          // PW_BLOCKSIZE of numpy's pairwise sum
constexpr int PW_DEPTH = 7;           // a buffer's tree is at most 7 deep: right child <= n / 2 + 8  ->  8192 / 128 + 16 <= 128

// node (k, p) of the pairwise tree over L elements: its length (0 = does not exist: an ancestor already is a leaf) and start
__device__ __forceinline__ int pw_node(int L, int k, int p, int& start) {
    int len = L;
    start = 0;
    for (int j = 0; j < k; ++j) {
        if (len <= NP_LEAF) return 0;
        const int n2 = (len >> 1) & ~7;
        if ((p >> (k - 1 - j)) & 1) { start += n2; len -= n2; } else len = n2;
    }
    return len;
}

__device__ __forceinline__ long long chunk_slot0(const long long* __restrict__ off, int clip) { return off[clip] / NP_BUF + clip; }

// chunk_sum[slot0(clip) + c] = pairwise sum of x[off[clip] + c * 8192 ...), exactly as numpy's FLOAT_pairwise_sum orders it
__global__ __launch_bounds__(256) void clip_chunk_sum_kernel(const float* __restrict__ x, const long long* __restrict__ off,
                                                             float* __restrict__ chunk_sum) {
    const int clip = blockIdx.y;
    const long long a = off[clip], n = off[clip + 1] - a;
    const long long n_chunks = (n + NP_BUF - 1) / NP_BUF;
    __shared__ float v[1 << PW_DEPTH];
    const int lane = threadIdx.x & 7;
    for (long long c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const float* xc = x + a + c * NP_BUF;
        const int L = (int)((n - c * NP_BUF) < NP_BUF ? (n - c * NP_BUF) : NP_BUF);
        for (int it = 0; it < 4; ++it) {
            const int slot = it * 32 + (threadIdx.x >> 3);                           // 7 path bits, most significant first
            int len = L, start = 0, k = 0;
            for (; k < PW_DEPTH && len > NP_LEAF; ++k) {
                const int n2 = (len >> 1) & ~7;
                if ((slot >> (PW_DEPTH - 1 - k)) & 1) { start += n2; len -= n2; } else len = n2;
            }
            const bool owner = (slot & ((1 << (PW_DEPTH - k)) - 1)) == 0;            // a leaf at depth k belongs to the slot with zero low bits
            if (!owner) len = 0;
            float r = 0.f;
            const int body = len - (len & 7);
            if (len >= 8) {                                                          // r[lane] = a[lane] + a[8 + lane] + ...
                r = xc[start + lane];
                for (int i = 8; i < body; i += 8) r += xc[start + i + lane];
            }
            r += __shfl_xor(r, 1);                                                   // (r0 + r1), (r2 + r3), ...
            r += __shfl_xor(r, 2);                                                   // ((r0 + r1) + (r2 + r3)), ...
            r += __shfl_xor(r, 4);
            if (lane == 0) {
                for (int i = (len >= 8 ? body : 0); i < len; ++i) r += xc[start + i];   // the tail (or a whole leaf of < 8) in order
                v[slot] = r;
            }
        }
        __syncthreads();
        for (int k = PW_DEPTH - 1; k >= 0; --k) {                                    // node = left child + right child, bottom up
            const int p = threadIdx.x;
            if (p < (1 << k)) {
                int st;
                if (pw_node(L, k, p, st) > NP_LEAF) v[p << (PW_DEPTH - k)] += v[(p << (PW_DEPTH - k)) + (1 << (PW_DEPTH - 1 - k))];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) chunk_sum[chunk_slot0(off, clip) + c] = v[0];
        __syncthreads();
    }
}

// stats[2 clip] = mean as numpy returns it: buffer sums accumulated in order (float32), divided by the count
__global__ void clip_mean_kernel(const long long* __restrict__ off, const float* __restrict__ chunk_sum, double* __restrict__ stats, int n_clips) {
    const int clip = blockIdx.x * blockDim.x + threadIdx.x;
    if (clip >= n_clips) return;
    const long long n = off[clip + 1] - off[clip];
    const long long n_chunks = (n + NP_BUF - 1) / NP_BUF;
    const float* cs = chunk_sum + chunk_slot0(off, clip);
    float s = 0.f;
    for (long long c = 0; c < n_chunks; ++c) s += cs[c];
    stats[2 * clip] = n > 0 ? (double)(float)((double)s / (double)n) : 0.0;
    ((unsigned*)(stats + 2 * clip + 1))[0] = 0u;
    ((unsigned*)(stats + 2 * clip + 1))[1] = 0u;
}

__global__ __launch_bounds__(256) void clip_peak_kernel(const float* __restrict__ x, const long long* __restrict__ off, double* __restrict__ stats) {
    const int clip = blockIdx.y;
    const long long a = off[clip], b = off[clip + 1];
    if (b <= a) return;
    const float mean = (float)stats[2 * clip];
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
    const float mean = (float)stats[2 * clip];
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

int vvk_ingest_pcm(const void* pcm, const long long* desc, int n_clips, long long max_out, float* out, hipStream_t st, const char** err) {
    if (n_clips <= 0 || max_out <= 0) { *err = "ingest_pcm: empty"; return -22; }
    long long bx = (max_out + 255) / 256;
    if (bx > 4096) bx = 4096;
    ingest_pcm_kernel<<<dim3((unsigned)bx, n_clips), 256, 0, st>>>((const unsigned char*)pcm, desc, out);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}

size_t vvk_normalize_scratch_bytes(int n_clips, long long total_len) {
    return sizeof(double) * 2 * (size_t)n_clips + sizeof(float) * (size_t)(total_len / NP_BUF + n_clips + 1);
}

int vvk_normalize_clips(const float* x, const long long* off, int n_clips, long long max_len, void* scratch, int16_t* out,
                        hipStream_t st, const char** err) {
    if (n_clips <= 0 || max_len <= 0) { *err = "normalize_clips: empty"; return -22; }
    if (max_len >= (1ll << 24)) { *err = "normalize_clips: clip of 2^24 samples or more (the float32 count of np.mean is exact below that)"; return -22; }
    double* stats = (double*)scratch;
    float* chunk_sum = (float*)(stats + 2 * (size_t)n_clips);
    const long long n_chunks = (max_len + NP_BUF - 1) / NP_BUF;
    clip_chunk_sum_kernel<<<dim3((unsigned)n_chunks, n_clips), 256, 0, st>>>(x, off, chunk_sum);
    clip_mean_kernel<<<(n_clips + 63) / 64, 64, 0, st>>>(off, chunk_sum, stats, n_clips);
    long long bx = (max_len + 256 * 8 - 1) / (256 * 8);
    if (bx > 256) bx = 256;
    dim3 grid((unsigned)bx, n_clips);
    clip_peak_kernel<<<grid, 256, 0, st>>>(x, off, stats);
    clip_scale_kernel<<<grid, 256, 0, st>>>(x, off, stats, out);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}
