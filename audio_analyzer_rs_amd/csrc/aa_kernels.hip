// gfx950 (MI355X / CDNA4) kernels for the STFT hot path.
//
// What they replace in the reference (paths relative to the reference repo):
//   window multiply      src/audio_io/stft.rs:296-299, src/analysis/onset.rs:254-257
//   forward real FFT     src/dsp/fft.rs:33-35, 66-71  (realfft/rustfft on the CPU)
//   magnitude            src/audio_io/stft.rs:314-318, src/analysis/onset.rs:271-272
//   inverse real FFT     src/dsp/fft.rs:39-41, 96-101
//
// Written for wave64 / CDNA4 only: DPP row operations, v_permlane{16,32}_swap and
// the 64-bank LDS are used directly.  No CUDA dual path.
#include "aa_internal.hpp"

namespace aa {
namespace {

// ------------------------------------------------------------------ helpers --
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 w)
{
    return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}
// a * (-i)
__device__ __forceinline__ float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }

template <int OUT>
__device__ __forceinline__ float bin_value(float re, float im, float log_floor)
{
    float p = re * re + im * im;
    if (OUT == kMag)
        return __builtin_amdgcn_sqrtf(p);
    if (OUT == kPower)
        return p;
    // 10*log10(p) = (10/log2(10)) * log2(p)
    return 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(p, log_floor));
}

// 8-point DFT over the register index, natural order in and out (radix-2 DIF,
// the bit-reversal is a compile-time renaming).
__device__ __forceinline__ void dft8(float2 (&v)[8])
{
    const float h = 0.70710678118654752440f;
    float2 a0 = cadd(v[0], v[4]), b0 = csub(v[0], v[4]);
    float2 a1 = cadd(v[1], v[5]), b1 = csub(v[1], v[5]);
    float2 a2 = cadd(v[2], v[6]), b2 = csub(v[2], v[6]);
    float2 a3 = cadd(v[3], v[7]), b3 = csub(v[3], v[7]);
    b1 = make_float2((b1.x + b1.y) * h, (b1.y - b1.x) * h);   // * W8^1
    b2 = cmul_mi(b2);                                         // * W8^2
    b3 = make_float2((b3.y - b3.x) * h, -(b3.x + b3.y) * h);  // * W8^3
    float2 c0 = cadd(a0, a2), c2 = csub(a0, a2);
    float2 c1 = cadd(a1, a3), c3 = cmul_mi(csub(a1, a3));
    float2 d0 = cadd(b0, b2), d2 = csub(b0, b2);
    float2 d1 = cadd(b1, b3), d3 = cmul_mi(csub(b1, b3));
    v[0] = cadd(c0, c1); v[4] = csub(c0, c1);
    v[2] = cadd(c2, c3); v[6] = csub(c2, c3);
    v[1] = cadd(d0, d1); v[5] = csub(d0, d1);
    v[3] = cadd(d2, d3); v[7] = csub(d2, d3);
}

// ---- cross-lane register transposes -----------------------------------------
// "Swap bit p of the register index with bit q of the lane index" for an array of
// 8 complex registers: for every pair (i, i|s), s = 1<<p, i&s == 0
//   lanes with lane-bit q == 1 :  r[i]   <- partner's r[i|s]
//   lanes with lane-bit q == 0 :  r[i|s] <- partner's r[i]        (partner = lane ^ (1<<q))
// DPP controls: quad_perm = 0x00-0xFF, row_shl:n = 0x100+n, row_shr:n = 0x110+n,
// row_ror:n = 0x120+n.
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_update(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src),
                                                      CTRL, 0xF, BANK_MASK, false));
}

// quad stages (lane bit 0 or 1): permute everything, then select on the lane bit.
template <int QUAD_PERM>
__device__ __forceinline__ void xpose_pair_quad(float &ri, float &rj, bool lane_bit)
{
    float ti = dpp_update<QUAD_PERM, 0xF>(ri, ri);   // partner's r[i]
    float tj = dpp_update<QUAD_PERM, 0xF>(rj, rj);   // partner's r[i|s]
    float ni = lane_bit ? tj : ri;
    float nj = lane_bit ? rj : ti;
    ri = ni;
    rj = nj;
}

// lane bit 2: row_shr:4 into banks 1,3 ; row_shl:4 into banks 0,2 (bank = 4 lanes).
__device__ __forceinline__ void xpose_pair_b2(float &ri, float &rj)
{
    float ni = dpp_update<0x114, 0xA>(ri, rj);   // lanes 4-7,12-15 read lane-4's r[i|s]
    float nj = dpp_update<0x104, 0x5>(rj, ri);   // lanes 0-3,8-11  read lane+4's r[i]
    ri = ni;
    rj = nj;
}

// lane bit 3: row_ror:8 (a rotation by half a row is its own inverse).
__device__ __forceinline__ void xpose_pair_b3(float &ri, float &rj)
{
    float ni = dpp_update<0x128, 0xC>(ri, rj);   // lanes 8-15 read lane-8's r[i|s]
    float nj = dpp_update<0x128, 0x3>(rj, ri);   // lanes 0-7  read lane+8's r[i]
    ri = ni;
    rj = nj;
}

// lane bit 4 / 5: v_permlane16_swap / v_permlane32_swap exchange the odd rows
// (upper half) of the first operand with the even rows (lower half) of the second.
__device__ __forceinline__ void xpose_pair_b4(float &ri, float &rj)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(ri), __float_as_uint(rj), false, false);
    ri = __uint_as_float(r[0]);
    rj = __uint_as_float(r[1]);
}
__device__ __forceinline__ void xpose_pair_b5(float &ri, float &rj)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(ri), __float_as_uint(rj), false, false);
    ri = __uint_as_float(r[0]);
    rj = __uint_as_float(r[1]);
}

// register bits 0,1,2 <-> lane bits 0,1,2
__device__ __forceinline__ void exchange_low(float2 (&v)[8], int lane)
{
    const bool l0 = lane & 1, l1 = lane & 2;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {   // p = 0 : pairs (i, i+1)
        xpose_pair_quad<0xB1>(v[i].x, v[i + 1].x, l0);
        xpose_pair_quad<0xB1>(v[i].y, v[i + 1].y, l0);
    }
#pragma unroll
    for (int g = 0; g < 8; g += 4)     // p = 1 : pairs (i, i+2)
#pragma unroll
        for (int i = g; i < g + 2; ++i) {
            xpose_pair_quad<0x4E>(v[i].x, v[i + 2].x, l1);
            xpose_pair_quad<0x4E>(v[i].y, v[i + 2].y, l1);
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // p = 2 : pairs (i, i+4)
        xpose_pair_b2(v[i].x, v[i + 4].x);
        xpose_pair_b2(v[i].y, v[i + 4].y);
    }
}

// register bits 0,1,2 <-> lane bits 3,4,5
__device__ __forceinline__ void exchange_high(float2 (&v)[8])
{
#pragma unroll
    for (int i = 0; i < 8; i += 2) {   // p = 0 <-> lane bit 3
        xpose_pair_b3(v[i].x, v[i + 1].x);
        xpose_pair_b3(v[i].y, v[i + 1].y);
    }
#pragma unroll
    for (int g = 0; g < 8; g += 4)     // p = 1 <-> lane bit 4
#pragma unroll
        for (int i = g; i < g + 2; ++i) {
            xpose_pair_b4(v[i].x, v[i + 2].x);
            xpose_pair_b4(v[i].y, v[i + 2].y);
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // p = 2 <-> lane bit 5
        xpose_pair_b5(v[i].x, v[i + 4].x);
        xpose_pair_b5(v[i].y, v[i + 4].y);
    }
}

// ------------------------------------------------ wave-per-frame, n_fft = 1024 --
// One wavefront transforms one frame: the 1024 windowed reals are read as 512
// complex points z[m] (8 per lane), 512 = 8 x 8 x 8:
//   pass A  radix-8 over j      (m = 64 j + n2, n2 = 8a + b; lane = a + 8b), twiddle W_512^(n2 k1)
//   E1      registers <-> lane bits 0-2 (DPP)                    -> lane = k1 + 8b, regs a
//   pass B  radix-8 over a, twiddle W_64^(b ka)
//   E2      registers <-> lane bits 3-5 (DPP row_ror + permlane swaps) -> lane = k1 + 8 ka, regs b
//   pass C  radix-8 over b      -> lane L holds Z[L + 64 kb], kb = 0..7
//   split   X[k], X[512-k] from Z[k], Z[512-k]; the partner half travels through a
//           wave-private LDS strip (4 ds_write_b64 + 4 ds_read_b64 per lane).
// A workgroup (4 waves) stages the samples of `frames_per_wg` consecutive frames in
// LDS once (each sample is used by n_fft/hop frames), padded by 8 dwords per 64 so
// that the lane = a + 8b read pattern is bank-conflict free for ds_read_b64.
// The window table carries a factor 0.5 (exact) so the split needs no halving.
template <int OUT>
__global__ __launch_bounds__(kWave1024Threads) void stft_wave1024_kernel(
    const float *__restrict__ x, float *__restrict__ out, const float *__restrict__ tab,
    int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg, float log_floor)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // XCD-aware chunk id: blocks b, b+8, b+16.. share an XCD (round-robin dispatch), so
    // give each XCD a contiguous run of chunks - neighbouring chunks share their halo
    // (n_fft - hop samples) through that XCD's L2.  Bijective for any grid size.
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int chunk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);

    const int64_t f0 = (int64_t)chunk * frames_per_wg;
    const int64_t left = n_frames - f0;
    const int nf = left < frames_per_wg ? (int)left : frames_per_wg;
    const float *src = x + (frame_begin + f0) * (int64_t)hop;
    const int span4 = ((nf - 1) * hop + 1024) >> 2;
    const int max_span = (frames_per_wg - 1) * hop + 1024;
    float *xch = smem + (max_span + 8 * ((max_span + 63) >> 6)) + wave * kWave1024XchDwords;

    // stage: float4 q -> dword 4q + 8*(q>>4)
    for (int q = tid; q < span4; q += kWave1024Threads) {
        float4 s = reinterpret_cast<const float4 *>(src)[q];
        *reinterpret_cast<float4 *>(&smem[4 * q + 8 * (q >> 4)]) = s;
    }

    // per-lane constants (table rows are lane-contiguous)
    float win[16];
    float2 twA[8], twB[8], tws[4];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        win[i] = tab[i * 64 + lane];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        twA[k] = make_float2(tab[(16 + 2 * (k - 1)) * 64 + lane], tab[(17 + 2 * (k - 1)) * 64 + lane]);
        twB[k] = make_float2(tab[(30 + 2 * (k - 1)) * 64 + lane], tab[(31 + 2 * (k - 1)) * 64 + lane]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        tws[j] = make_float2(tab[(44 + 2 * j) * 64 + lane], tab[(45 + 2 * j) * 64 + lane]);

    const int n2 = 8 * (lane & 7) + (lane >> 3);
    const int lane_off = 2 * n2 + 8 * (n2 >> 5);
    constexpr int NB = 513;
    constexpr int COLS = (OUT == kComplex) ? 2 * NB : NB;

    __syncthreads();

    for (int t = wave; t < nf; t += 4) {
        const int fb = t * hop;                       // multiple of 64
        const float *fr = smem + fb + (fb >> 3) + lane_off;
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float2 s = *reinterpret_cast<const float2 *>(fr + 144 * j);
            v[j] = make_float2(s.x * win[2 * j], s.y * win[2 * j + 1]);
        }
        dft8(v);
#pragma unroll
        for (int k = 1; k < 8; ++k)
            v[k] = cmul(v[k], twA[k]);
        exchange_low(v, lane);
        dft8(v);
#pragma unroll
        for (int k = 1; k < 8; ++k)
            v[k] = cmul(v[k], twB[k]);
        exchange_high(v);
        dft8(v);

        // ---- split: partner half through the wave-private LDS strip --------------
        // strip[c] (complex) holds Z[256 + c], c = 0..255, and strip[256] = Z[512] := Z[0]
        float2 *strip = reinterpret_cast<float2 *>(xch);
#pragma unroll
        for (int j = 4; j < 8; ++j)
            strip[lane + 64 * (j - 4)] = v[j];
        if (lane == 0)
            strip[256] = v[0];
        __builtin_amdgcn_wave_barrier();
        float *row = out + (f0 + t) * (int64_t)COLS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = lane + 64 * j;
            float2 a = v[j];
            float2 p = strip[256 - k];
            float2 bc = make_float2(p.x, -p.y);
            float2 e = cadd(a, bc), o = csub(a, bc);
            float2 tt = cmul(o, tws[j]);
            float2 xk = cadd(e, tt);
            float2 xm = csub(e, tt);                  // X[512-k] = conj(xm)
            if (OUT == kComplex) {
                reinterpret_cast<float2 *>(row)[k] = xk;
                reinterpret_cast<float2 *>(row)[512 - k] = make_float2(xm.x, -xm.y);
            } else {
                row[k] = bin_value<OUT>(xk.x, xk.y, log_floor);
                row[512 - k] = bin_value<OUT>(xm.x, xm.y, log_floor);
            }
        }
        if (lane == 0) {                              // k = 256 pairs with itself
            float2 z = v[4];
            if (OUT == kComplex)
                reinterpret_cast<float2 *>(row)[256] = make_float2(2.0f * z.x, -2.0f * z.y);
            else
                row[256] = bin_value<OUT>(2.0f * z.x, 2.0f * z.y, log_floor);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------- generic forward paths --
__device__ __forceinline__ float2 split_tw(const float2 *tw_split, int k, int m)
{
    // W_n^k for 0 <= k <= m from the table of k <= m/2:  W_n^k = -conj(W_n^(m-k))
    if (2 * k <= m)
        return tw_split[k];
    float2 w = tw_split[m - k];
    return make_float2(-w.x, w.y);
}

template <int OUT>
__device__ __forceinline__ void store_bin(float *row, int k, float re, float im, float log_floor)
{
    if (OUT == kComplex)
        reinterpret_cast<float2 *>(row)[k] = make_float2(re, im);
    else
        row[k] = bin_value<OUT>(re, im, log_floor);
}

// Real-FFT split of an m-point complex spectrum Z (in LDS) into n/2+1 bins.
template <int OUT>
__device__ __forceinline__ void split_and_store(const float2 *Z, const float2 *tw_split, int m,
                                                float *row, float log_floor)
{
    for (int k = threadIdx.x; 2 * k <= m; k += blockDim.x) {
        if (k == 0) {
            float2 z = Z[0];
            store_bin<OUT>(row, 0, z.x + z.y, 0.0f, log_floor);
            store_bin<OUT>(row, m, z.x - z.y, 0.0f, log_floor);
            continue;
        }
        const int kk = m - k;
        float2 a = Z[k], p = Z[kk];
        float2 bc = make_float2(p.x, -p.y);
        float2 e = make_float2(0.5f * (a.x + bc.x), 0.5f * (a.y + bc.y));
        float2 o = make_float2(0.5f * (a.x - bc.x), 0.5f * (a.y - bc.y));
        float2 wo = cmul(o, tw_split[k]);
        float2 tt = make_float2(wo.y, -wo.x);          // -i * W * O
        store_bin<OUT>(row, k, e.x + tt.x, e.y + tt.y, log_floor);
        store_bin<OUT>(row, kk, e.x - tt.x, -(e.y - tt.y), log_floor);
    }
}

// LDS Stockham radix-2, one workgroup per frame, any power-of-two n >= 4.
template <int OUT>
__global__ void stft_lds_radix2_kernel(const float *__restrict__ x, float *__restrict__ out,
                                       const float *__restrict__ window,
                                       const float2 *__restrict__ tw_m,
                                       const float2 *__restrict__ tw_split, int n,
                                       int64_t frame_begin, int64_t hop, float log_floor)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int m = n >> 1;
    float2 *A = reinterpret_cast<float2 *>(smem);
    float2 *B = A + m;
    const int64_t t = blockIdx.x;
    const float *src = x + (frame_begin + t) * hop;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        float a = src[2 * i], b = src[2 * i + 1];
        if (window) {
            a *= window[2 * i];
            b *= window[2 * i + 1];
        }
        A[i] = make_float2(a, b);
    }
    __syncthreads();
    const int half = m >> 1;
    for (int ns = 1; ns < m; ns <<= 1) {
        const int tstep = half / ns;
        for (int i = threadIdx.x; i < half; i += blockDim.x) {
            const int k = i & (ns - 1);
            float2 a = A[i];
            float2 b = cmul(A[i + half], tw_m[k * tstep]);
            const int j = ((i - k) << 1) + k;
            B[j] = cadd(a, b);
            B[j + ns] = csub(a, b);
        }
        __syncthreads();
        float2 *tmp = A;
        A = B;
        B = tmp;
    }
    const int cols = (OUT == kComplex) ? 2 * (m + 1) : (m + 1);
    split_and_store<OUT>(A, tw_split, m, out + t * (int64_t)cols, log_floor);
}

// Direct O(m^2) DFT for even lengths that are not powers of two (the reference's
// realfft accepts any length; it only ever instantiates 2048 and 256).
template <int OUT>
__global__ void stft_dft_kernel(const float *__restrict__ x, float *__restrict__ out,
                                const float *__restrict__ window, const float2 *__restrict__ tw_m,
                                const float2 *__restrict__ tw_split, int n, int64_t frame_begin,
                                int64_t hop, float log_floor)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int m = n >> 1;
    float2 *A = reinterpret_cast<float2 *>(smem);
    float2 *B = A + m;
    const int64_t t = blockIdx.x;
    const float *src = x + (frame_begin + t) * hop;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        float a = src[2 * i], b = src[2 * i + 1];
        if (window) {
            a *= window[2 * i];
            b *= window[2 * i + 1];
        }
        A[i] = make_float2(a, b);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < m; k += blockDim.x) {
        float2 acc = make_float2(0.0f, 0.0f);
        int idx = 0;
        for (int j = 0; j < m; ++j) {
            float2 w = tw_m[idx];
            float2 a = A[j];
            acc.x += a.x * w.x - a.y * w.y;
            acc.y += a.x * w.y + a.y * w.x;
            idx += k;
            if (idx >= m)
                idx -= m;
        }
        B[k] = acc;
    }
    __syncthreads();
    const int cols = (OUT == kComplex) ? 2 * (m + 1) : (m + 1);
    split_and_store<OUT>(B, tw_split, m, out + t * (int64_t)cols, log_floor);
}

// ------------------------------------------------------------- generic inverse --
// spec row: m+1 complex; out row: n reals.  Z[k] = (X[k] + conj(X[m-k])) + i conj(W_n^k)
// (X[k] - conj(X[m-k])), then an unnormalised inverse m-point complex FFT.
template <bool POW2>
__global__ void irfft_kernel(const float *__restrict__ spec, float *__restrict__ out,
                             const float2 *__restrict__ tw_m, const float2 *__restrict__ tw_split,
                             int n)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int m = n >> 1;
    float2 *A = reinterpret_cast<float2 *>(smem);
    float2 *B = A + m;
    const int64_t t = blockIdx.x;
    const float2 *S = reinterpret_cast<const float2 *>(spec + t * (int64_t)(2 * (m + 1)));
    for (int k = threadIdx.x; k < m; k += blockDim.x) {
        const int kk = m - k;
        float2 a = S[k], p = S[kk];
        if (k == 0) {
            a.y = 0.0f;   // realfft ignores/rejects Im X[0], Im X[n/2]
            p.y = 0.0f;
        }
        float2 bc = make_float2(p.x, -p.y);
        float2 e = cadd(a, bc), o = csub(a, bc);
        float2 w = split_tw(tw_split, k, m);
        float2 wc = make_float2(w.x, -w.y);
        float2 wo = cmul(o, wc);
        A[k] = make_float2(e.x - wo.y, e.y + wo.x);   // e + i*wo
    }
    __syncthreads();
    float2 *Z;
    if (POW2) {
        const int half = m >> 1;
        for (int ns = 1; ns < m; ns <<= 1) {
            const int tstep = half / ns;
            for (int i = threadIdx.x; i < half; i += blockDim.x) {
                const int k = i & (ns - 1);
                float2 w = tw_m[k * tstep];
                float2 a = A[i];
                float2 b = cmul(A[i + half], make_float2(w.x, -w.y));
                const int j = ((i - k) << 1) + k;
                B[j] = cadd(a, b);
                B[j + ns] = csub(a, b);
            }
            __syncthreads();
            float2 *tmp = A;
            A = B;
            B = tmp;
        }
        Z = A;
    } else {
        for (int k = threadIdx.x; k < m; k += blockDim.x) {
            float2 acc = make_float2(0.0f, 0.0f);
            int idx = 0;
            for (int j = 0; j < m; ++j) {
                float2 w = tw_m[idx];
                float2 a = A[j];
                acc.x += a.x * w.x + a.y * w.y;   // conj twiddle
                acc.y += a.y * w.x - a.x * w.y;
                idx += k;
                if (idx >= m)
                    idx -= m;
            }
            B[k] = acc;
        }
        __syncthreads();
        Z = B;
    }
    float2 *orow = reinterpret_cast<float2 *>(out + t * (int64_t)n);
    for (int j = threadIdx.x; j < m; j += blockDim.x)
        orow[j] = Z[j];
}

}  // namespace

// ------------------------------------------------------------------- launchers --
hipError_t launch_wave1024(int out_kind, const float *x, float *out, const float *table,
                           int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg,
                           float log_floor, hipStream_t stream)
{
    if (n_frames <= 0)
        return hipSuccess;
    const int64_t chunks = (n_frames + frames_per_wg - 1) / frames_per_wg;
    const size_t lds = wave1024_lds_bytes(frames_per_wg, hop);
    dim3 grid((unsigned)chunks), block(kWave1024Threads);
#define AA_LAUNCH_W1024(K)                                                                     \
    hipLaunchKernelGGL((stft_wave1024_kernel<K>), grid, block, lds, stream, x, out, table,     \
                       frame_begin, n_frames, hop, frames_per_wg, log_floor)
    switch (out_kind) {
    case kComplex: AA_LAUNCH_W1024(kComplex); break;
    case kMag: AA_LAUNCH_W1024(kMag); break;
    case kPower: AA_LAUNCH_W1024(kPower); break;
    case kLogPower: AA_LAUNCH_W1024(kLogPower); break;
    default: return hipErrorInvalidValue;
    }
#undef AA_LAUNCH_W1024
    return hipGetLastError();
}

hipError_t launch_generic_forward(bool pow2, int out_kind, const float *x, float *out,
                                  const float *window, const float *tw_m, const float *tw_split,
                                  int n, int64_t frame_begin, int64_t n_frames, int64_t hop,
                                  float log_floor, hipStream_t stream)
{
    if (n_frames <= 0)
        return hipSuccess;
    const int m = n / 2;
    const size_t lds = sizeof(float2) * 2 * (size_t)m;
    dim3 grid((unsigned)n_frames), block(m >= 512 ? 256 : 64);
    const float2 *twm = reinterpret_cast<const float2 *>(tw_m);
    const float2 *tws = reinterpret_cast<const float2 *>(tw_split);
#define AA_LAUNCH_GEN(KERNEL, K)                                                               \
    hipLaunchKernelGGL((KERNEL<K>), grid, block, lds, stream, x, out, window, twm, tws, n,     \
                       frame_begin, hop, log_floor)
#define AA_SWITCH_GEN(KERNEL)                                                                  \
    switch (out_kind) {                                                                        \
    case kComplex: AA_LAUNCH_GEN(KERNEL, kComplex); break;                                     \
    case kMag: AA_LAUNCH_GEN(KERNEL, kMag); break;                                             \
    case kPower: AA_LAUNCH_GEN(KERNEL, kPower); break;                                         \
    case kLogPower: AA_LAUNCH_GEN(KERNEL, kLogPower); break;                                   \
    default: return hipErrorInvalidValue;                                                      \
    }
    if (pow2) {
        AA_SWITCH_GEN(stft_lds_radix2_kernel)
    } else {
        AA_SWITCH_GEN(stft_dft_kernel)
    }
#undef AA_SWITCH_GEN
#undef AA_LAUNCH_GEN
    return hipGetLastError();
}

hipError_t launch_generic_inverse(bool pow2, const float *spec, float *out, const float *tw_m,
                                  const float *tw_split, int n, int64_t batch, hipStream_t stream)
{
    if (batch <= 0)
        return hipSuccess;
    const int m = n / 2;
    const size_t lds = sizeof(float2) * 2 * (size_t)m;
    dim3 grid((unsigned)batch), block(m >= 512 ? 256 : 64);
    const float2 *twm = reinterpret_cast<const float2 *>(tw_m);
    const float2 *tws = reinterpret_cast<const float2 *>(tw_split);
    if (pow2)
        hipLaunchKernelGGL((irfft_kernel<true>), grid, block, lds, stream, spec, out, twm, tws, n);
    else
        hipLaunchKernelGGL((irfft_kernel<false>), grid, block, lds, stream, spec, out, twm, tws, n);
    return hipGetLastError();
}

}  // namespace aa
