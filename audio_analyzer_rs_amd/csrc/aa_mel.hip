// Build-defined post-ops on the power spectrogram (ABSENT from the reference — SURVEY.md
// §8a rows a10/a11; parity is pinned only against the build's own float64 definition in
// oracle/truth.py):
//   mel     [frames x n_bins] . [n_bins x n_mels]  on the f32 matrix cores
//           (v_mfma_f32_32x32x2_f32: exact f32 fma chain — bf16/f16 inputs would break the
//           1e-5 budget), skipping the K range where a 32-mel column tile is all zero
//           (triangular filters: each tile touches a narrow band of bins);
//   log     10*log10(max(mel, floor)) in the GEMM epilogue;
//   DCT-II  orthonormal, [n_mels -> n_mfcc]: the same GEMM kernel with the DCT matrix as B.
// Also the stereo -> mono average of the capture callback (src/audio_io/mod.rs:784-794).
#include "aa_internal.hpp"
#include <type_traits>

namespace aa {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kGemmK = 32;              // bins staged per LDS round
constexpr int kGemmRows = 128;          // frames per workgroup (32 per wave)
constexpr int kGemmLdA = kGemmK + 1;    // +1: the 32 rows one MFMA operand reads sit on 32 banks

// C[F x n_cols] = op(A[F x K] . B[K x n_cols]) on the f32 matrix cores, B zero padded to
// [k_pad x n_pad] (k_pad % 32 == 0, n_pad % 32 == 0).  tile_k holds, per 32-column tile of B,
// the [k_lo, k_hi) range outside of which the tile is all zero (multiples of 32): triangular
// mel filters touch a narrow band of bins, so most (tile, K-chunk) pairs are skipped.
// Workgroup = 4 waves = 128 rows x (n_pad <= 128) columns; wave w owns rows 32w..32w+31 and one
// 32x32 accumulator per column tile.  Exact f32: the MFMA is a k-ordered fmaf chain.
template <bool LOG>
__global__ __launch_bounds__(256) void rows_gemm_mfma_kernel(const float *__restrict__ A, int lda,
                                                             const float *__restrict__ B,
                                                             const int2 *__restrict__ tile_k,
                                                             float *__restrict__ C, int64_t n_rows,
                                                             int K, int n_cols, int n_pad,
                                                             float log_floor)
{
    __shared__ float As[kGemmRows * kGemmLdA];
    __shared__ float Bs[kGemmK * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * kGemmRows;
    const int n_tiles = n_pad >> 5;                       // <= 4
    int2 rng[4];
    int k_lo = 1 << 30, k_hi = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        rng[c] = c < n_tiles ? tile_k[c] : make_int2(0, 0);
        if (c < n_tiles && rng[c].y > rng[c].x) {
            k_lo = rng[c].x < k_lo ? rng[c].x : k_lo;
            k_hi = rng[c].y > k_hi ? rng[c].y : k_hi;
        }
    }
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        acc[c] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int ai = lane & 31, ah = lane >> 5;
    const float *a_row = As + (32 * wave + ai) * kGemmLdA + ah;
    const float *b_col = Bs + ah * 128 + ai;

    // Software pipeline: the global loads of chunk i+1 are issued before the MFMAs of chunk i
    // and land in registers (16 + 16 VGPRs); one LDS write + barrier pair per chunk.
    float ra[16];
    float4 rb[4];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u;
            const int r = e >> 5, k = e & 31;
            const int64_t gr = row0 + r;
            ra[u] = (gr < n_rows && kb + k < K) ? A[gr * lda + kb + k] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;
            const int k = e >> 5, c4 = e & 31;
            rb[u] = (4 * c4 < n_pad) ? *reinterpret_cast<const float4 *>(B + (int64_t)(kb + k) * n_pad + 4 * c4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (k_lo < k_hi)
        fetch(k_lo);
    for (int kb = k_lo; kb < k_hi; kb += kGemmK) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u;
            As[(e >> 5) * kGemmLdA + (e & 31)] = ra[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;
            *reinterpret_cast<float4 *>(Bs + (e >> 5) * 128 + 4 * (e & 31)) = rb[u];
        }
        __syncthreads();
        if (kb + kGemmK < k_hi)
            fetch(kb + kGemmK);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < n_tiles && kb + kGemmK > rng[c].x && kb < rng[c].y) {
#pragma unroll
                for (int k = 0; k < kGemmK; k += 2)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_row[k], b_col[k * 128 + 32 * c], acc[c], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 f32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = 32 * c + ai;
        if (c >= n_tiles || col >= n_cols)
            continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t gr = row0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * ah;
            if (gr < n_rows) {
                float v = acc[c][r];
                if (LOG)
                    v = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(v, log_floor));
                C[gr * n_cols + col] = v;
            }
        }
    }
}

// Onset front end (src/analysis/onset.rs:261-291), first of the "next" rows after the STFT:
// per frame the 3-bin-smoothed, frequency-weighted positive spectral flux against the previous
// frame's magnitudes and the frame energy.
// r5: ONE LANE PER FRAME, walking the bins in order - the reference's own sequential f32 sums (fp contract off, IEEE
// division by 3, the weight 1 - i / n evaluated once per bin with the reference's expression), so flux and energy are
// BIT-IDENTICAL to the CPU restatement; rounds 1-4 ran one wave per frame (a tree sum: 1e-5, four dependent dword loads
// per element, a binary search over the stream offsets per frame and lane: 348 us for the 450 k frames of ten minutes).
// A wave takes blocks of 64 consecutive frames: the 65 rows a block needs (its own and the one before it) are ONE
// contiguous span of the spectrogram, fetched as coalesced 16-byte pieces into registers - the next block's while the
// current one is walked - and handed over through LDS, where lane t walks row t at an odd stride (n_bins = n_fft/2 + 1:
// conflict free).  Every element leaves HBM once.  Spectra wider than 129 bins (the onset geometry) go through LDS in
// column tiles of 112 bins with the same accumulators (FLAT = false).
// offsets (nullable): frame_offsets[n_streams + 1] of concatenated streams; the first frame of a stream takes
// prev0 + s * n_bins (or zeros) as its previous row.  The stream of a block's first frame is found ONCE per block by a
// 64-ary search across the lanes, the starts inside the block are scattered into an LDS row (ds_max: of several empty
// streams that start on one frame the last one owns it).
constexpr int kFluxT = 64;                          // frames per block: one per lane
constexpr int kFluxV4 = 33;                         // 16-byte pieces per lane of a flat span: 65 * 129 + 3 floats
constexpr int kFluxCh = 16;                         // bins per batch of LDS reads (a lone wave must not wait per bin)
constexpr int kFluxTileBins = 7 * kFluxCh;          // interior bins per column tile: chunk starts stay 1 mod 16
constexpr int kFluxTileStride = kFluxTileBins + 3;  // bins kb .. kb + 113 (+ 1: odd stride), 115 floats per row
// x / 3.0f, correctly rounded, in four instructions instead of the ten of the IEEE division sequence: q0 = x * RN(1/3),
// the exact remainder x - 3 q0 by one FMA, one FMA to correct, and the division's own last step, v_div_fixup_f32, for the
// three inputs the remainder cannot serve (+-inf: its remainder is NaN; -0: the add loses the sign).  Equal to the
// division for EVERY f32 bit pattern: tests/test_oracle.py::test_div3_core_is_the_division runs all 2^32 through the C
// copy of the three-operation core (exactly those three differ), tests/test_gpu_parity.py::test_flux_div3_exhaustive all
// 2^32 through this function on the device.
__device__ __forceinline__ float flux_div3(float x)
{
    const float r = 0x1.555556p-2f;
    const float q0 = x * r;
    const float e = __builtin_fmaf(-3.0f, q0, x);
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(e, r, q0), 3.0f, x);
}
__global__ void flux_div3_check_kernel(uint32_t first, uint64_t n, unsigned long long *mismatches)
{
#pragma clang fp contract(off)
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(first + (uint32_t)i);
        const float a = flux_div3(x), b = x / 3.0f;
        bad += (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) ? 1 : 0;
    }
    if (bad)
        atomicAdd(mismatches, bad);
}
// The bins [k0, k1) of one tile, in order, for this lane's frame: bin i at cur[i - kb], the frame before it at prv[i - kb]
// (first: the frame starts a stream - its previous row is p0, or zeros).  Bins 0 and n_bins - 1 are not smoothed
// (onset.rs:264-268); the interior bins run in batches of 16 whose LDS reads are all issued up front at immediate
// offsets - no branch and no wait inside a batch.  Batches start at bins 1 mod 16 (their weights: aligned float4s).
template <bool ANYFIRST>
__device__ __forceinline__ void flux_walk(const float *cur, const float *prv, bool first, const float *p0,
                                          const float *wsh, int kb, int k0, int k1, int n_bins, float &fl, float &en)
{
#pragma clang fp contract(off)
    const int last = n_bins - 1;
    const float *pq = (ANYFIRST && first) ? cur : prv;           // (a stream's first frame has no row before it in the tile)
    auto prev_of = [&](int i) {
        if (!ANYFIRST)
            return prv[i - kb];
        const float own = (first && p0) ? p0[i] : 0.0f;
        return first ? own : pq[i - kb];
    };
    auto end_bin = [&](int i) {                                  // an end bin: the magnitude itself
        const float m = cur[i - kb];
        en += m;
        fl += fmaxf(m - prev_of(i), 0.0f) * (i == 0 ? 1.0f : wsh[i - 1]);
    };
    if (k0 == 0)
        end_bin(0);
    const int i_lo = k0 > 1 ? k0 : 1, i_hi = k1 < last ? k1 : last;
    if (i_lo < i_hi) {
        float m_lo = cur[i_lo - 1 - kb], m_i = cur[i_lo - kb];
        float c[kFluxCh], pv[kFluxCh];
        v4f w[kFluxCh / 4];
        auto step = [&](int j) {
            const float m_hi = c[j];
            en += m_i;
            const float sm = flux_div3(m_lo + m_i + m_hi);
            fl += fmaxf(sm - pv[j], 0.0f) * w[j >> 2][j & 3];
            m_lo = m_i;
            m_i = m_hi;
        };
        // (issuing the next batch's reads in front of the current batch's arithmetic was measured: +-0, twice the code)
        int i = i_lo;
        for (; i + kFluxCh <= i_hi; i += kFluxCh) {
            const float *cp = cur + (i + 1 - kb), *pp = pq + (i - kb);
#pragma unroll
            for (int j = 0; j < kFluxCh; ++j)
                c[j] = cp[j];
#pragma unroll
            for (int j = 0; j < kFluxCh; ++j) {
                if (ANYFIRST) {
                    const float own = (first && p0) ? p0[i + j] : 0.0f;
                    pv[j] = first ? own : pp[j];
                } else {
                    pv[j] = pp[j];
                }
            }
#pragma unroll
            for (int j = 0; j < kFluxCh / 4; ++j)
                w[j] = *reinterpret_cast<const v4f *>(wsh + i - 1 + 4 * j);
#pragma unroll
            for (int j = 0; j < kFluxCh; ++j)
                step(j);
        }
        if (i < i_hi) {                                          // the last, short batch: reads past it stay in the row
            const int cnt = i_hi - i;
#pragma unroll
            for (int j = 0; j < kFluxCh; ++j) {
                const int ic = i + 1 + j < last ? i + 1 + j : last, ip = i + j < last ? i + j : last;
                c[j] = cur[ic - kb];
                pv[j] = prev_of(ip);
            }
#pragma unroll
            for (int j = 0; j < kFluxCh / 4; ++j)
                w[j] = *reinterpret_cast<const v4f *>(wsh + i - 1 + 4 * j);
#pragma unroll
            for (int j = 0; j < kFluxCh; ++j)
                if (j < cnt)
                    step(j);
        }
    }
    if (k1 == n_bins && last > 0)
        end_bin(last);
}
template <bool FLAT>
__global__ __launch_bounds__(64) void onset_flux_kernel(const float *__restrict__ mags,
                                                        const float *__restrict__ prev0,
                                                        float *__restrict__ flux,
                                                        float *__restrict__ energy, int64_t n_frames,
                                                        int n_bins, const int64_t *__restrict__ offsets,
                                                        int n_streams)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float flux_lds[];
    const int lane = threadIdx.x;
    float *tile = flux_lds;
    // wsh[q] = the reference's weight of bin q + 1 (chunks start at bins 1 mod 16: their weights are aligned float4s),
    // w of bin 0 is 1 - 0 / n = 1 exactly
    float *wsh = tile + (FLAT ? 4 * 64 * kFluxV4 : (kFluxT + 1) * kFluxTileStride + 1);
    const int wsh_len = ((n_bins + 3) & ~3) + kFluxCh;
    int *starts = reinterpret_cast<int *>(wsh + wsh_len);       // [64] stream that starts at the block's frame (+ 1)
    // a row of zeros: the "previous row" of a stream's first frame when the caller hands none over (prev0 == NULL) - the
    // frame then walks like any other (no select per bin)
    float *zrow = reinterpret_cast<float *>(starts + 64);
    const int zrow_len = FLAT ? wsh_len : kFluxTileStride + 1;
    for (int q = lane; q < zrow_len; q += 64)
        zrow[q] = 0.0f;
    for (int q = lane; q < wsh_len; q += 64)
        wsh[q] = q + 1 < n_bins ? 1.0f - ((float)(q + 1) / (float)n_bins) : 0.0f;
    // every wave owns a contiguous run of blocks (consecutive blocks share a row, and the stream cursor below moves on
    // by one probe per block)
    const int64_t n_blocks = (n_frames + kFluxT - 1) / kFluxT;
    const int64_t per = (n_blocks + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
    if (b0 >= b1)
        return;
    const uintptr_t mags_addr = reinterpret_cast<uintptr_t>(mags);

    // ---- the flat span of a block: floats [s0, s0 + n) of the spectrogram, fetched from the 16-byte boundary below s0.
    // An interior block (a frame in front of it and more than a row behind it in the caller's buffer) is 33 unconditional
    // 16-byte loads per lane; at the two ends of the spectrogram pieces that lie inside the span are 16-byte loads and
    // the (at most two) pieces that straddle an end of it are fetched float by float by lanes 0-3: nothing outside the
    // caller's rows is ever read ----
    v4f regs[FLAT ? kFluxV4 : 1];
    float edge_head = 0.0f, edge_tail = 0.0f;
    auto span_of = [&](int64_t b, int64_t &s0, int64_t &n, int &lead, bool &interior) {
        const int64_t t0 = b * kFluxT;
        s0 = (t0 > 0 ? t0 - 1 : 0) * n_bins;
        const int64_t te = t0 + kFluxT < n_frames ? t0 + kFluxT : n_frames;
        n = te * n_bins - s0;
        lead = (int)(((mags_addr + 4u * (uint64_t)s0) & 15u) >> 2);
        // (33 pieces per lane reach 4 * 64 * 33 floats past the boundary: inside the buffer if that many rows follow)
        interior = t0 > 0 && (n_frames - (t0 - 1)) * (int64_t)n_bins >= 4 * 64 * kFluxV4;
    };
    auto fetch_flat = [&](int64_t b) {
        int64_t s0, n;
        int lead;
        bool interior;
        span_of(b, s0, n, lead, interior);
        const float *base = mags + s0 - lead;                    // 16-byte aligned
        if (interior) {
            const v4f *bp = reinterpret_cast<const v4f *>(base) + lane;
#pragma unroll
            for (int u = 0; u < kFluxV4; ++u)
                regs[u] = bp[64 * u];
            return;
        }
#pragma unroll
        for (int u = 0; u < kFluxV4; ++u) {
            const int64_t j = 4 * (int64_t)(lane + 64 * u) - lead;   // first float of the piece, relative to s0
            regs[u] = (j >= 0 && j + 4 <= n) ? *reinterpret_cast<const v4f *>(base + 4 * (lane + 64 * u)) : v4f{0.f, 0.f, 0.f, 0.f};
        }
        const int64_t jh = lane - lead, jt = ((lead + n) & ~(int64_t)3) - lead + lane;
        edge_head = (lane < 4 && jh >= 0 && jh < n) ? mags[s0 + jh] : 0.0f;
        edge_tail = (lane < 4 && jt >= 0 && jt < n) ? mags[s0 + jt] : 0.0f;
    };
    if (FLAT)
        fetch_flat(b0);

    // ---- the stream cursor (batches): plo = the last stream that starts at or in front of the block's first frame,
    // probe = offsets[plo + lane] - the starts that can fall into the block.  Found once per wave by a 64-ary search
    // across the lanes (under the first fetch), then moved on by ONE probe per block, itself fetched a block ahead ----
    int plo = 0;
    int64_t probe = INT64_MAX;
    auto load_probe = [&](int lo) { return lo + lane < n_streams ? offsets[lo + lane] : INT64_MAX; };
    if (offsets) {
        const int64_t t0 = b0 * kFluxT;
        int lo = 0, hi = n_streams;                              // the last stream with offsets[s] <= t0 is in [lo, hi)
        while (hi - lo > 1) {
            const int step = (hi - lo + 63) >> 6;
            const int idx = lo + lane * step;
            const bool le = idx < hi && offsets[idx] <= t0;
            const int c = __popcll(__ballot(le));                // monotone: a prefix of the lanes (lane 0 always)
            const int nlo = lo + (c > 0 ? c - 1 : 0) * step;
            hi = lo + c * step < hi ? lo + c * step : hi;
            lo = nlo;
        }
        plo = lo;
        probe = load_probe(plo);
    }

    for (int64_t b = b0; b < b1; ++b) {
        const int64_t t0 = b * kFluxT, t = t0 + lane;
        const bool live = t < n_frames;
        __syncthreads();                                         // (one wave: orders the LDS accesses of two blocks)
        int64_t s0, n;
        int lead;
        bool interior;
        span_of(b, s0, n, lead, interior);
        if (FLAT) {
#pragma unroll
            for (int u = 0; u < kFluxV4; ++u)
                reinterpret_cast<v4f *>(tile)[lane + 64 * u] = regs[u];
            if (!interior) {
                __syncthreads();
                const int pt = (int)((lead + n) >> 2);           // the piece that holds the span's last floats
                if (lane < 4 && lane >= lead && lane - lead < n)
                    tile[lane] = edge_head;
                if (lane < 4 && ((lead + n) & 3) != 0 && lane < (int)((lead + n) & 3) && 4 * pt + lane >= lead)
                    tile[4 * pt + lane] = edge_tail;
            }
        }
        starts[lane] = 0;
        __syncthreads();
        if (FLAT && b + 1 < b1)
            fetch_flat(b + 1);                                   // in flight while this block is walked

        // ---- which lanes start a stream, and which one ----
        bool first = false;
        const float *p0 = nullptr;
        if (offsets) {
            int lo = plo;
            int64_t o = probe;
            for (;;) {                                           // (one round, unless > 63 streams start inside the block)
                if (o >= t0 && o < t0 + kFluxT)
                    atomicMax(&starts[(int)(o - t0)], lo + lane + 1);
                const int c = __popcll(__ballot(o <= t0 + kFluxT));   // the next block's cursor: offsets ascend
                if (c < 64) {
                    plo = lo + (c > 0 ? c - 1 : 0);
                    break;
                }
                lo += 63;                                        // (lane 63's stream becomes lane 0's: it is the cursor so far)
                o = load_probe(lo);
            }
            probe = load_probe(plo);                             // (for the next block: in flight during the walk)
            __syncthreads();
            const int sf = starts[lane];
            first = sf != 0 || t == 0;
            p0 = (sf != 0 && prev0) ? prev0 + (int64_t)(sf - 1) * n_bins : nullptr;
        } else {
            first = t == 0;
            p0 = first ? prev0 : nullptr;
        }
        // (the select-per-bin form only where a starting frame brings a previous row of its own)
        const bool any_first = __ballot(first && live && p0 != nullptr) != 0;
        const bool zero_prev = first && p0 == nullptr;

        float fl = 0.0f, en = 0.0f;
        auto walk = [&](const float *cur, const float *prv, int kb, int k0, int k1) {
            if (any_first)
                flux_walk<true>(cur, prv, first, p0, wsh, kb, k0, k1, n_bins, fl, en);
            else
                flux_walk<false>(cur, zero_prev ? zrow : prv, first, p0, wsh, kb, k0, k1, n_bins, fl, en);
        };
        if (FLAT) {
            // float j of the span sits at tile[lead + j]; row r of the block (r = 0: the frame before it) starts n_bins
            // floats further on - except in block 0, whose span has no such row
            const float *row0 = tile + lead - (t0 > 0 ? 0 : n_bins);
            walk(row0 + (lane + 1) * n_bins, row0 + lane * n_bins, 0, 0, n_bins);
        } else {
            for (int kb = 0; kb == 0 || kb + 1 < n_bins; kb += kFluxTileBins) {
                // column c of the tile holds bin kb + c; the tile walks bins [kb + 1, kb + 113) (the first one bin 0 too)
                const int k0 = kb ? kb + 1 : 0, k1 = kb + 1 + kFluxTileBins < n_bins ? kb + 1 + kFluxTileBins : n_bins;
                __syncthreads();
                for (int r = 0; r <= kFluxT; ++r) {
                    const int64_t f = t0 - 1 + r;
#pragma unroll
                    for (int c = lane; c < kFluxTileBins + 2; c += 64) {
                        const int k = kb + c;
                        tile[r * kFluxTileStride + c] = (f >= 0 && f < n_frames && k < n_bins) ? mags[f * n_bins + k] : 0.0f;
                    }
                }
                __syncthreads();
                walk(tile + (lane + 1) * kFluxTileStride, tile + lane * kFluxTileStride, kb, k0, k1);
            }
        }
        if (live) {
            flux[t] = fl;
            energy[t] = en;
        }
    }
}

// Rest of the onset detector's per-frame logic after the flux (src/analysis/onset.rs:298-371,
// 531-535; "next" row f-2): per-bin rise-once burst floor, silence gate, energy EMA, FluxTracker,
// combined trigger, energy_rising, velocity and the frames_since_onset gate; r4 / r5: what the decision takes from the
// MusicalTransport - tick guard, calibration window, calibration timeout - as a per-frame mask, the calibration mode as an
// argument, calibration_done in the state (onset.rs:361-371, 393-440).  A recurrence over frames: ONE wave walks the
// frames of a stream in order, lanes = bins (floors in LDS), the burst count is a ballot + popcount
// and the largest excess a wave max (both order-independent), the scalar recurrences run
// redundantly in every lane in the reference's operation order -> bit-identical to the CPU
// restatement.  Magnitude rows are prefetched a block of frames ahead into registers and handed
// over through LDS; per-frame scalars in and out travel as one lane per frame of the block.
constexpr int kGatePrefetchRegs = 65;               // a block of frames holds <= 64 * 65 floats
// wave-wide max without LDS traffic: DPP row rotations inside the 16-lane rows, then
// v_permlane16/32_swap across rows (a single wave has nothing to hide a ds_bpermute behind)
__device__ __forceinline__ float gate_wave_max(float v)
{
    auto dpp = [](float x, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), decltype(ctrl)::value,
                                                          0xF, 0xF, false));
    };
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x121>()));   // row_ror:1
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x122>()));   // row_ror:2
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x124>()));   // row_ror:4
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x128>()));   // row_ror:8
    auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
    auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
}
// BPL = bins per lane (ceil(n_bins / 64)): the floors live in registers, bin k = 64 j + lane.
template <int BPL>
__global__ __launch_bounds__(64) void onset_gates_kernel(
    const float *__restrict__ mags, const float *__restrict__ flux, const float *__restrict__ energy,
    const float *__restrict__ global_floor, float *__restrict__ state, int initialized, int64_t n_frames,
    int n_bins, int frames_per_block, float *__restrict__ out_flux, float *__restrict__ out_ema,
    float *__restrict__ out_excess, int32_t *__restrict__ out_count, uint8_t *__restrict__ out_flags,
    float *__restrict__ out_velocity, const int64_t *__restrict__ offsets, const int32_t *__restrict__ init_flags,
    const uint8_t *__restrict__ suppress, int calibrated)
{
#pragma clang fp contract(off)
    extern __shared__ float gate_lds[];
    const int lane = threadIdx.x;
    if (offsets) {
        // batch: workgroup s walks stream s = frames [offsets[s], offsets[s+1]) of the concatenated arrays
        const int64_t f0 = offsets[blockIdx.x];
        n_frames = offsets[blockIdx.x + 1] - f0;
        mags += f0 * n_bins;
        flux += f0;
        energy += f0;
        global_floor += f0;
        out_flux += f0;
        out_ema += f0;
        out_excess += f0;
        out_count += f0;
        out_flags += f0;
        out_velocity += f0;
        if (suppress)
            suppress += f0;
        state += (int64_t)blockIdx.x * (n_bins + 4);
        initialized = init_flags ? init_flags[blockIdx.x] : 0;
    }
    if (n_frames <= 0)
        return;                                      // (nothing seen: the state stays as it was, and uninitialised)
    float *blk = gate_lds;                           // frames_per_block rows of magnitudes (+ one pad row)
    const float ENERGY_EMA_RISE = 0.84f, ENERGY_EMA_DECAY = 0.95f;
    const float BIN_BURST_RATIO = 2.5f, FLOOR_OVERCOMPENSATE = 1.3f, FLOOR_RISE = 0.1f, FLOOR_DECAY = 0.04f;
    const float multiplier = 1.5f, rise_memory = 0.84f, decay_memory = 0.89f;
    float fl[BPL];
    bool valid[BPL];
#pragma unroll
    for (int j = 0; j < BPL; ++j) {
        valid[j] = 64 * j + lane < n_bins;
        fl[j] = valid[j] ? state[64 * j + lane] : 1.0f;
    }
    float ema = 0.0f, thr = 0.0f;
    int since = 4, cal_done = 0;
    bool init = initialized != 0;
    if (init) {
        ema = state[n_bins];
        thr = state[n_bins + 1];
        since = __float_as_int(state[n_bins + 2]);
        cal_done = __float_as_int(state[n_bins + 3]);
    }
    if (calibrated != 2)
        cal_done = calibrated == 1;
    const int block_floats = frames_per_block * n_bins;
    const int64_t n_blocks = (n_frames + frames_per_block - 1) / frames_per_block;
    float nxt[kGatePrefetchRegs];
    auto fetch = [&](int64_t b) {
        const int64_t base = b * block_floats;
        const int64_t left = n_frames * n_bins - base;
#pragma unroll
        for (int i = 0; i < kGatePrefetchRegs; ++i) {
            const int idx = i * 64 + lane;
            nxt[i] = (idx < block_floats && idx < left) ? mags[base + idx] : 0.0f;
        }
    };
    fetch(0);
    for (int64_t b = 0; b < n_blocks; ++b) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kGatePrefetchRegs; ++i) {
            const int idx = i * 64 + lane;
            if (idx < block_floats)
                blk[idx] = nxt[i];
        }
        __syncthreads();
        if (b + 1 < n_blocks)
            fetch(b + 1);                            // in flight while this block is walked
        const int64_t t0 = b * frames_per_block;
        const int nfb = (int)((n_frames - t0) < frames_per_block ? (n_frames - t0) : frames_per_block);
        const bool mine = lane < nfb;
        const float in_flux = mine ? flux[t0 + lane] : 0.0f, in_energy = mine ? energy[t0 + lane] : 0.0f;
        const float in_gf = mine ? global_floor[t0 + lane] : 0.0f;
        const int in_sup = (mine && suppress) ? (int)suppress[t0 + lane] : (calibrated == 2 ? 2 : 0);
        float o_flux = 0.0f, o_ema = 0.0f, o_exc = 0.0f, o_vel = 0.0f;
        int o_count = 0, o_flags = 0;
        // this frame's magnitudes; the next frame's are read from LDS one iteration ahead
        float mcur[BPL], mnext[BPL];
#pragma unroll
        for (int j = 0; j < BPL; ++j)
            mcur[j] = valid[j] ? blk[64 * j + lane] : 0.0f;
        for (int i = 0; i < nfb; ++i) {
            const float *mn = blk + (i + 1) * n_bins;            // row nfb is the pad row
#pragma unroll
            for (int j = 0; j < BPL; ++j)
                mnext[j] = valid[j] ? mn[64 * j + lane] : 0.0f;
            float current_flux = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(in_flux), i));
            const float frame_energy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(in_energy), i));
            const float gf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(in_gf), i));
            const float floor_eps = fmaxf(gf, 0.01f);
            if (!init) {
#pragma unroll
                for (int j = 0; j < BPL; ++j)
                    fl[j] = fmaxf(mcur[j], gf);
                init = true;
            }
            float mx = 0.0f;
            int count = 0;
#pragma unroll
            for (int j = 0; j < BPL; ++j) {
                const float mag = mcur[j], f = fl[j];
                const float r = mag / fmaxf(f, floor_eps);
                const bool burst = valid[j] && r > BIN_BURST_RATIO;
                float nf;
                if (burst)
                    nf = mag * FLOOR_OVERCOMPENSATE;
                else if (mag > f)
                    nf = f + FLOOR_RISE * (mag - f);
                else
                    nf = f + FLOOR_DECAY * (mag - f);
                fl[j] = nf;
                count += __popcll(__ballot(burst));
                if (valid[j] && r > mx)
                    mx = r;
            }
            mx = gate_wave_max(mx);
            if (count < 2)
                current_flux = 0.0f;
            const float ema_memory = frame_energy > ema ? ENERGY_EMA_RISE : ENERGY_EMA_DECAY;
            ema = ema * ema_memory + frame_energy * (1.0f - ema_memory);
            const float memory = current_flux > thr ? rise_memory : decay_memory;
            const bool is_onset = current_flux > thr;
            thr = thr * memory + current_flux * (1.0f - memory);
            if (thr < 0.9f)
                thr = 0.9f;
            const bool flux_onset = is_onset && current_flux > thr * multiplier;
            const bool burst_onset = mx > 3.0f && count >= 3;
            const bool detected = flux_onset && burst_onset;
            const bool rising = frame_energy > ema * 1.5f;
            // onset.rs:361-371, 393-440: the transport's share of the decision, supplied by the caller as a per-frame
            // mask (bit 0 tick guard, bit 1 calibration click played and the residual plausible, bit 2 calibration timed
            // out) and the calibration mode; a gated onset fires once the calibration is done, or completes it (mode 2)
            const int mk = __builtin_amdgcn_readlane(in_sup, i);
            if (calibrated == 2 && (mk & 4))
                cal_done = 1;
            const bool tick = detected && (mk & 1) != 0;
            const bool gated = detected && !tick && rising && since >= 3;
            const bool fired = gated && cal_done != 0;
            const bool consumed = gated && cal_done == 0 && calibrated == 2 && (mk & 2) != 0;
            if (consumed)
                cal_done = 1;
            float vel = fmaxf(current_flux, mx * 5.0f) / 50.0f;
            vel = vel < 0.0f ? 0.0f : (vel > 1.0f ? 1.0f : vel);
            if (fired || consumed || (detected && since < 3))
                since = 0;
            else if (since < 0x7fffffff)
                since += 1;
            if (lane == i) {
                o_flux = current_flux;
                o_ema = ema;
                o_exc = mx;
                o_vel = vel;
                o_count = count;
                o_flags = (int)flux_onset | ((int)burst_onset << 1) | ((int)detected << 2) | ((int)rising << 3) |
                          ((int)fired << 4) | ((int)tick << 5) | ((int)gated << 6) | ((int)consumed << 7);
            }
#pragma unroll
            for (int j = 0; j < BPL; ++j)
                mcur[j] = mnext[j];
        }
        if (mine) {
            out_flux[t0 + lane] = o_flux;
            out_ema[t0 + lane] = o_ema;
            out_excess[t0 + lane] = o_exc;
            out_count[t0 + lane] = o_count;
            out_flags[t0 + lane] = (uint8_t)o_flags;
            out_velocity[t0 + lane] = o_vel;
        }
    }
#pragma unroll
    for (int j = 0; j < BPL; ++j)
        if (valid[j])
            state[64 * j + lane] = fl[j];
    if (lane == 0) {
        state[n_bins] = ema;
        state[n_bins + 1] = thr;
        state[n_bins + 2] = __int_as_float(since);
        state[n_bins + 3] = __int_as_float(cal_done);
    }
}

// Tuner multi-pitch extraction (src/audio_io/stft.rs:443-620; rest of "next" row f-1 up to the
// PitchTracker): per frame, peaks above the effective noise floor, parabolic log-magnitude
// interpolation, harmonic-sum score over <= 14 harmonics, 50 % cutoff, harmonic-ghost suppression,
// sort by score, 2-bin dedupe, <= 8 notes in [min_freq, max_freq].  Frames are independent: one
// wave per frame, rows in LDS; peaks / candidates are compacted in bin order with ballots, every
// peak is scored by one lane with the reference's sequential harmonic walk, the O(C^2) passes
// (suppression, ranking) run one candidate per lane.  f32 in the reference's operation order;
// ln / log2 are ocml's logf / log2f (<= 1 ulp, not libm's bits), so scores agree to ~1e-6
// relative and a decision can flip only on an exact tie.  Equal scores keep bin order (the
// reference's sort_unstable_by leaves that unspecified).
constexpr int kMaxNotes = 8, kMaxHarmonics = 14;
// PREF (n_bins <= 64 * 17, the tuner's 1025 included): the frame's two rows come in as 2 x 17 loads issued back to back into
// registers - the rows of the block's NEXT frame while the current one is scored - instead of a loop of dependent
// load -> LDS store pairs (17 serialised global round trips per frame: two thirds of the kernel's wave cycles were that wait).
// onset_gates_split_kernel (r4): the same gate chain for ONE stream per workgroup with the recurrences taken apart, so
// that a single live stream (the reference's shape: onset.rs runs one) no longer walks its frames in one wave at 0.7 us
// each.  The per-bin floor recurrence (onset.rs:311-332) is independent per bin and feeds the frame only through the
// burst count and the largest excess ratio; the scalar recurrences (energy EMA :345-350, FluxTracker :66-83, frame gate
// :531-535) need only those two numbers.  Five waves, three roles, a block of T frames apart each:
//   A (3 waves) one lane per bin (<= 192 bins): the floor recurrence of block b, the bin's T magnitudes in registers
//               (the next block's prefetched from global memory), the ratio of every (frame, bin) into LDS
//   R (1 wave)  block b - 1: lanes = (frame, half of the bins) walk their column of ratios: count of ratios above
//               BIN_BURST_RATIO and their maximum (order-free, so bit-identical to the sequential loop); 32 reads in flight
//   B (1 wave)  block b - 2: everything per frame that is no recurrence in the lanes (silence gate, burst trigger,
//               velocity), then the recurrences: the energy EMA and the flux threshold are two INDEPENDENT chains of the
//               same shape - s = s * (x > s ? k_rise : k_decay) + (x > s ? a_rise : a_decay) - so they run side by side in
//               the two halves of the wave, one frame per step, operands from LDS rows, in VGPRs throughout (eight VALU
//               instructions per frame; as wave-uniform scalars the compiler moved them through SGPRs and a branch: 404
//               cycles per frame, tools/ubench_gate_chain.hip); the frames-since counter only moves at detected frames
//               and is stepped on the scalar unit over the set bits of the block's detection mask.
// A lone wave issues one instruction every ~4.4 cycles, so the roles are placed by SIMD: the workgroup's five waves land
// on four SIMDs, the two that share one become R and B (their instruction counts per frame add up to about A's), the
// three with a SIMD to themselves run A.  One workgroup barrier per block.  Every f32 operation of a bin / of the chains
// is the reference's, in its order (fp contract off): the outputs are bit-identical to onset_gates_kernel and to the C
// restatement.
// r5: T = frames per block.  32 for a few long streams (the lone waves' prefetch depth: 194 registers, two waves per SIMD,
// i.e. ONE workgroup per CU); 16 for batches: half the registers and half the LDS, several workgroups per CU, and a
// stream's pipeline fills and drains in two steps of 16 frames instead of 32.
__device__ __forceinline__ int64_t uniform_i64(int64_t v)     // a wave-uniform value into scalar registers
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
#ifndef AA_GATES_T16_WAVES
#define AA_GATES_T16_WAVES 5
#endif
#ifndef AA_GATES_T32_WAVES
#define AA_GATES_T32_WAVES 1
#endif
constexpr int gate_ops_dwords(int T) { return T * 2 * 4 + T * 2 * 2 + 8; }   // operand rows, result rows, role table
template <int T>
__global__ __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(T == 16 ? AA_GATES_T16_WAVES : AA_GATES_T32_WAVES)))
void onset_gates_split_kernel(
    const float *__restrict__ mags, const float *__restrict__ flux, const float *__restrict__ energy,
    const float *__restrict__ global_floor, float *__restrict__ state, int initialized, int64_t n_frames,
    int n_bins, float *__restrict__ out_flux, float *__restrict__ out_ema,
    float *__restrict__ out_excess, int32_t *__restrict__ out_count, uint8_t *__restrict__ out_flags,
    float *__restrict__ out_velocity, const int64_t *__restrict__ offsets, const int32_t *__restrict__ init_flags,
    const uint8_t *__restrict__ suppress, int calibrated)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float gate_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (offsets) {
        // (made scalar by hand: the two offsets arrive through vector loads, and every pointer advanced by them would
        // count as per-lane - each row load with a 64-bit address pair of its own)
        const int64_t f0 = uniform_i64(offsets[blockIdx.x]);
        n_frames = uniform_i64(offsets[blockIdx.x + 1]) - f0;
        mags += f0 * n_bins;
        flux += f0;
        energy += f0;
        global_floor += f0;
        out_flux += f0;
        out_ema += f0;
        out_excess += f0;
        out_count += f0;
        out_flags += f0;
        out_velocity += f0;
        if (suppress)
            suppress += f0;
        state += (int64_t)blockIdx.x * (n_bins + 4);
        initialized = init_flags ? init_flags[blockIdx.x] : 0;
    }
    if (n_frames <= 0)
        return;                                      // (nothing seen: the state stays as it was, and uninitialised)
    // rows of the ratio image: n_bins columns + a spare one the lanes without a bin write to (no per-frame lane
    // masks), an odd stride (the column walk of the R wave is bank-conflict free)
    const int stride = (n_bins + 1) | 1;
    float *ops = gate_lds;                           // [T][2][4]: (x, a_rise, a_decay, clamp) of the EMA / of the threshold
    float *res = ops + T * 2 * 4;               // [T][2][2]: (state behind the frame, x > state in front of it)
    int *roles = reinterpret_cast<int *>(res + T * 2 * 2);             // [8]
    float *ratio = gate_lds + gate_ops_dwords(T);        // [2][T][stride]
    int *cnts = reinterpret_cast<int *>(ratio + 2 * T * stride);      // [2][T]
    float *maxs = reinterpret_cast<float *>(cnts + 2 * T);            // [2][T]
    const float ENERGY_EMA_RISE = 0.84f, ENERGY_EMA_DECAY = 0.95f;
    const float BIN_BURST_RATIO = 2.5f, FLOOR_OVERCOMPENSATE = 1.3f, FLOOR_RISE = 0.1f, FLOOR_DECAY = 0.04f;
    const float multiplier = 1.5f, rise_memory = 0.84f, decay_memory = 0.89f;
    const int64_t n_blocks = (n_frames + T - 1) / T;

    // ---- roles by SIMD (HW_ID bits 5:4): 0..2 = A for bins 64 role .. 64 role + 63, 3 = R, 4 = B ----
    if (lane == 0)
        roles[wave] = (int)(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3);
    __syncthreads();
    int role = wave;
    {
        int simd[5], n_on[4] = {0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < 5; ++w) {
            simd[w] = __builtin_amdgcn_readfirstlane(roles[w]);
            n_on[simd[w]] += 1;
        }
        int shared = -1;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (n_on[q] == 2 && n_on[(q + 1) & 3] == 1 && n_on[(q + 2) & 3] == 1 && n_on[(q + 3) & 3] == 1)
                shared = q;
        if (shared >= 0) {                           // (any other placement: roles by wave index, correct all the same)
            int next_a = 0, next_rb = 3;
#pragma unroll
            for (int w = 0; w < 5; ++w) {
                const int r = simd[w] == shared ? next_rb++ : next_a++;
                role = w == wave ? r : role;
            }
        }
    }
    role = __builtin_amdgcn_readfirstlane(role);
    __syncthreads();

    // ---- state of the three roles ----
    const int bin = 64 * role + lane;
    const bool valid = role < 3 && bin < n_bins;
    const int col = valid ? bin : n_bins;            // (the spare column)
    float fl = valid ? state[bin] : 1.0f;
    float cur[T], nxt[T];
    float cur_gf = 0.0f, nxt_gf = 0.0f;              // lane i: global_floor of frame i of the block (read by readlane)
    auto fetch = [&](int64_t b, float (&dst)[T], float &gfv) {
        const int64_t t0 = b * T;
        if (t0 + T <= n_frames) {               // (a full block: no per-frame bounds)
            // a scalar row pointer + ONE 32-bit lane offset: as per-lane 64-bit addresses the T loads held 2 T registers and
            // took two vector instructions each to form
            const char *row = reinterpret_cast<const char *>(mags + t0 * n_bins);
            const unsigned voff = valid ? 4u * (unsigned)bin : 0u;      // (a BYTE offset: the form the scalar-base load takes)
#pragma unroll
            for (int i = 0; i < T; ++i) {
                unsigned vo = voff;
                asm volatile("" : "+v"(vo));     // (opaque per row: or base + offset is formed once, as a per-lane pointer)
                dst[i] = *reinterpret_cast<const float *>(row + vo);
                row += 4 * n_bins;
            }
        } else {
            // the stream's last block: the rows that exist (wave-uniform count), the others read the last one again
            const int nrow = (int)(n_frames - t0);
            const char *row = reinterpret_cast<const char *>(mags + t0 * n_bins);
            const unsigned voff = valid ? 4u * (unsigned)bin : 0u;
#pragma unroll
            for (int i = 0; i < T; ++i) {
                unsigned vo = voff;
                asm volatile("" : "+v"(vo));
                dst[i] = *reinterpret_cast<const float *>(row + vo);
                if (i + 1 < nrow)
                    row += 4 * n_bins;
            }
        }
        gfv = (lane < T && t0 + lane < n_frames) ? global_floor[t0 + lane] : 0.0f;
    };
    // B: the per-frame inputs of a block, fetched a step ahead as well
    float b_flux = 0.0f, b_energy = 0.0f, n_flux = 0.0f, n_energy = 0.0f;
    int b_sup = 0, n_sup = 0;
    auto fetch_b = [&](int64_t b, float &vf, float &ve, int &vs) {
        const int64_t t = b * T + lane;
        const bool in = lane < T && t < n_frames;
        vf = in ? flux[t] : 0.0f;
        ve = in ? energy[t] : 0.0f;
        vs = (in && suppress) ? (int)suppress[t] : (calibrated == 2 ? 2 : 0);
    };
    // B: lanes 0-31 carry the energy EMA, lanes 32-63 the flux threshold (the same instruction stream for both)
    const int grp = lane >> 5;
    float chain_s = 0.0f;
    const float k_rise = grp ? rise_memory : ENERGY_EMA_RISE, k_decay = grp ? decay_memory : ENERGY_EMA_DECAY;
    int since = 4, cal_done = 0;
    if (initialized) {
        chain_s = grp ? state[n_bins + 1] : state[n_bins];
        since = __builtin_amdgcn_readfirstlane(__float_as_int(state[n_bins + 2]));
        cal_done = __builtin_amdgcn_readfirstlane(__float_as_int(state[n_bins + 3]));
    }
    if (calibrated != 2)
        cal_done = calibrated == 1;

    // one frame of one bin (onset.rs:311-332), branch-free: the three candidate floors, two selects
    auto bin_step = [&](float mag, float gf) {
        const float floor_eps = fmaxf(gf, 0.01f);
        const float r = mag / fmaxf(fl, floor_eps);
        const float d = mag - fl;
        const float burst_f = mag * FLOOR_OVERCOMPENSATE, rise_f = fl + FLOOR_RISE * d, decay_f = fl + FLOOR_DECAY * d;
        const float slow = mag > fl ? rise_f : decay_f;
        fl = r > BIN_BURST_RATIO ? burst_f : slow;
        return r;
    };

    // r5: a loop per role (the waves meet at one barrier per step all the same): with the roles as branches of ONE loop the
    // registers every role carries from step to step - A's two blocks of magnitudes above all - were live in all of them
    if (role < 3) {
        fetch(0, cur, cur_gf);
        if (!initialized)                            // onset.rs:304-309: the floors start at the first frame's magnitudes
            fl = fmaxf(cur[0], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur_gf), 0)));
        for (int64_t step = 0; step < n_blocks + 2; ++step) {
            // ---- per-bin floor recurrence of block `step` ----
            if (step < n_blocks) {
                if (step + 1 < n_blocks)
                    fetch(step + 1, nxt, nxt_gf);    // in flight while this block is walked
                const int64_t t0 = step * T;
                const int nfb = (int)((n_frames - t0) < T ? (n_frames - t0) : T);
                float *rb = ratio + (step & 1) * T * stride + col;
                if (nfb == T) {
#pragma unroll
                    for (int i = 0; i < T; ++i)
                        rb[i * stride] = bin_step(cur[i], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur_gf), i)));
                } else {
#pragma unroll
                    for (int i = 0; i < T; ++i)          // (the stream's last block: a wave-uniform branch per frame)
                        if (i < nfb)
                            rb[i * stride] = bin_step(cur[i], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur_gf), i)));
                }
#pragma unroll
                for (int i = 0; i < T; ++i)
                    cur[i] = nxt[i];
                cur_gf = nxt_gf;
            }
            __syncthreads();
        }
        if (valid)
            state[bin] = fl;
    } else if (role == 3) {
        for (int64_t step = 0; step < n_blocks + 2; ++step) {
            // ---- burst count and largest excess of every frame of block `step - 1` ----
            if (step >= 1 && step <= n_blocks) {
                const int pb = (int)((step - 1) & 1);
                constexpr int PARTS = 64 / T;            // lanes = (frame, part of the bins)
                const int i = lane & (T - 1), part = lane / T;
                const int nbh = (n_bins + PARTS - 1) / PARTS;
                const int k0 = part * nbh < n_bins ? part * nbh : n_bins, k1 = k0 + nbh < n_bins ? k0 + nbh : n_bins;
                const float *colp = ratio + pb * T * stride + i * stride;
                int count = 0;
                float mx = 0.0f;
                int k = k0;
                // thirty-two reads in flight: a lone wave sees the whole LDS round trip on every wait
                for (; k + T <= k1; k += T) {
                    float r[T];
#pragma unroll
                    for (int j = 0; j < T; ++j)
                        r[j] = colp[k + j];
#pragma unroll
                    for (int j = 0; j < T; ++j) {
                        count += r[j] > BIN_BURST_RATIO ? 1 : 0;
                        mx = r[j] > mx ? r[j] : mx;
                    }
                }
                for (; k + 8 <= k1; k += 8) {
                    float r[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        r[j] = colp[k + j];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        count += r[j] > BIN_BURST_RATIO ? 1 : 0;
                        mx = r[j] > mx ? r[j] : mx;
                    }
                }
                for (; k < k1; ++k) {
                    const float r = colp[k];
                    count += r > BIN_BURST_RATIO ? 1 : 0;
                    mx = r > mx ? r : mx;
                }
#pragma unroll
                for (int off = 32; off >= T; off >>= 1) {
                    count += __shfl_xor(count, off);
                    const float other = __shfl_xor(mx, off);
                    mx = other > mx ? other : mx;
                }
                if (part == 0) {
                    cnts[pb * T + i] = count;
                    maxs[pb * T + i] = mx;
                }
            }
            __syncthreads();
        }
    } else {
        fetch_b(0, b_flux, b_energy, b_sup);
        for (int64_t step = 0; step < n_blocks + 2; ++step) {
            // ---- the frame's gates and the recurrences of block `step - 2` ----
            if (step >= 2) {
                const int64_t b = step - 2, t0 = b * T;
                const int pb = (int)(b & 1);
                const int nfb = (int)((n_frames - t0) < T ? (n_frames - t0) : T);
                const bool mine = lane < nfb;
                const int count = mine ? cnts[pb * T + lane] : 0;
                const float mx = mine ? maxs[pb * T + lane] : 0.0f;
                float v_flux = b_flux;
                const float v_energy = b_energy;
                const int v_sup = b_sup;
                if (b + 1 < n_blocks)
                    fetch_b(b + 1, n_flux, n_energy, n_sup);     // in flight while this block's chains run
                // per frame, no recurrence (lane = frame): silence gate, burst trigger, velocity (onset.rs:336-338, 358,
                // 381-382) and the products the two memories may pick from (the reference's own operands: x * (1 - memory))
                if (count < 2)
                    v_flux = 0.0f;
                const bool bo = mx > 3.0f && count >= 3;
                float vel = fmaxf(v_flux, mx * 5.0f) / 50.0f;
                vel = vel < 0.0f ? 0.0f : (vel > 1.0f ? 1.0f : vel);
                if (lane < T) {
                    v4f *row = reinterpret_cast<v4f *>(ops + 8 * lane);
                    row[0] = v4f{v_energy, v_energy * (1.0f - ENERGY_EMA_RISE), v_energy * (1.0f - ENERGY_EMA_DECAY), -INFINITY};
                    row[1] = v4f{v_flux, v_flux * (1.0f - rise_memory), v_flux * (1.0f - decay_memory), 0.9f};
                }
                __builtin_amdgcn_wave_barrier();
                // the two recurrences, one frame per step.  The row addresses go through an opaque VGPR: as wave-uniform
                // values the operands would be loaded into SGPRs and every compare / select of the chain would take the
                // VALU -> SGPR -> VALU detour again.
                unsigned row_off = 16u * (unsigned)grp, res_off = 8u * (unsigned)grp;
                asm volatile("" : "+v"(row_off), "+v"(res_off));
                const char *ops_b = reinterpret_cast<const char *>(ops) + row_off;
                char *res_b = reinterpret_cast<char *>(res) + res_off;
                auto chain = [&](const v4f o, int i) {                  // o: x, a_rise, a_decay, clamp
                    const bool up = o.x > chain_s;
                    chain_s = chain_s * (up ? k_rise : k_decay) + (up ? o.y : o.z);
                    chain_s = chain_s < o.w ? o.w : chain_s;          // (the threshold's 0.9 floor; -inf for the EMA)
                    *reinterpret_cast<v2f *>(res_b + 16 * i) = v2f{chain_s, up ? 1.0f : 0.0f};
                };
                if (nfb == T) {
                    // operand rows T / 4 frames at a time, the next ones in flight while these are stepped (the reads
                    // stand in front of the result writes in program order: the compiler need not reorder LDS accesses
                    // it cannot tell apart, and the wave does not sit out an LDS round trip per frame)
                    constexpr int G = T / 4;
                    v4f o[2][G];
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        o[0][j] = *reinterpret_cast<const v4f *>(ops_b + 32 * j);
#pragma unroll
                    for (int k = 0; k < T / G; ++k) {
                        if (k + 1 < T / G) {
#pragma unroll
                            for (int j = 0; j < G; ++j)
                                o[(k + 1) & 1][j] = *reinterpret_cast<const v4f *>(ops_b + 32 * (G * (k + 1) + j));
                        }
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            chain(o[k & 1][j], G * k + j);
                    }
                } else {
                    for (int i = 0; i < nfb; ++i)
                        chain(*reinterpret_cast<const v4f *>(ops_b + 32 * i), i);
                }
                __builtin_amdgcn_wave_barrier();
                // back to lane = frame: the frame's flags from the two states behind it
                const int fr = lane < T ? lane : 0;
                const v2f r_ema = *reinterpret_cast<const v2f *>(res + 4 * fr), r_thr = *reinterpret_cast<const v2f *>(res + 4 * fr + 2);
                const float ema_i = r_ema.x, thr_i = r_thr.x;
                const bool is_onset = r_thr.y != 0.0f;
                const bool flux_onset = mine && is_onset && v_flux > thr_i * multiplier;
                const bool detected = flux_onset && bo;
                const bool rising = v_energy > ema_i * 1.5f;
                // the transport's share (onset.rs:361-371, 393-440), a per-frame mask: bit 0 tick guard, bit 1 calibration
                // click played and the residual plausible, bit 2 calibration timed out
                const bool tick = detected && (v_sup & 1) != 0;
                const bool armed = detected && !tick && rising;          // gated if >= 3 frames since the last onset
                // frames since the last onset: + 1 per frame, except at a detected frame (0 when it fires, completes the
                // calibration or comes within 3 frames of the last one) - stepped over the set bits of the block's
                // detection mask; the calibration (mode 2) is done from the first timed-out frame on, or from the
                // first gated frame whose residual is plausible, which it consumes
                const uint64_t det_mask = __ballot(detected), arm_mask = __ballot(armed);
                const uint64_t open_mask = __ballot(mine && calibrated == 2 && (v_sup & 2) != 0);
                const uint64_t tmo_mask = __ballot(mine && calibrated == 2 && (v_sup & 4) != 0);
                const int tmo_at = tmo_mask ? __builtin_ctzll(tmo_mask) : 64;
                uint64_t fired_mask = 0, gated_mask = 0, used_mask = 0, todo = det_mask;
                int at = 0;                                             // frames of the block already counted
                auto advance = [&](int n) { since = since > 0x7fffffff - n ? 0x7fffffff : since + n; };
                while (todo) {
                    const int i = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    advance(i - at);
                    const bool g = ((arm_mask >> i) & 1) != 0 && since >= 3;
                    const bool done = cal_done != 0 || i >= tmo_at;
                    const bool f = g && done, u = g && !done && ((open_mask >> i) & 1) != 0;
                    gated_mask |= (uint64_t)g << i;
                    fired_mask |= (uint64_t)f << i;
                    used_mask |= (uint64_t)u << i;
                    if (u)
                        cal_done = 1;
                    if (f || u || since < 3)
                        since = 0;
                    else
                        advance(1);
                    at = i + 1;
                }
                advance(nfb - at);
                if (tmo_at < 64)
                    cal_done = 1;
                const bool fired = ((fired_mask >> lane) & 1) != 0, gated = ((gated_mask >> lane) & 1) != 0;
                const bool used = ((used_mask >> lane) & 1) != 0;
                const int fl_bits = (int)flux_onset | ((int)bo << 1) | ((int)detected << 2) | ((int)rising << 3) |
                                    ((int)fired << 4) | ((int)tick << 5) | ((int)gated << 6) | ((int)used << 7);
                if (mine) {
                    out_flux[t0 + lane] = v_flux;
                    out_ema[t0 + lane] = ema_i;
                    out_excess[t0 + lane] = mx;
                    out_count[t0 + lane] = count;
                    out_flags[t0 + lane] = (uint8_t)fl_bits;
                    out_velocity[t0 + lane] = vel;
                }
                b_flux = n_flux;
                b_energy = n_energy;
                b_sup = n_sup;
            }
            __syncthreads();
        }
        if (lane == 0)
            state[n_bins] = chain_s;
        if (lane == 32)
            state[n_bins + 1] = chain_s;
        if (lane == 0) {
            state[n_bins + 2] = __int_as_float(since);
            state[n_bins + 3] = __int_as_float(cal_done);
        }
    }
}

// r5: the kernel of rounds 1-4 walked the 13 harmonics of every peak one peak per lane - 780 of its 1310 vector
// instructions per frame, executed for the ~10 lanes whose peak stands 5 x above its floor - and kept both rows, a byte per
// bin and six per-peak arrays of n_bins entries in LDS (25.7 KB: six waves per CU, half of their time waiting).  Now:
//   * the peak test runs on the rows as they arrive in registers (neighbours from the LDS copy of the magnitudes); all
//     peaks go into a BIT MASK (one ballot per 64 bins), only the STRONG ones (magnitude >= 5 x floor: the others score 0 and
//     can never pass the 50 % cutoff, stft.rs:489-491, 556-560) into the compacted list;
//   * the harmonic search is FLATTENED over (peak, harmonic) pairs, 64 pairs per pass: the window of harmonic n
//     (stft.rs:507-515: at most four bins) is four bits of the mask, its strongest peak found without a loop;
//   * the sequential part of the walk - a found harmonic may not be found again (search_start >= last + 1, :511) - only
//     matters where the windows of consecutive harmonics touch (fundamentals below ~70 Hz); every pair checks that
//     against the previous window's end, and a peak with no such contact sums its 13 results in order (bit-identical to
//     the loop); the others take the reference's loop, one peak per lane as before;
//   * ghost suppression runs over (candidate, candidate) pairs across the lanes; the 2-bin dedupe keeps the accepted
//     notes one per lane.
// 17.7 KB of LDS per frame in flight at the tuner's 1025 bins (nine waves per CU).
constexpr int kPitH = kMaxHarmonics - 1;            // harmonics n = 2 .. 14
#ifndef AA_PIT_GROUP
#define AA_PIT_GROUP 16
#endif
constexpr int kPitGroup = AA_PIT_GROUP;             // strong peaks per scoring pass
__device__ __forceinline__ float pit_wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
template <bool PREF>
__global__ __launch_bounds__(64) void tuner_pitches_kernel(
    const float *__restrict__ mags, const float *__restrict__ eff_floor, int64_t n_frames, int n_bins,
    float bin_width, float min_freq, float max_freq, int min_bin, int max_bin, int max_peaks,
    float *__restrict__ out_freq, float *__restrict__ out_score, int32_t *__restrict__ out_count)
{
#pragma clang fp contract(off)
    extern __shared__ float pit_lds[];
    const int lane = threadIdx.x;
    const int nbp = (n_bins + 3) & ~3, nw = (n_bins + 63) >> 6;
    const int half_size = n_bins;
    float *m = pit_lds;
    uint32_t *mask32 = reinterpret_cast<uint32_t *>(m + nbp);              // [2 nw + 2] peak bits, 64 bins per pair of words
    // strong peaks, bin order: the bin in 15 bits, bit 15 = "fundamental below 15 x its floor" (all the score needs of the
    // floor, stft.rs:546) - r5: 10 bytes per possible peak instead of 16, the kept notes on the dead magnitude row; the
    // kernel's time follows the waves a CU holds (9 at 17.7 KB: 191 us, 11: 175, 15: 120 with a narrower peak range)
    uint16_t *peak_bin = reinterpret_cast<uint16_t *>(mask32 + 2 * nw + 2);
    float *frac = reinterpret_cast<float *>(peak_bin + ((max_peaks + 1) & ~1)), *score = frac + max_peaks;
    uint32_t *hw = reinterpret_cast<uint32_t *>(score + max_peaks);         // [kPitGroup][13] harmonic results of a group
    float *hm = reinterpret_cast<float *>(hw + kPitGroup * kPitH);
    // later stages on dead areas (every stage ends in a workgroup barrier): the magnitudes are not read behind the scores
    uint16_t *cand = peak_bin;
    float *kfrac = 2 * max_peaks <= nbp ? m : hm + kPitGroup * kPitH, *kscore = kfrac + max_peaks;
    float *sfrac = frac, *sscore = score;
    const unsigned long long lt = (1ull << lane) - 1ull;
    constexpr int kIt = 17;
    float rm[kIt], rn[kIt];
    auto row_loads = [&](int64_t t) {
#pragma unroll
        for (int i = 0; i < kIt; ++i) {
            const int k = lane + 64 * i;
            if (k < n_bins) {
                rm[i] = mags[t * n_bins + k];
                if (64 * i + 63 > min_bin && 64 * i < max_bin)    // (wave uniform) the floor is read where peaks may lie
                    rn[i] = eff_floor[t * n_bins + k];
            }
        }
    };
    // the reference's loop for one peak (stft.rs:497-540), used where harmonic windows touch
    auto walk_slow = [&](int k, float frac_bin, float fund_mag, int &longest_run, int &total_harms) {
        float acc = fund_mag;
        int last = k, current_run = 0;
        longest_run = 0;
        total_harms = 0;
        for (int n = 2; n <= kMaxHarmonics; ++n) {
            const float expected_f = frac_bin * (float)n;
            if (expected_f >= (float)half_size)
                break;
            const float lo = floorf(expected_f - 1.0f), hi = ceilf(expected_f + 1.0f);
            int search_start = lo > 0.0f ? (int)lo : 0;
            if (search_start < last + 1)
                search_start = last + 1;
            int search_end = (int)hi;
            if (search_end > half_size - 1)
                search_end = half_size - 1;
            int best_hbin = 0;
            float best_mag = 0.0f;
            for (int h = search_start; h <= search_end; ++h)
                if (((mask32[h >> 5] >> (h & 31)) & 1u) && m[h] > best_mag) {
                    best_mag = m[h];
                    best_hbin = h;
                }
            if (best_hbin != 0) {
                acc += best_mag;
                last = best_hbin;
                current_run += 1;
                total_harms += 1;
            } else {
                if (current_run > longest_run)
                    longest_run = current_run;
                current_run = 0;
            }
        }
        if (current_run > longest_run)
            longest_run = current_run;
        return acc;
    };
    if ((int64_t)blockIdx.x < n_frames) {
        if (PREF)
            row_loads(blockIdx.x);
    }
    for (int64_t t = blockIdx.x; t < n_frames; t += gridDim.x) {
        __syncthreads();
        if (PREF) {
#pragma unroll
            for (int i = 0; i < kIt; ++i) {
                const int k = lane + 64 * i;
                if (k < n_bins)
                    m[k] = rm[i];
            }
        } else {
            for (int k = lane; k < n_bins; k += 64)
                m[k] = mags[t * n_bins + k];
        }
        if (lane < 2 * nw + 2)
            mask32[lane] = 0u;
        for (int q = 64 + lane; q < 2 * nw + 2; q += 64)
            mask32[q] = 0u;
        __syncthreads();
        // ---- peaks: all into the bit mask, the strong ones compacted in bin order ----
        int n_strong = 0;
        auto peak_row = [&](int i, float mk, float nfk) {
            const int k = lane + 64 * i;
            const bool in = k > min_bin && k < max_bin;
            const int kc = in ? k : 1;
            const bool pk = in && mk > nfk && mk >= m[kc - 1] && mk >= m[kc + 1];
            const unsigned long long mask = __ballot(pk);
            if (lane == 0) {
                mask32[2 * i] = (uint32_t)mask;
                mask32[2 * i + 1] = (uint32_t)(mask >> 32);
            }
            const bool strong = pk && !(mk < nfk * 5.0f);
            const unsigned long long smask = __ballot(strong);
            if (strong) {
                const int pos = n_strong + __popcll(smask & lt);
                peak_bin[pos] = (uint16_t)(k | (mk < 15.0f * nfk ? 0x8000 : 0));
            }
            n_strong += __popcll(smask);
        };
        if (PREF) {
#pragma unroll
            for (int i = 0; i < kIt; ++i)
                if (64 * i + 63 > min_bin && 64 * i < max_bin)        // (wave uniform)
                    peak_row(i, rm[i], rn[i]);
            if (t + gridDim.x < n_frames)
                row_loads(t + gridDim.x);                             // in flight while this frame is scored
        } else {
            for (int i = (min_bin + 1) >> 6; 64 * i < max_bin; ++i) {
                const int k = lane + 64 * i;
                const bool ok = k < n_bins;
                peak_row(i, ok ? m[k] : 0.0f, ok ? eff_floor[t * n_bins + k] : 0.0f);
            }
        }
        __syncthreads();
        int n_out = 0;
        float my_freq = 0.0f, my_score = 0.0f;             // lane i < 8 carries output slot i
        float mx = 0.0f;
        // ---- harmonic-sum score of the strong peaks, 64 at a time ----
        for (int g0 = 0; g0 < n_strong; g0 += kPitGroup) {
            const int gcount = n_strong - g0 < kPitGroup ? n_strong - g0 : kPitGroup;
            const bool mine = lane < gcount;
            int k = 2;
            float fund_mag = 0.0f, frac_bin = 0.0f;
            bool below15 = false;
            if (mine) {
                k = peak_bin[g0 + lane] & 0x7fff;
                below15 = (peak_bin[g0 + lane] & 0x8000) != 0;
                fund_mag = m[k];
                frac_bin = (float)k;
                if (k >= 1 && k + 1 < half_size) {
                    const float y_l = logf(m[k - 1]), y_c = logf(fund_mag), y_r = logf(m[k + 1]);
                    const float denom = y_l - 2.0f * y_c + y_r;
                    float delta = 0.0f;
                    if (!(fabsf(denom) < 1e-30f)) {
                        delta = 0.5f * (y_l - y_r) / denom;
                        delta = delta < -1.0f ? -1.0f : (delta > 1.0f ? 1.0f : delta);
                    }
                    frac_bin = (float)k + delta;
                }
                frac[g0 + lane] = frac_bin;
            }
            __syncthreads();
            // (peak, harmonic) pairs across the lanes: pair e = 13 * peak + (n - 2)
            const int n_items = gcount * kPitH;
            for (int e0 = 0; e0 < n_items; e0 += 64) {
                const int e = e0 + lane;
                if (e < n_items) {
                    const int sl = (e * 1261) >> 14;                  // e / 13 for e < 832
                    const int n = e - kPitH * sl + 2;
                    const float fb = frac[g0 + sl];
                    const float expected_f = fb * (float)n;
                    uint32_t word = 0u;
                    float best_mag = 0.0f;
                    if (!(expected_f >= (float)half_size)) {
                        const float lo = floorf(expected_f - 1.0f), hi = ceilf(expected_f + 1.0f);
                        int ss = lo > 0.0f ? (int)lo : 0;
                        int se = (int)hi;
                        if (se > half_size - 1)
                            se = half_size - 1;
                        // the end of what the walk can have taken before: the peak's own bin (n = 2), or the previous window
                        int pe;
                        if (n == 2) {
                            pe = peak_bin[g0 + sl] & 0x7fff;
                        } else {
                            pe = (int)ceilf(fb * (float)(n - 1) + 1.0f);
                            if (pe > half_size - 1)
                                pe = half_size - 1;
                        }
                        const bool noconf = pe + 1 <= ss;
                        int len = se - ss + 1;                        // 3 or 4 bins (fewer at the top of the spectrum)
                        len = len < 0 ? 0 : (len > 4 ? 4 : len);
                        const uint32_t w0 = mask32[ss >> 5], w1 = mask32[(ss >> 5) + 1];
                        const uint64_t w64 = (uint64_t)w0 | ((uint64_t)w1 << 32);
                        const uint32_t bits = (uint32_t)(w64 >> (ss & 31)) & ((1u << len) - 1u);
                        int best_hbin = 0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if ((bits >> u) & 1u) {
                                const float mg = m[ss + u];
                                if (mg > best_mag) {
                                    best_mag = mg;
                                    best_hbin = ss + u;
                                }
                            }
                        }
                        word = (uint32_t)best_hbin | ((uint32_t)noconf << 24) | (1u << 25);
                    }
                    hw[e] = word;
                    hm[e] = best_mag;
                }
            }
            __syncthreads();
            if (mine) {
                // the 13 results of this lane's peak, in order: found -> add, run + 1; in range and not found -> run ends;
                // past the spectrum -> the loop is over (stft.rs:503-505)
                float acc = fund_mag;
                int longest_run = 0, current_run = 0, total_harms = 0;
                bool all_noconf = true;
#pragma unroll
                for (int hn = 0; hn < kPitH; ++hn) {
                    const uint32_t word = hw[kPitH * lane + hn];
                    const float mg = hm[kPitH * lane + hn];
                    const bool in_range = (word >> 25) & 1u, found = (word & 0xFFFFFFu) != 0u;
                    all_noconf = all_noconf && (!in_range || ((word >> 24) & 1u));
                    if (found) {
                        acc += mg;
                        current_run += 1;
                        total_harms += 1;
                    } else if (in_range) {
                        longest_run = current_run > longest_run ? current_run : longest_run;
                        current_run = 0;
                    }
                }
                longest_run = current_run > longest_run ? current_run : longest_run;
                if (!all_noconf)
                    acc = walk_slow(k, frac_bin, fund_mag, longest_run, total_harms);
                float sc = 0.0f;
                if (!(longest_run < 3 && below15)) {
                    const float log_score = log2f(0.5f + acc);
                    const float struct_mult =
                        (1.0f + (float)longest_run + (float)total_harms / 2.0f) / (1.0f + (float)kMaxHarmonics);
                    sc = log_score * struct_mult;
                }
                score[g0 + lane] = sc;
                mx = fmaxf(mx, sc);
            }
            __syncthreads();
        }
        mx = pit_wave_max(mx);
        if (n_strong > 0 && mx != 0.0f) {
            const float cutoff = mx * 0.5f;
            // ---- candidates (score >= cutoff), bin order; cand aliases peak_bin: entry q is written at or below q ----
            int n_cand = 0;
            for (int base = 0; base < n_strong; base += 64) {
                const int p = base + lane;
                const bool c = p < n_strong && score[p] >= cutoff;
                const unsigned long long mask = __ballot(c);
                if (c)
                    cand[n_cand + __popcll(mask & lt)] = (uint16_t)p;
                n_cand += __popcll(mask);
            }
            __syncthreads();
            // ---- harmonic ghosts (stft.rs:562-585): candidate i against every other one ----
            int n_kept = 0;
            auto ghost = [&](float freq_i, float si, int j) {
                const float freq_j = frac[cand[j]] * bin_width, score_j = score[cand[j]];
                const float ratio = freq_i / freq_j;
                const float nearest = roundf(ratio);
                return nearest >= 2.0f && nearest <= 5.0f && fabsf(ratio / nearest - 1.0f) < 0.03f && si < score_j * 1.05f;
            };
            if (n_cand <= 32) {
                // (i, j) pairs across the lanes: P = the power of two >= n_cand lanes per candidate i
                int lg = 0;
                while ((1 << lg) < n_cand)
                    ++lg;
                const int P = 1 << lg, per_pass = 64 >> lg;
                const int j = lane & (P - 1);
                for (int i0 = 0; i0 < n_cand; i0 += per_pass) {
                    const int i = i0 + (lane >> lg);
                    const bool ok = i < n_cand && j < n_cand;
                    float fi = 0.0f, si = 0.0f;
                    bool sup = false;
                    if (ok) {
                        fi = frac[cand[i]];
                        si = score[cand[i]];
                        sup = j != i && ghost(fi * bin_width, si, j);
                    }
                    const unsigned long long supm = __ballot(sup);
                    const unsigned long long grp = P == 64 ? supm : (supm >> (lane & ~(P - 1))) & ((1ull << P) - 1ull);
                    const bool keep = ok && j == 0 && grp == 0ull;
                    const unsigned long long mask = __ballot(keep);
                    if (keep) {
                        const int pos = n_kept + __popcll(mask & lt);
                        kfrac[pos] = fi;
                        kscore[pos] = si;
                    }
                    n_kept += __popcll(mask);
                }
            } else {
                for (int base = 0; base < n_cand; base += 64) {
                    const int i = base + lane;
                    bool keep = false;
                    float fi = 0.0f, si = 0.0f;
                    if (i < n_cand) {
                        fi = frac[cand[i]];
                        si = score[cand[i]];
                        bool suppressed = false;
                        for (int j = 0; j < n_cand && !suppressed; ++j)
                            suppressed = j != i && ghost(fi * bin_width, si, j);
                        keep = !suppressed;
                    }
                    const unsigned long long mask = __ballot(keep);
                    if (keep) {
                        const int pos = n_kept + __popcll(mask & lt);
                        kfrac[pos] = fi;
                        kscore[pos] = si;
                    }
                    n_kept += __popcll(mask);
                }
            }
            __syncthreads();
            // ---- descending score, equal scores in bin order: rank = elements that come first ----
            for (int i = lane; i < n_kept; i += 64) {
                const float si = kscore[i];
                int rank = 0;
                for (int j = 0; j < n_kept; ++j)
                    rank += (kscore[j] > si || (kscore[j] == si && j < i)) ? 1 : 0;
                sfrac[rank] = kfrac[i];
                sscore[rank] = si;
            }
            __syncthreads();
            // ---- 2-bin dedupe in score order, first 8 (stft.rs:594-607): lane a < 8 holds accepted note a ----
            float acc_frac = 0.0f;
            int n_acc = 0;
            for (int i = 0; i < n_kept && n_acc < kMaxNotes; ++i) {
                const float fi = sfrac[i];
                const bool conflict = __ballot(lane < n_acc && fabsf(fi - acc_frac) < 2.0f) != 0ull;
                if (!conflict) {
                    if (lane == n_acc)
                        acc_frac = fi;
                    ++n_acc;
                    const float freq = fi * bin_width;
                    if (freq >= min_freq && freq <= max_freq) {
                        if (lane == n_out) {
                            my_freq = freq;
                            my_score = sscore[i];
                        }
                        ++n_out;
                    }
                }
            }
        }
        if (lane < kMaxNotes) {
            out_freq[t * kMaxNotes + lane] = my_freq;
            out_score[t * kMaxNotes + lane] = my_score;
        }
        if (lane == 0)
            out_count[t] = n_out;
    }
}

// One frame of one bin of the tuner's noise floor (stft.rs:326-367): the state (floor, previous magnitude, volatility)
// moves on, the effective floor min(floor, 2.5 x global floor) is returned.  The stream's very first frame sets the
// state instead (tuner_floor_first); the callers peel it off their loops - as a flag tested per frame it cost a branch
// and an exec mask in every one of them.
__device__ __forceinline__ float tuner_floor_first(float &fl, float &prev, float mag, float gf)
{
#pragma clang fp contract(off)
    fl = fmaxf(mag, gf * 5.0f);
    prev = mag;
    return fminf(fl, gf * 2.5f);
}
__device__ __forceinline__ float tuner_floor_step(float &fl, float &prev, float &vol, float mag, float gf)
{
#pragma clang fp contract(off)
    const float FLOOR_BASE_ALPHA = 0.04f, FLOOR_FAST_ALPHA = 0.35f, FLOOR_RELEASE = 0.02f;
    const float VOL_MEMORY = 0.75f, NOTE_VOL_MAX = 0.15f;     // (NOTE_RATIO = 1.5: in the exact test below)
    const float delta = fabsf(mag - prev);
    vol = vol * VOL_MEMORY + delta * (1.0f - VOL_MEMORY);
    prev = mag;
    // r5: `mag / floor.max(0.01) > NOTE_RATIO` (stft.rs:346, 351) without the division.  The quotient is used for
    // nothing but this comparison, and for a correctly rounded q = RN(a / b): q > 1.5 <=> a / b > 1.5 + 2^-24 (the
    // midpoint above 1.5; a tie rounds to the even 1.5) <=> a - 1.5 b > b 2^-24.  With b in [2^e, 2^(e+1)) both a
    // (near 1.5 b) and 1.5 b are multiples of 2^(e-24), so wherever the difference is small enough to matter
    // (< 2^e) ONE fused multiply-add returns it exactly, and beyond that its rounded value has the right sign
    // and size; b 2^-24 is an exact scaling (b >= 0.01).  Infinities, NaNs, over- and underflowing quotients come
    // out the same (tests/test_gpu_parity.py::test_exact_ratio_test_is_the_division); three f32 instructions for
    // twelve (first version of the round: a double product, four instructions at half rate).
    const float den = fmaxf(fl, 0.01f);
    const bool above = __builtin_fmaf(den, -1.5f, mag) > den * 5.9604644775390625e-08f;
    float vn = vol / fmaxf(mag, 0.05f);
    vn = vn < 0.0f ? 0.0f : (vn > 1.0f ? 1.0f : vn);
    const bool sustained = above && vn < NOTE_VOL_MAX;
    const float alpha = mag > fl ? FLOOR_BASE_ALPHA + (FLOOR_FAST_ALPHA - FLOOR_BASE_ALPHA) * vn : FLOOR_RELEASE;
    const float moved = fl + alpha * (mag - fl);
    fl = sustained ? fl : moved;
    return fminf(fl, gf * 2.5f);
}

// r5: the batch form.  tuner_floor_kernel below gives every wave 64 bins of one stream: 256-byte pieces of rows that are
// 4100 bytes apart - three cache lines for two, the partial ones filled before they are written (343 + 254 MB moved for
// 2 x 231 MB) - and with trivial arithmetic in place of the recurrence the launch takes the same 155-165 us: it is bound
// by that access pattern.  Here ONE workgroup owns a stream: the rows of kRowsT consecutive frames are one contiguous
// span of the spectrogram, fetched as aligned 16-byte pieces (the next span's while this one is walked), handed over
// through LDS, walked in place one lane per bin, and written back as aligned 16-byte pieces of the effective-floor
// rows.  A row of the tuner geometry has 1025 bins and a workgroup 1024 lanes: wave 0 walks the last bin as a second
// pass.  Same arithmetic, same order: bit-identical to the kernel below and to the C restatement.
// a workgroup barrier that orders the LDS accesses only: __syncthreads() also waits for every global load and store in
// flight - the next span's pieces, fetched to arrive DURING the walk, and the row stores of the span before
__device__ __forceinline__ void rows_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// kRowsT frames per span, kRowsP 16-byte pieces per lane (>= (kRowsT * 1025 + 9) / 4 / 1024).  Eight frames while a
// CU holds one workgroup (fewer steps: 137 us at 256 streams against 174 with four); four frames from two workgroups
// per CU on - the other workgroup's walk fills the time this one moves its rows (121 us against 140 at 512 streams).
template <int kRowsT, int kRowsP>
__global__ __launch_bounds__(1024) void tuner_floor_rows_kernel(
    const float *__restrict__ mags, const float *__restrict__ global_floor, float *__restrict__ state,
    float *__restrict__ eff, int n_bins, const int64_t *__restrict__ offsets, const int32_t *__restrict__ init_flags)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float rows_lds[];
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t f0 = uniform_i64(offsets[blockIdx.x]);
    const int64_t n_frames = uniform_i64(offsets[blockIdx.x + 1]) - f0;
    if (n_frames <= 0)
        return;
    mags += f0 * n_bins;
    eff += f0 * n_bins;
    global_floor += f0;
    state += (int64_t)blockIdx.x * 3 * n_bins;
    const bool seen = (init_flags ? init_flags[blockIdx.x] : 0) != 0;
    const bool valid = tid < n_bins;
    const int k = valid ? tid : n_bins - 1;
    const bool extra = n_bins == nthr + 1 && wave == 0;          // (wave-uniform) the row's last bin rides on wave 0
    const int kx = n_bins - 1;
    float fl = state[k], prev = state[n_bins + k], vol = state[2 * n_bins + k];
    float xfl = 0.0f, xprev = 0.0f, xvol = 0.0f;
    if (extra) {
        xfl = state[kx];
        xprev = state[n_bins + kx];
        xvol = state[2 * n_bins + kx];
    }
    const int n_pieces = (kRowsT * n_bins + 6 + 3) >> 2;         // covers a span at any 16-byte phase
    const int64_t n_steps = (n_frames + kRowsT - 1) / kRowsT;
    v4f regs[kRowsP];
    float gfv = 0.0f;
    // span of step s: floats [s0, s0 + n) of the stream's rows; piece q = floats 4 q - lead .. + 3 of it, fetched from
    // the 16-byte boundary below s0 (pieces that straddle an end of the span float by float: nothing outside is touched)
    auto span_of = [&](int64_t s, const float *base, int64_t &s0, int &n, int &lead) {
        s0 = s * kRowsT * (int64_t)n_bins;
        const int64_t left = n_frames - s * kRowsT;
        n = (int)(left < kRowsT ? left : kRowsT) * n_bins;
        lead = (int)((reinterpret_cast<uintptr_t>(base + s0) & 15u) >> 2);
    };
    auto fetch = [&](int64_t s) {
        int64_t s0;
        int n, lead;
        span_of(s, mags, s0, n, lead);
        const float *base = mags + s0 - lead;                    // 16-byte aligned
#pragma unroll
        for (int u = 0; u < kRowsP; ++u) {
            const int q = tid + nthr * u, j = 4 * q - lead;
            v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (j >= 0 && j + 4 <= n) {
                v = *reinterpret_cast<const v4f *>(base + 4 * q);
            } else if (j + 4 > 0 && j < n) {
                v.x = (j + 0 >= 0 && j + 0 < n) ? base[4 * q + 0] : 0.0f;
                v.y = (j + 1 >= 0 && j + 1 < n) ? base[4 * q + 1] : 0.0f;
                v.z = (j + 2 >= 0 && j + 2 < n) ? base[4 * q + 2] : 0.0f;
                v.w = (j + 3 >= 0 && j + 3 < n) ? base[4 * q + 3] : 0.0f;
            }
            regs[u] = v;
        }
        const int64_t t = s * kRowsT + lane;                     // lane i < kRowsT: the global floor of frame i of the span
        gfv = (lane < kRowsT && t < n_frames) ? global_floor[t] : 0.0f;
    };
    fetch(0);
    for (int64_t s = 0; s < n_steps; ++s) {
        int64_t s0;
        int n, lead;
        span_of(s, mags, s0, n, lead);
#pragma unroll
        for (int u = 0; u < kRowsP; ++u) {
            const int q = tid + nthr * u;
            if (q < n_pieces)
                reinterpret_cast<v4f *>(rows_lds)[q] = regs[u];
        }
        const float gf_here = gfv;
        rows_barrier();                                          // the span is in LDS
        if (s + 1 < n_steps)
            fetch(s + 1);                                        // in flight while this span is walked
        const int nfb = n / n_bins;
        // All reads of a column stand in front of its writes (in place the compiler cannot move one over the other).  No
        // lane is masked: the lanes past the row (always in the wave of its last bin) and the 64 lanes of wave 0 on the
        // extra bin read the same values in lockstep and write the same results to the same slots.
        const bool fresh = s == 0 && !seen;                      // (uniform) the span opens with the stream's first frame
        auto walk = [&](float *col, float &a, float &b, float &c) {
            float m[kRowsT];
#pragma unroll
            for (int i = 0; i < kRowsT; ++i)
                m[i] = col[i * n_bins];                          // (rows past the span's last one: stale LDS, not used)
            auto gf_of = [&](int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gf_here), i)); };
            if (fresh)
                col[0] = tuner_floor_first(a, b, m[0], gf_of(0));
            else
                col[0] = tuner_floor_step(a, b, c, m[0], gf_of(0));
            if (nfb == kRowsT) {
#pragma unroll
                for (int i = 1; i < kRowsT; ++i)
                    col[i * n_bins] = tuner_floor_step(a, b, c, m[i], gf_of(i));
            } else {
#pragma unroll
                for (int i = 1; i < kRowsT; ++i)
                    if (i < nfb)                                 // (wave-uniform: a stream's last span)
                        col[i * n_bins] = tuner_floor_step(a, b, c, m[i], gf_of(i));
            }
        };
        walk(rows_lds + lead + k, fl, prev, vol);
        if (extra)
            walk(rows_lds + lead + kx, xfl, xprev, xvol);
        rows_barrier();                                          // the span holds the effective floors
        {
            const int lead_e = (int)((reinterpret_cast<uintptr_t>(eff + s0) & 15u) >> 2);
            float *base = eff + s0 - lead_e;                     // 16-byte aligned
#pragma unroll
            for (int u = 0; u < kRowsP; ++u) {
                const int q = tid + nthr * u, j = 4 * q - lead_e;
                const float *src = rows_lds + lead + j;          // float j of the span
                if (j >= 0 && j + 4 <= n) {
                    // (both buffers at one 16-byte phase - the usual case: one ds_read_b128; four dword reads at a
                    // stride of four banks are served eight lanes at a time)
                    const v4f v = lead == lead_e ? reinterpret_cast<const v4f *>(rows_lds)[q] : v4f{src[0], src[1], src[2], src[3]};
                    *reinterpret_cast<v4f *>(base + 4 * q) = v;
                } else if (j + 4 > 0 && j < n) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (j + c >= 0 && j + c < n)
                            base[4 * q + c] = src[c];
                }
            }
        }
        rows_barrier();                                          // every lane has read its pieces: the next span may land
    }
    if (valid) {
        state[k] = fl;
        state[n_bins + k] = prev;
        state[2 * n_bins + k] = vol;
    }
    if (extra && lane == 0) {
        state[kx] = xfl;
        state[n_bins + kx] = xprev;
        state[2 * n_bins + kx] = xvol;
    }
}

// Tuner front end (src/audio_io/stft.rs:326-367), first stage of "next" row f-1: the variance-
// aware per-bin noise floor.  It is a recurrence over frames, independent per bin: one lane per
// bin walks the frames in order with exactly the reference's operation order (contraction off,
// IEEE division), so the result is bit-identical to the CPU restatement.  Parallelism is only
// n_bins lanes per stream; rows are read and written coalesced across bins.
__global__ __launch_bounds__(64) void tuner_floor_kernel(const float *__restrict__ mags,
                                                         const float *__restrict__ global_floor,
                                                         float *__restrict__ state, int initialized,
                                                         float *__restrict__ eff, int64_t n_frames,
                                                         int n_bins, const int64_t *__restrict__ offsets,
                                                         const int32_t *__restrict__ init_flags)
{
#pragma clang fp contract(off)
    // (every lane of the wave stays: the frames' global floors are handed round by v_readlane; lanes past the row
    // walk its last bin and store nothing)
    const bool valid = (int)(blockIdx.x * 64 + threadIdx.x) < n_bins;
    const int k = valid ? (int)(blockIdx.x * 64 + threadIdx.x) : n_bins - 1;
    if (offsets) {
        // batch: blockIdx.y walks stream s = frames [offsets[s], offsets[s+1]) of the concatenated arrays
        const int64_t f0 = uniform_i64(offsets[blockIdx.y]);        // (scalar: the row pointers below stay scalar)
        n_frames = uniform_i64(offsets[blockIdx.y + 1]) - f0;
        mags += f0 * n_bins;
        eff += f0 * n_bins;
        global_floor += f0;
        state += (int64_t)blockIdx.y * 3 * n_bins;
        initialized = init_flags ? init_flags[blockIdx.y] : 0;
    }
    float fl = state[k], prev = state[n_bins + k], vol = state[2 * n_bins + k];
    auto frame = [&](float mag, float gf) { return tuner_floor_step(fl, prev, vol, mag, gf); };
    // r4: blocks of kFloorT frames - the bin's magnitudes and the frames' global floors of a block are fetched together
    // (one memory round trip per block instead of one per frame: a ten-minute stream 26.7 -> 7.6 ms), then walked.
    // r5: the NEXT block is fetched before this one is walked (a lone wave sat out the round trip once per block:
    // 7.6 -> 6.8 ms for the one stream); rows are addressed as a scalar pointer + one byte offset per lane (per-lane
    // 64-bit pointers cost a vector add per load and per store); the lanes past the row walk its last bin and store ITS
    // value again - no branch; the stream's first frame is peeled off.  This kernel serves up to 255 streams: at 256
    // streams its launch (153-163 us by box) moves 343 + 254 MB for 2 x 231 MB algorithmic (rows of 4100 B in 256-byte
    // pieces per wave: three cache lines for two, partial lines filled before they are written) and takes the same
    // time with the recurrence replaced by an addition - tuner_floor_rows_kernel above takes the batches.
    constexpr int kFloorT = 32;
    const int lane = threadIdx.x;
    const unsigned voff = 4u * (unsigned)k;
    float cur[kFloorT], nxt[kFloorT];
    float cur_gf = 0.0f, nxt_gf = 0.0f;
    auto fetch = [&](int64_t t0, float (&dst)[kFloorT], float &gfv) {
        const char *row = reinterpret_cast<const char *>(mags + t0 * n_bins);
#pragma unroll
        for (int i = 0; i < kFloorT; ++i) {
            unsigned vo = voff;
            asm volatile("" : "+v"(vo));         // (opaque per row: or base + offset becomes a per-lane pointer again)
            dst[i] = *reinterpret_cast<const float *>(row + vo);
            row += 4 * n_bins;
        }
        gfv = global_floor[t0 + (lane & (kFloorT - 1))];                 // lane i: frame i of the block
    };
    int64_t t = 0;
    if (!initialized && n_frames > 0) {                          // the stream's first frame sets the state
        const float e = tuner_floor_first(fl, prev, mags[k], global_floor[0]);
        if (valid)
            eff[k] = e;
        t = 1;
    }
    if (t + kFloorT <= n_frames)
        fetch(t, cur, cur_gf);
    for (; t + kFloorT <= n_frames; t += kFloorT) {
        if (t + 2 * kFloorT <= n_frames)
            fetch(t + kFloorT, nxt, nxt_gf);
        char *erow = reinterpret_cast<char *>(eff + t * n_bins);
#pragma unroll
        for (int i = 0; i < kFloorT; ++i) {
            const float e = frame(cur[i], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur_gf), i)));
            unsigned vo = voff;
            asm volatile("" : "+v"(vo));
            *reinterpret_cast<float *>(erow + vo) = e;
            erow += 4 * n_bins;
        }
#pragma unroll
        for (int i = 0; i < kFloorT; ++i)
            cur[i] = nxt[i];
        cur_gf = nxt_gf;
    }
    for (; t < n_frames; ++t) {
        const float e = frame(mags[t * n_bins + k], global_floor[t]);
        if (valid)
            eff[t * n_bins + k] = e;
    }
    if (valid) {
        state[k] = fl;
        state[n_bins + k] = prev;
        state[2 * n_bins + k] = vol;
    }
}

// mixed = (sum of the first min(channels, 2) channels) / that count — mod.rs:784-794
__global__ void downmix_kernel(const float *__restrict__ in, float *__restrict__ out,
                               int64_t n_frames, int channels)
{
    const int use = channels < 2 ? channels : 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_frames;
         i += (int64_t)gridDim.x * blockDim.x) {
        float acc = 0.0f;
        for (int c = 0; c < use; ++c)
            acc += in[i * channels + c];
        out[i] = acc / (float)use;
    }
}

// Interleaved stereo, 16-byte-aligned buffers: four frames per lane step - two 16-byte loads
// (L0 R0 L1 R1 | L2 R2 L3 R3), one nontemporal 16-byte store.  (L + R) * 0.5 is the very value of the
// general kernel: 0 + L + R is exact in the same order and the halving is exact.
typedef float dm4 __attribute__((ext_vector_type(4)));
__global__ void downmix_stereo4_kernel(const dm4 *__restrict__ in, dm4 *__restrict__ out, int64_t n_quads)
{
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads;
         q += (int64_t)gridDim.x * blockDim.x) {
        const dm4 a = __builtin_nontemporal_load(in + 2 * q), b = __builtin_nontemporal_load(in + 2 * q + 1);
        dm4 r;
        r.x = (0.0f + a.x + a.y) / 2.0f;
        r.y = (0.0f + a.z + a.w) / 2.0f;
        r.z = (0.0f + b.x + b.y) / 2.0f;
        r.w = (0.0f + b.z + b.w) / 2.0f;
        __builtin_nontemporal_store(r, out + q);
    }
}

// Diagnostics (include/aa_stft.h, aa_diag_stream_copy_device): the memory system's own ceiling for a launch's byte count,
// measured with the product's access idiom and NO arithmetic - 16-byte loads, 16-byte nontemporal stores, persistent
// workgroups on contiguous runs.  (Plain loads, as the product's LDS-DMA: a source that fits the 256-MB Infinity Cache
// is then served from it launch after launch, like the product's 115 MB of samples - 6.5 TB/s for the 173-MB copy
// where nontemporal loads, which pass the cache by, reach 5.1; tools/ubench_copy.hip, profiles/r05_ubench_copy.txt.)  WRITES = 1: a plain copy; WRITES = 2: every piece stored twice (dst[i], dst[n + i]),
// the 1 : 2 read : write mix of the magnitude launch (hop floats in, ~2 hop floats out per frame); WRITES = 4: the 1 : 4
// mix of the complex rows (hop floats in, 2 (n_fft / 2 + 1) out).
template <int WRITES>
__global__ __launch_bounds__(256) void stream_copy_kernel(const v4f *__restrict__ src, v4f *__restrict__ dst, int64_t n4)
{
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < n4 ? i0 + per : n4;
    int64_t i = i0 + threadIdx.x;
    for (; i + 3 * 256 < i1; i += 4 * 256) {
        v4f r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            r[u] = src[i + 256 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int w = 0; w < WRITES; ++w)
                __builtin_nontemporal_store(r[u], dst + w * n4 + i + 256 * u);
        }
    }
    for (; i < i1; i += 256) {
        const v4f r = src[i];
#pragma unroll
        for (int w = 0; w < WRITES; ++w)
            __builtin_nontemporal_store(r, dst + w * n4 + i);
    }
}

}  // namespace

hipError_t launch_stream_copy(const float *src, float *dst, int64_t n_floats, int writes, hipStream_t stream)
{
    if (n_floats <= 0)
        return hipSuccess;
    if ((n_floats & 3) || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) ||
        (writes != 1 && writes != 2 && writes != 4))
        return hipErrorInvalidValue;
    const int64_t n4 = n_floats / 4;
    int64_t blocks = (n4 + 1023) / 1024;
    if (blocks > 2048)
        blocks = 2048;                               // eight workgroups per CU
    if (writes == 1)
        hipLaunchKernelGGL(stream_copy_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream,
                           reinterpret_cast<const v4f *>(src), reinterpret_cast<v4f *>(dst), n4);
    else if (writes == 2)
        hipLaunchKernelGGL(stream_copy_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream,
                           reinterpret_cast<const v4f *>(src), reinterpret_cast<v4f *>(dst), n4);
    else
        hipLaunchKernelGGL(stream_copy_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream,
                           reinterpret_cast<const v4f *>(src), reinterpret_cast<v4f *>(dst), n4);
    return hipGetLastError();
}

hipError_t launch_rows_gemm(bool take_log, const float *A, int lda, const float *B, const int *tile_k,
                            float *C, int64_t n_rows, int K, int n_cols, int n_pad, float log_floor,
                            hipStream_t stream)
{
    if (n_rows <= 0)
        return hipSuccess;
    if (n_pad > 128 || (n_pad & 31))
        return hipErrorInvalidValue;
    dim3 grid((unsigned)((n_rows + kGemmRows - 1) / kGemmRows));
    const int2 *tk = reinterpret_cast<const int2 *>(tile_k);
    if (take_log)
        hipLaunchKernelGGL(rows_gemm_mfma_kernel<true>, grid, dim3(256), 0, stream, A, lda, B, tk, C, n_rows,
                           K, n_cols, n_pad, log_floor);
    else
        hipLaunchKernelGGL(rows_gemm_mfma_kernel<false>, grid, dim3(256), 0, stream, A, lda, B, tk, C, n_rows,
                           K, n_cols, n_pad, log_floor);
    return hipGetLastError();
}

hipError_t launch_onset_flux(const float *mags, const float *prev0, float *flux, float *energy,
                             int64_t n_frames, int n_bins, hipStream_t stream, const int64_t *offsets,
                             int n_streams)
{
    if (n_frames <= 0)
        return hipSuccess;
    const int64_t n_blocks = (n_frames + kFluxT - 1) / kFluxT;
    // the flat form holds a block's 65 rows (+ 3 floats of alignment) in 33 sixteen-byte pieces per lane
    const bool flat = (int64_t)(kFluxT + 1) * n_bins + 3 <= 4 * 64 * kFluxV4;
    const int wsh_len = ((n_bins + 3) & ~3) + kFluxCh;
    const size_t lds = sizeof(float) * (size_t)((flat ? 4 * 64 * kFluxV4 : (kFluxT + 1) * kFluxTileStride + 1) + wsh_len + 64 +
                                                (flat ? wsh_len : kFluxTileStride + 1));   // image | weights | starts | zero row
    if (lds > 64 * 1024)
        return hipErrorInvalidValue;
    // persistent waves, one per workgroup: four fit a CU's LDS (34 KB each), blocks dealt round robin
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0)
            cus = 256;
    }
    int64_t blocks = (int64_t)cus * 4;
    if (blocks > n_blocks)
        blocks = n_blocks;
    if (flat)
        hipLaunchKernelGGL(onset_flux_kernel<true>, dim3((unsigned)blocks), dim3(64), lds, stream, mags, prev0, flux,
                           energy, n_frames, n_bins, offsets, n_streams);
    else
        hipLaunchKernelGGL(onset_flux_kernel<false>, dim3((unsigned)blocks), dim3(64), lds, stream, mags, prev0, flux,
                           energy, n_frames, n_bins, offsets, n_streams);
    return hipGetLastError();
}

hipError_t launch_flux_div3_check(uint32_t first, uint64_t n, unsigned long long *mismatches_dev, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(flux_div3_check_kernel, dim3(4096), dim3(256), 0, stream, first, n, mismatches_dev);
    return hipGetLastError();
}

// blocks of 16 frames from here on: up to 3 x 256 streams the T = 32 workgroups are all resident (measured: 512 streams
// 0.108 ms against 0.115, 1024 streams 0.144 against 0.100)
#ifndef AA_GATES_T16_FROM
#define AA_GATES_T16_FROM 768
#endif

hipError_t launch_onset_gates(const float *mags, const float *flux, const float *energy,
                              const float *global_floor, float *state, int initialized, int64_t n_frames,
                              int n_bins, float *out_flux, float *out_ema, float *out_excess,
                              int32_t *out_count, uint8_t *out_flags, float *out_velocity, hipStream_t stream,
                              const int64_t *offsets, const int32_t *init_flags, int n_streams,
                              const uint8_t *suppress, int calibrated)
{
    if (n_frames <= 0 || n_streams <= 0)
        return hipSuccess;
    if (n_bins < 1 || n_bins > 64 * kGatePrefetchRegs)
        return hipErrorInvalidValue;
    // the onset geometry: the recurrences taken apart, one workgroup of five waves per stream (a single ten-minute
    // stream: 313 -> ~30 ms).  r5: at every stream count - rounds 1-4 went back to one wave per stream from 256 streams
    // on, i.e. 1024 streams were 1024 lone waves walking 440 frames at 0.7 us each (351 us).
    // r5, second pass (tools/gates_sweep.py, 449 997 frames of 129 bins): a loop per role and scalar-base row loads took
    // the kernel from 194 to 118 registers (T = 32: three workgroups per CU, one before) and to 78 in blocks of 16 frames
    // (four per CU) - 1024 streams 186 -> 100 us, 8192 streams 0.48 -> 0.14 ms, and ahead of the one-wave kernel down to
    // streams of three frames (131 072 streams: 0.89 against 1.58 ms), which is left with the wide spectra
    if (n_bins <= 192) {
        const int stride = (n_bins + 1) | 1;
#define AA_GATES_SPLIT(T)                                                                                             \
    hipLaunchKernelGGL(onset_gates_split_kernel<T>, dim3((unsigned)n_streams), dim3(320),                              \
                       sizeof(float) * (size_t)(gate_ops_dwords(T) + 2 * T * stride + 4 * T), stream, mags, flux,      \
                       energy, global_floor, state, initialized, n_frames, n_bins, out_flux, out_ema, out_excess,      \
                       out_count, out_flags, out_velocity, offsets, init_flags, suppress, calibrated)
        if (n_streams > AA_GATES_T16_FROM)
            AA_GATES_SPLIT(16);
        else
            AA_GATES_SPLIT(32);
#undef AA_GATES_SPLIT
        return hipGetLastError();
    }
    int fpb = 64 * kGatePrefetchRegs / n_bins;       // frames per prefetched block, one lane per frame
    if (fpb > 64)
        fpb = 64;
    const size_t lds = sizeof(float) * (size_t)((fpb + 1) * n_bins);   // + the pad row read one frame ahead
    const int bpl = (n_bins + 63) / 64;
#define AA_GATES(B)                                                                                      \
    hipLaunchKernelGGL(onset_gates_kernel<B>, dim3((unsigned)n_streams), dim3(64), lds, stream, mags, flux, \
                       energy, global_floor, state, initialized, n_frames, n_bins, fpb, out_flux, out_ema,  \
                       out_excess, out_count, out_flags, out_velocity, offsets, init_flags, suppress, calibrated)
    if (bpl <= 3)
        AA_GATES(3);        // 129 bins: the onset geometry (onset.rs:122-125)
    else if (bpl <= 9)
        AA_GATES(9);
    else if (bpl <= 17)
        AA_GATES(17);
    else
        AA_GATES(65);
#undef AA_GATES
    return hipGetLastError();
}

hipError_t launch_tuner_pitches(const float *mags, const float *eff_floor, int64_t n_frames, int n_bins,
                                float bin_width, float min_freq, float max_freq, float *out_freq,
                                float *out_score, int32_t *out_count, hipStream_t stream)
{
    if (n_frames <= 0)
        return hipSuccess;
    // the reference's bin range (stft.rs:454-455), evaluated in f32 like it
    int min_bin = (int)ceilf(min_freq / bin_width);
    if (min_bin < 1)
        min_bin = 1;
    int max_bin = (int)floorf(max_freq / bin_width);
    if (max_bin > n_bins - 2)
        max_bin = n_bins - 2;
    // peaks lie strictly between min_bin and max_bin (plateaus can make neighbouring bins peaks)
    int max_peaks = max_bin - min_bin - 1;
    if (max_peaks < 1)
        max_peaks = 1;
    const int nbp = (n_bins + 3) & ~3, nw = (n_bins + 63) >> 6;
    if (n_bins > 0x7fff)
        return hipErrorInvalidValue;                 // (a peak's bin travels in 15 bits)
    size_t words = (size_t)nbp + (size_t)(2 * nw + 2) + (size_t)((max_peaks + 1) / 2) + 2 * (size_t)max_peaks +
                   2 * (size_t)(kPitGroup * kPitH);
    if (2 * max_peaks > nbp)
        words += 2 * (size_t)max_peaks;              // the kept notes do not fit the dead magnitude row
    const size_t lds = sizeof(float) * words;
    if (lds > 64 * 1024)
        return hipErrorInvalidValue;
    int64_t blocks = n_frames < 16384 ? n_frames : 16384;
    if (n_bins <= 64 * 17)
        hipLaunchKernelGGL(tuner_pitches_kernel<true>, dim3((unsigned)blocks), dim3(64), lds, stream, mags, eff_floor,
                           n_frames, n_bins, bin_width, min_freq, max_freq, min_bin, max_bin, max_peaks, out_freq,
                           out_score, out_count);
    else
        hipLaunchKernelGGL(tuner_pitches_kernel<false>, dim3((unsigned)blocks), dim3(64), lds, stream, mags, eff_floor,
                           n_frames, n_bins, bin_width, min_freq, max_freq, min_bin, max_bin, max_peaks, out_freq,
                           out_score, out_count);
    return hipGetLastError();
}

#ifndef AA_FLOOR_ROWS_FROM
#define AA_FLOOR_ROWS_FROM 256
#endif
hipError_t launch_tuner_floor(const float *mags, const float *global_floor, float *state,
                              int initialized, float *eff, int64_t n_frames, int n_bins,
                              hipStream_t stream, const int64_t *offsets, const int32_t *init_flags, int n_streams)
{
    if (n_frames <= 0 || n_streams <= 0)
        return hipSuccess;
    // batches of streams whose rows fit a workgroup (one lane per bin, the tuner's 1025th bin as a second pass of wave 0):
    // rows through LDS in contiguous spans (tuner_floor_rows_kernel)
    const int rows_threads = n_bins == 1025 ? 1024 : ((n_bins + 63) / 64) * 64;
    if (offsets && n_streams >= AA_FLOOR_ROWS_FROM && n_bins >= 64 && rows_threads <= 1024) {
        const int T = n_streams >= 2 * AA_FLOOR_ROWS_FROM ? 4 : 8, P = T == 4 ? 2 : 3;
        const size_t lds = sizeof(float) * (size_t)(4 * ((T * n_bins + 6 + 3) >> 2));
        if ((T * n_bins + 6 + 3) / 4 <= P * rows_threads && lds <= 64 * 1024) {
            if (T == 4)
                hipLaunchKernelGGL((tuner_floor_rows_kernel<4, 2>), dim3((unsigned)n_streams), dim3((unsigned)rows_threads),
                                   lds, stream, mags, global_floor, state, eff, n_bins, offsets, init_flags);
            else
                hipLaunchKernelGGL((tuner_floor_rows_kernel<8, 3>), dim3((unsigned)n_streams), dim3((unsigned)rows_threads),
                                   lds, stream, mags, global_floor, state, eff, n_bins, offsets, init_flags);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL(tuner_floor_kernel, dim3((unsigned)((n_bins + 63) / 64), (unsigned)n_streams), dim3(64), 0,
                       stream, mags, global_floor, state, initialized, eff, n_frames, n_bins, offsets, init_flags);
    return hipGetLastError();
}

hipError_t launch_downmix(const float *in, float *out, int64_t n_frames, int channels,
                          hipStream_t stream)
{
    if (n_frames <= 0)
        return hipSuccess;
    // stereo from aligned buffers (the reference's capture format, mod.rs:789-793): the vector kernel
    // for the whole quads, the general one for the last 0..3 frames
    int64_t done = 0;
    if (channels == 2 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0 &&
        n_frames >= 4) {
        const int64_t quads = n_frames / 4;
        int64_t blocks = (quads + 255) / 256;
        if (blocks > 8192)
            blocks = 8192;
        hipLaunchKernelGGL(downmix_stereo4_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                           reinterpret_cast<const dm4 *>(in), reinterpret_cast<dm4 *>(out), quads);
        done = 4 * quads;
        if (done == n_frames)
            return hipGetLastError();
    }
    int64_t blocks = (n_frames - done + 255) / 256;
    if (blocks > 4096)
        blocks = 4096;
    hipLaunchKernelGGL(downmix_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in + done * channels,
                       out + done, n_frames - done, channels);
    return hipGetLastError();
}

}  // namespace aa
