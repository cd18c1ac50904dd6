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

// AA_SAFE_WAITS=1 (tools/check_counted_waits.sh only): every counted s_waitcnt becomes a full wait.  A
// library built that way must produce the same bits as the shipped one; a counted wait that let a
// load or an LDS-DMA piece slip would show as a difference.
#ifndef AA_SAFE_WAITS
#define AA_SAFE_WAITS 0
#endif
// ------------------------------------------------------------------ helpers --
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// 32-bit LDS byte address of a __shared__ object (for hand-issued DS instructions).
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p;
}
// One ds_read_b64 that the load/store optimizer cannot fuse into ds_read2_b64 (which
// costs 8 LDS cycles per pair instead of 2+2 and banks on 32 instead of 64 banks —
// profiles/r01_ubench_valu_gfx950.txt).  The result is valid after lds_wait_*().
template <int BYTE_OFF>
__device__ __forceinline__ v2f lds_read_b64(unsigned addr)
{
    v2f r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(BYTE_OFF) : "memory");
    return r;
}
__device__ __forceinline__ void lds_wait_8(v2f &a, v2f &b, v2f &c, v2f &d, v2f &e, v2f &f, v2f &g, v2f &h)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::"memory");
}
// "At most N vector-memory operations still in flight."  vmcnt retires in issue order and counts
// stores, so with N = the number of row stores a wave issues per chunk AFTER the LDS-DMA loads of
// the next chunk, the loads are complete while those stores stay in flight.  N must never exceed
// that number (a larger N would let the last DMA piece still be flying): every kernel derives it
// from the constexpr store count of the epilogue it instantiates.
template <int N>
__device__ __forceinline__ void wait_vmcnt_le()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter on gfx9");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AA_SAFE_WAITS ? 0 : N) : "memory");
}
// One LDS-DMA piece (global_load_lds_dwordx4: every lane moves 16 bytes from its own global address to
// lds_base + 16 lane), issued by hand so that the COMPILER does not know an LDS-DMA is in flight.  For the
// builtin it does: SIInsertWaitcnts puts an s_waitcnt vmcnt(0) in front of the first DS access that might
// alias the DMA's destination - with one LDS allocation, every access - i.e. in the middle of the chunk every
// wave waited for the samples of the NEXT chunk (and for all its row stores), which is exactly what the counted
// chunk-top waits exist to avoid.  Safety is those waits' job: data moved by a piece is read only behind
// wait_vmcnt_le<..>() + the workgroup barrier of the chunk that uses it.  lds_base must be wave-uniform.
__device__ __forceinline__ void lds_dma_16(const float *src_lane, float *lds_base)
{
    if (AA_SAFE_WAITS) {
        // the checking build (tools/check_counted_waits.sh): the compiler-tracked form, so that the compiler also puts
        // its own full waits in front of the LDS accesses that may alias a DMA in flight
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src_lane,
                                         (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
        return;
    }
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr(lds_base));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(src_lane), "s"(base)
                 : "memory");   // no "m0" clobber: the compiler rejects it (m0 is a reserved register, -Winline-asm), and what it emits
                                // itself that reads m0 (s_movrel, sendmsg, the LDS-DMA builtin) sets m0 right before the use - it holds no
                                // value in m0 across statements for this asm to destroy
}
// Row stores: nontemporal (the rows are written once and not read by this kernel).
template <typename T>
__device__ __forceinline__ void row_store(T v, T *p)
{
    __builtin_nontemporal_store(v, p);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 w)
{
    return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}

template <int OUT>
__device__ __forceinline__ float bin_from_power(float p, float log_floor)
{
    if (OUT == kMag)
        return __builtin_amdgcn_sqrtf(p);
    if (OUT == kPower || OUT == kLogMel || OUT == kLogMelM || OUT == kMfccM)
        return p;
    return 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(p, log_floor));
}

template <int OUT>
__device__ __forceinline__ float bin_value(float re, float im, float log_floor)
{
    float p = re * re + im * im;
    if (OUT == kMag)
        return __builtin_amdgcn_sqrtf(p);
    if (OUT == kPower || OUT == kLogMel || OUT == kLogMelM || OUT == kMfccM)
        return p;
    // 10*log10(p) = (10/log2(10)) * log2(p)
    return 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(p, log_floor));
}

// ---- packed-f32 complex arithmetic ------------------------------------------------
// A complex value lives in an aligned VGPR pair (v2f) and is combined with v_pk_*_f32;
// the rotations by -i / +i, conjugations and the complex multiply's swizzles ride on the
// VOP3P op_sel / neg modifiers, so they cost no instruction.  hipcc emits v_xor/v_mov pairs
// for those from vector code, hence the inline asm (pure register ops, no hazards of
// their own; the asm is not volatile so the compiler schedules it freely).
// Measured on gfx950 with 4 waves/SIMD: v_pk_add/mul/fma_f32 3.3 cycles for two lanes-ops
// against 2.0-2.8 for one scalar op (profiles/r01_ubench_valu_gfx950.txt).
__device__ __forceinline__ v2f pk_add_mi(v2f a, v2f b)   // a + (-i) b = (a.x + b.y, a.y - b.x)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_add_pi(v2f a, v2f b)   // a + (+i) b = (a.x - b.y, a.y + b.x)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_add_conj(v2f a, v2f b)  // a + conj(b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_sub_conj(v2f a, v2f b)  // a - conj(b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a.x + b.x, a.x - b.x) and (a.y + b.y, a.y - b.y): the real / imaginary parts of a + b and a - b
// side by side, so that |a + b|^2 and |a - b|^2 come out of one v_pk_mul + one v_pk_fma.
__device__ __forceinline__ v2f pk_sum_diff_re(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_sum_diff_im(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// The two halves of pk_cmul for loops that want every multiply ahead of every dependent FMA (a
// dependent pair back to back costs a wait state: the hazard recogniser puts an s_nop between them).
__device__ __forceinline__ v2f pk_cmul_lo(v2f a, v2f w)
{
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    return t;
}
__device__ __forceinline__ v2f pk_cmul_hi(v2f a, v2f w, v2f t)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// v[1..7] *= w[1..7]
__device__ __forceinline__ void pk_twiddle7(v2f (&v)[8], const v2f (&w)[8])
{
    v2f t[8];
#pragma unroll
    for (int kk = 1; kk < 8; ++kk)
        t[kk] = pk_cmul_lo(v[kk], w[kk]);
#pragma unroll
    for (int kk = 1; kk < 8; ++kk)
        v[kk] = pk_cmul_hi(v[kk], w[kk], t[kk]);
}
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f w)      // a * w
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));            // a * w.xx
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"     // + (-a.y, a.x) * w.yy
        : "=v"(r)
        : "v"(a), "v"(w), "v"(t));
    return r;
}
// c + (-i) h g  and  c + (+i) h g  for a packed constant h = (s, s): the rotation is the half swap
// of g (op_sel) and a sign on one half of h (neg), the scale rides in the FMA.
__device__ __forceinline__ v2f pk_fma_mi(v2f g, v2f h, v2f c)   // (c.x + h g.y, c.y - h g.x)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(g), "v"(h), "v"(c));
    return r;
}
__device__ __forceinline__ v2f pk_fma_pi(v2f g, v2f h, v2f c)   // (c.x - h g.y, c.y + h g.x)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(g), "v"(h), "v"(c));
    return r;
}
// 8-point DFT over the register index on packed values (same radix-2 DIF flow as dft8).  The two
// multiplications by sqrt(1/2) (W8^1, W8^3) are folded into the FMAs of the last stage:
// with p = (1 - i) b1 and q = (1 + i) b3:  W8^1 b1 = h p,  W8^3 b3 = -h q.
__device__ __forceinline__ void pk_dft8(v2f (&v)[8])
{
    const v2f hp = {0.70710678118654752440f, 0.70710678118654752440f};
    v2f a0 = v[0] + v[4], b0 = v[0] - v[4];
    v2f a1 = v[1] + v[5], b1 = v[1] - v[5];
    v2f a2 = v[2] + v[6], b2 = v[2] - v[6];
    v2f a3 = v[3] + v[7], b3 = v[3] - v[7];
    // (statement order: no operation reads the result of the one just before it - a dependent
    // packed pair back to back costs a wait state)
    const v2f p = pk_add_mi(b1, b1);                 // (x + y, y - x)
    const v2f q = pk_add_pi(b3, b3);                 // (x - y, x + y)
    v2f c0 = a0 + a2, c2 = a0 - a2;
    v2f c1 = a1 + a3, u = a1 - a3;                   // c3 = -i u
    v2f d0 = pk_add_mi(b0, b2), d2 = pk_add_pi(b0, b2);   // b0 +- (-i b2)
    const v2f e = p - q, g = p + q;                  // d1 = h e, w = h g
    v[0] = c0 + c1; v[4] = c0 - c1;
    v[2] = pk_add_mi(c2, u); v[6] = pk_add_pi(c2, u);
    v[1] = d0 + e * hp; v[5] = d0 - e * hp;
    v[3] = pk_fma_mi(g, hp, d2); v[7] = pk_fma_pi(g, hp, d2);
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

// The packed-math asm is invisible to hipcc's hazard recogniser: a VALU write followed by a
// DPP / permlane read of the same VGPR needs two wait states.  Tying the registers to the
// nop keeps every producer before it and every cross-lane consumer after it.
__device__ __forceinline__ void hazard_fence(v2f (&v)[8])
{
    asm volatile("s_nop 1"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}

// same exchange on packed values (vector elements are not addressable: go through floats)
__device__ __forceinline__ void xpose_pk(v2f &a, v2f &b, int stage)
{
    float ax = a.x, ay = a.y, bx = b.x, by = b.y;
    if (stage == 0) {
        xpose_pair_b3(ax, bx);
        xpose_pair_b3(ay, by);
    } else if (stage == 1) {
        xpose_pair_b4(ax, bx);
        xpose_pair_b4(ay, by);
    } else {
        xpose_pair_b5(ax, bx);
        xpose_pair_b5(ay, by);
    }
    a = v2f{ax, ay};
    b = v2f{bx, by};
}
__device__ __forceinline__ void exchange_high(v2f (&v)[8])
{
#pragma unroll
    for (int i = 0; i < 8; i += 2)
        xpose_pk(v[i], v[i + 1], 0);
#pragma unroll
    for (int g = 0; g < 8; g += 4)
#pragma unroll
        for (int i = g; i < g + 2; ++i)
            xpose_pk(v[i], v[i + 2], 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        xpose_pk(v[i], v[i + 4], 2);
}

// ---- fused mel reduction (OUT == kLogMel) ---------------------------------------------
// The Slaney filterbank is triangular (98 % zeros; a bin feeds at most two filters), so the fused
// path is a gather on the VALU from the power rows the epilogue left in the wave's LDS scratch —
// the rows never travel to HBM.  The wave's F frames get 64/F lanes each; within a frame, lane =
// (K-slice h, mel-in-group msub).  Group g (MelSched) covers (64/F) >> part_log2[g] mels from
// mel_base[g] and loops trips[g] bins per slice (wave-uniform; the filter width grows
// monotonically with the mel index, so a group's filters are alike).  The weight of a filter at
// offset i from its (host-adjusted, row-clamped) first bin is med3(U0 + i DU, D0 - i DD, 0): the
// two edges of the triangle with the Slaney norm folded in — zero outside the support, so no
// masks and no weight table.  Slices are summed with v_permlane32/16_swap and a DPP row_ror:8.
// Hand-issued ds_read2_b32 (two consecutive floats at dword offset O0 from a 4-byte-aligned LDS
// byte address) for the mel gather: the compiler would rotate the loop so that every step waits
// for its own loads — one LDS round trip per 4 bins.  The destination is tied ("+v"): the value
// lives in ONE register pair from issue to use, so the compiler never copies it while the load
// is still in flight.  Valid after mel_lds_wait().
template <int O0>
__device__ __forceinline__ void mel_lds_read2(v2f &r, unsigned addr)
{
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "+v"(r) : "v"(addr), "n"(O0), "n"(O0 + 1) : "memory");
}
// Same with a fresh (untied) destination, for straight-line code where every load has its own
// variable: no copy of a previous value is ever needed.
template <int O0>
__device__ __forceinline__ v2f mel_lds_load2(unsigned addr)
{
    v2f r;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(addr), "n"(O0), "n"(O0 + 1) : "memory");
    return r;
}
template <int KEEP>   // wait until at most KEEP newer LDS operations are outstanding (LDS returns in order)
__device__ __forceinline__ void mel_lds_wait(v2f &a, v2f &b)
{
    if (KEEP == 2)
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(AA_SAFE_WAITS ? 0 : 2) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");
}
// four bins of one filter slice: weights from the two triangle edges, accumulate.  The two edge
// lines advance as one packed pair (u, d) += (DU, -DD) and the products accumulate two bins per
// v_pk_fma_f32: the wave is issue bound here, packed math halves the slots of those parts.
__device__ __forceinline__ void mel_step4(v2f &acc2, float &fi, v2f pa, v2f pb, v2f ud0, v2f dud)
{
    const v2f f2 = {fi, fi};
    const v2f e0 = f2 * dud + ud0;
    const v2f e1 = e0 + dud, e2 = e1 + dud, e3 = e2 + dud;
    const v2f w01 = {__builtin_amdgcn_fmed3f(e0.x, e0.y, 0.0f), __builtin_amdgcn_fmed3f(e1.x, e1.y, 0.0f)};
    const v2f w23 = {__builtin_amdgcn_fmed3f(e2.x, e2.y, 0.0f), __builtin_amdgcn_fmed3f(e3.x, e3.y, 0.0f)};
    acc2 = w01 * pa + acc2;
    acc2 = w23 * pb + acc2;
    fi += 4.0f;
}
template <int F>
__device__ __forceinline__ void mel_reduce(const float *rows, int nb, const float *par, const MelSched &sch,
                                           float *stage)
{
    constexpr int LPF = 64 / F;                      // lanes per frame
    const int lane = threadIdx.x & 63;
    const int f = (F == 2) ? (lane >> 5) : 0;
    int lf = lane & (LPF - 1);
    // opaque to the optimiser: otherwise the per-group products of the lane index (loop
    // invariant) are hoisted out of the frame loop, held across the FFT phase and spilled
    asm volatile("" : "+v"(lf));
    const float *row = rows + f * nb;
    auto params_of = [&](int g) {
        const int mi = sch.mel_base[g] + (lf & ((LPF >> sch.part_log2[g]) - 1));
        return par + kMelParDwords * (mi < kMelMaxMels ? mi : kMelMaxMels);
    };
    // parameters of group 0; those of group g+1 are fetched while group g is reduced
    const float *pm = params_of(0);
    float K0 = pm[0], U0 = pm[1], DU = pm[2], D0 = pm[3], DD = pm[4];
#pragma unroll
    for (int g = 0; g < kMelMaxGroups; ++g) {
        if (g < sch.n_groups) {
            const int L = sch.trips[g], pl = sch.part_log2[g];
            const int msub = lf & ((LPF >> pl) - 1), h = lf >> (F == 2 ? 5 - pl : 6 - pl);
            const v2f ud0 = {U0, D0}, dud = {DU, -DD};
            unsigned addr = lds_addr(row + __float_as_int(K0) + h * L);
            if (g + 1 < kMelMaxGroups && g + 1 < sch.n_groups) {
                const float *pn = params_of(g + 1);
                K0 = pn[0]; U0 = pn[1]; DU = pn[2]; D0 = pn[3]; DD = pn[4];
            }
            float fi = (float)(h * L);
            v2f acc2 = {0.0f, 0.0f};
            // trips are multiples of 4 (host-padded): four bins per step, the next four already in
            // flight in the other register set (A/B ping-pong, two steps per iteration); a step's
            // prefetch may read <= 4 floats past the slice (inside the scratch, never used) and
            // everything is drained before the registers are reused
            v2f a0 = {0.0f, 0.0f}, a1 = a0, b0 = a0, b1 = a0;
            mel_lds_read2<0>(a0, addr);
            mel_lds_read2<2>(a1, addr);
            for (int i = 0; i < L; i += 8) {
                mel_lds_read2<4>(b0, addr);
                mel_lds_read2<6>(b1, addr);
                mel_lds_wait<2>(a0, a1);
                mel_step4(acc2, fi, a0, a1, ud0, dud);
                if (i + 4 < L) {
                    mel_lds_read2<8>(a0, addr);
                    mel_lds_read2<10>(a1, addr);
                    mel_lds_wait<2>(b0, b1);
                    mel_step4(acc2, fi, b0, b1, ud0, dud);
                }
                addr += 32;
            }
            mel_lds_wait<0>(a0, a1);
            mel_lds_wait<0>(b0, b1);
            float acc = acc2.x + acc2.y;
            // sum the K-slices: h sits in the top bits of the frame's lane index
            if (pl >= 1) {
                if (F == 1) {
                    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
                    acc = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                } else {
                    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
                    acc = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                }
            }
            if (pl >= 2) {
                if (F == 1) {
                    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
                    acc = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                } else {
                    acc += dpp_update<0x128, 0xF>(acc, acc);   // row_ror:8: lane ^ 8 within the row
                }
            }
            const int mel = sch.mel_base[g] + msub;
            if (h == 0 && mel < kMelMaxMels)
                stage[128 * f + mel] = acc;
        }
    }
}
// Mel values of the wave's F frames (stage[128 f + mel], written by mel_reduce) ->
// 10 log10(max(., floor)) -> rows of n_mels floats, two coalesced dword stores per row.
template <int F>
__device__ __forceinline__ void mel_store(const float *stage, const MelSched &sch, float log_floor, float *row_a,
                                          float *row_b, bool valid_b)
{
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ff = 0; ff < F; ++ff) {
        float *row = ff == 0 ? row_a : row_b;
        const float v0 = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(stage[128 * ff + lane], log_floor));
        const float v1 = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(stage[128 * ff + 64 + lane], log_floor));
        if (ff == 0 || valid_b) {
            if (lane < sch.n_mels)
                row[lane] = v0;
            if (lane + 64 < sch.n_mels)
                row[lane + 64] = v1;
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- kLogMelM / kMfccM: the mel contraction band-blocked on v_mfma_f32_4x4x1_16b_f32 (aa_internal.hpp) ----
// Per-lane residents of a wave: the float4 of weights of its (block, mel i) per step, the byte offset of the
// block's first 4-bin piece in a power row and the emission word (aa_internal.hpp).
struct MelMfma {
    v4f w[kMelMfmaSteps];
    unsigned off;
    unsigned emit;
};
__device__ __forceinline__ void mel_mfma_load(MelMfma &mm, const float *tab_wave)
{
    const int lane = threadIdx.x & 63;
    const v4f *w4 = reinterpret_cast<const v4f *>(tab_wave);
#pragma unroll
    for (int s = 0; s < kMelMfmaSteps; ++s)
        mm.w[s] = w4[s * 64 + lane];
    const unsigned *u = reinterpret_cast<const unsigned *>(tab_wave + kMelMfmaSteps * 64 * 4);
    mm.off = u[lane];
    mm.emit = u[64 + lane];
}
// The SIMD's instruction arbiter serves waves by priority, then age.  With equal priorities the
// workgroups that share a CU do not progress together: they finish one after the other in dispatch
// order (measured: 38 / 50 / 62 / 73 us for the four residents of every CU,
// tools/ablate_w1024.hip), the oldest wave of each SIMD runs nearly unimpeded and the others take
// the leftover issue slots.  Rotating the priority by (dispatch round + a step that advances four
// times per chunk) hands every resident the head of the queue in turn: -3 % on the sustained
// 1024 / 256 launch.  Once per chunk gives -2.6 %, eight times per chunk nothing (the branches cut
// the interleaved phases apart); a static youngest-first order, a progress-based priority and a
// dynamic chunk queue were measured too (tools/experiments/) and lost.
// step is uniform across the workgroup; only the issue order changes, never a result.
// Experiment switches of stft_wave1024_kernel (tools/ab_sustained.py builds one library per setting).
#ifndef AA_V_PAIR
#define AA_V_PAIR 1        // a wave takes two ADJACENT frames of the chunk; their rows leave as one 4104-byte piece
#endif
#ifndef AA_V_PKPOW
#define AA_V_PKPOW 1       // |X[k]|^2 and |X[512-k]|^2 as one packed pair (one v_pk_mul + one v_pk_fma for both)
#endif
#ifndef AA_V_PKPOW2048
#define AA_V_PKPOW2048 AA_V_PKPOW   // the same in stft_wave2048_kernel
#endif
#ifndef AA_V_DMA_LATE
#define AA_V_DMA_LATE 1    // magnitude-type and log-mel rows: the next chunk's LDS-DMA issued behind the frame reads of this chunk (r3: -0.8 %,
                           // 69.2 -> 68.65 us; log-mel -0.7 %); 2: every output kind
#endif
#ifndef AA_V_SPLIT
#define AA_V_SPLIT 1       // uneven chunk runs: the remainder goes to the workgroups dispatched first
#endif
// The persistent workgroup's run of chunks [c0, c1).  The chunks that do not divide evenly go to the
// workgroups dispatched FIRST: the residents of a CU finish in dispatch order even with the priority
// rotation (64 / 66 / 66 / 68 us for the four dispatch rounds, tools/wg_finish_order.py), so the late
// ones get the shorter runs: -1.3 % on the sustained 1024 / 256 launch against an even spread.
__device__ __forceinline__ void chunk_run(int64_t n_chunks, int &c0, int &c1, int grid, int b)
{
    if (AA_V_SPLIT) {
        const int q = (int)(n_chunks / grid), r = (int)(n_chunks % grid);
        c0 = b < r ? b * (q + 1) : r * (q + 1) + (b - r) * q;
        c1 = c0 + q + (b < r ? 1 : 0);
    } else {
        c0 = (int)((int64_t)b * n_chunks / grid);
        c1 = (int)((int64_t)(b + 1) * n_chunks / grid);
    }
}
__device__ __forceinline__ void chunk_run(int64_t n_chunks, int &c0, int &c1)
{
    chunk_run(n_chunks, c0, c1, (int)gridDim.x, (int)blockIdx.x);
}
// Several equal-length channels (streams) in one launch: the grid is n_streams equal groups of
// workgroups, group s works on x + s in_stride -> out + s out_stride exactly as a launch of its own
// would (same chunk runs, same results); only the tail of the launch is shared.
__device__ __forceinline__ void stream_of_block(int n_streams, int64_t in_stride, int64_t out_stride,
                                                int64_t &in_off, int64_t &out_off, int &grid, int &b)
{
    grid = (int)gridDim.x;
    b = (int)blockIdx.x;
    in_off = 0;
    out_off = 0;
    if (n_streams > 1) {
        grid /= n_streams;
        const int s = b / grid;
        b -= s * grid;
        in_off = s * in_stride;
        out_off = s * out_stride;
    }
}
#ifndef AA_PRIO_ROTATE
#define AA_PRIO_ROTATE 1      // 0: diagnostic builds without the rotation (tools/ab_sustained.py)
#endif
#ifndef AA_W1024_PRIO
#define AA_W1024_PRIO -1   // stft_wave1024_kernel: 0 the four-level rotation; abcd: priority by phase, the same in every workgroup - level
#endif                     // a from the chunk top, b behind frame a's first exchange, c behind frame b's, d from the epilogues on;
                           // -1 (shipped): 3310 for complex rows (-1.5 %: 91.9 -> 90.5 us), the rotation for every other kind (r3
                           // sweep: magnitudes level with 3310 / 2210 / 3320 and +1 % with 3210, log-mel +9 %, MFCC +5 % by phase)
__device__ __forceinline__ void rotate_wave_priority(int step)
{
    if (!AA_PRIO_ROTATE)
        return;
    switch (step & 3) {
    case 0: __builtin_amdgcn_s_setprio(3); break;
    case 1: __builtin_amdgcn_s_setprio(2); break;
    case 2: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(0); break;
    }
}

// The fused log-mel variant of the 2048 kernel keeps THREE workgroups per CU and has a third long phase
// (the mel gather): it rotates over three priority levels at three points per frame - chunk top, after the
// first exchange, in front of the gather: 113.5 -> 107.5 us per channel (A/B, alternating processes).  The
// magnitude variants lose 1.7 % with three levels and keep the four-level pattern; the 1024 mel kernels
// lose 5 % with a fifth rotation point and 5-7 % with three levels.
__device__ __forceinline__ void rotate_wave_priority3(int step)
{
    if (!AA_PRIO_ROTATE)
        return;
    switch (step % 3) {
    case 0: __builtin_amdgcn_s_setprio(2); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(0); break;
    }
}

template <int PATTERN>
__device__ __forceinline__ void w1024_prio(int point, int prio_step)
{
    if (PATTERN == 0) {
        rotate_wave_priority(prio_step + point);
        return;
    }
    const int lvl = point == 0 ? (PATTERN / 1000) % 10 : point == 1 ? (PATTERN / 100) % 10
                  : point == 2 ? (PATTERN / 10) % 10 : PATTERN % 10;
    if (lvl == 0) __builtin_amdgcn_s_setprio(0);
    if (lvl == 1) __builtin_amdgcn_s_setprio(1);
    if (lvl == 2) __builtin_amdgcn_s_setprio(2);
    if (lvl == 3) __builtin_amdgcn_s_setprio(3);
}

// ---------------------------------------------------- wave-per-frame, n_fft = 1024 --
// One wavefront transforms one frame: the 1024 windowed reals are read as 512 complex points
// z[m] (8 per lane), 512 = 8 x 8 x 8:
//   pass A  radix-8 over j (lane reads m = 64 j + lane from the staged samples), x W_512^(lane k1)
//   E1      registers <-> lanes through a wave-private LDS scratch: writer lane (a = l>>3,
//           b = l&7) stores Y[k1] at complex 66 b + 8 k1 + a (ds_write_b64), reader lane
//           (k1 = L&7, b = L>>3) loads 8 consecutive complex from 66 b + 8 k1 (4 x
//           ds_read_b128); the skew 66 keeps both sides bank-conflict free
//   pass B  radix-8 over a, x W_64^(b ka)
//   E2      registers <-> lane bits 3-5 in registers: v_mov_b32_dpp row_ror:8 with bank masks,
//           v_permlane16_swap, v_permlane32_swap (no select: v_cndmask_b32 costs 12-19 cycles)
//   pass C  radix-8 over b -> lane L holds Z[L + 64 kb]
//   split   X[k], X[512-k] from Z[k], conj(Z[512-k]); the partner half crosses the scratch
//   store   |X| row staged in the scratch at its 16-byte-aligned window, nontemporal dwordx4
// A persistent workgroup (4 waves) owns a contiguous run of 8-frame chunks; chunk c+1 arrives by
// LDS-DMA (global_load_lds_dwordx4) in the other half of a double buffer while chunk c is
// transformed; each wave interleaves its two frames of the chunk.  All complex arithmetic is
// packed f32 (pk_* helpers above).  (r4: the ablation masks and the measured-and-rejected variants of
// this kernel - software pipeline, halo copy inside LDS, interleaved chunk assignment, shared sample
// blocks, split frame waits, E2 through LDS - live in tools/experiments/r04_wave1024_removed_variants.diff,
// with their numbers; log-mel / MFCC rows run stft_wave1024_mel_kernel below, kLogMel here is the
// fallback for filterbanks its plan does not hold.)
template <int OUT, bool WIDE = false>
__global__ __launch_bounds__(kWave1024Threads)
void stft_wave1024_kernel(
    const float *__restrict__ x, float *__restrict__ out, const float *__restrict__ tab,
    int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg, int64_t n_chunks,
    float log_floor, const float *__restrict__ mel_par, MelSched sch, int n_streams, int64_t stream_in_stride,
    int64_t stream_out_stride)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int vgrid, vblock;
    {
        int64_t in_off, out_off;
        stream_of_block(n_streams, stream_in_stride, stream_out_stride, in_off, out_off, vgrid, vblock);
        x += in_off;
        out += out_off;
    }

    // This workgroup's run of chunks [c0, c1).  Everything per chunk below is 32-bit scalar
    // arithmetic on running pointers (a lone wave issues a scalar instruction only every few
    // cycles: the 64-bit multiplies of the first version cost ~450 cycles per frame).
    int c0, c1;
    chunk_run(n_chunks, c0, c1, vgrid, vblock);
    const int max_span = (frames_per_wg - 1) * hop + 1024;
    v2f *scr = reinterpret_cast<v2f *>(smem + 2 * max_span + wave * kWave1024ScratchDwords);
    constexpr int NB = 513;
    constexpr int kPrioPattern = AA_W1024_PRIO >= 0 ? AA_W1024_PRIO : OUT == kComplex ? 3310 : 0;
    constexpr bool MEL = OUT == kLogMel;
    constexpr bool PAIR = AA_V_PAIR && WIDE && !MEL && OUT != kComplex;
    // global stores a wave issues for its two rows of a full 8-frame chunk (every one is issued by
    // every wave: the lane masks never empty the wave) - the count the chunk-top wait leaves in flight
    constexpr int kRowStoresPerChunk = OUT == kComplex ? (WIDE ? 2 * 5 : 2 * 9)     // 4 dwordx4 + 1 dwordx2 | 9 dwordx2
                                       : !WIDE         ? 2 * 9                      // 9 dwords per row
                                       : PAIR          ? 5                          // 4 dwordx4 + 1 dwordx2 per row PAIR
                                                       : 2 * 3;                     // 2 dwordx4 + 1 dword per row
    const int COLS = (OUT == kComplex) ? 2 * NB : MEL ? sch.n_mels : NB;
    const int64_t first_frame = (int64_t)c0 * frames_per_wg;
    const float *src_chunk = x + (frame_begin + first_frame) * (int64_t)hop;   // samples of chunk c
    float *out_chunk = out + first_frame * COLS;                                // rows of chunk c
    int frames_left = (int)(n_frames - first_frame);                            // from chunk c on
    const int src_stride = frames_per_wg * hop, out_stride = frames_per_wg * COLS;
    const int frames_step = frames_per_wg;
    // fused mel: per-mel triangle parameters in LDS behind the scratch areas (visible after the
    // first chunk-top barrier)
    float *melpar = smem + 2 * max_span + 4 * kWave1024ScratchDwords;
    float *melstage = melpar + kMelParTableDwords + wave * kMelStageDwords;   // mel values of the wave's frames
    if (OUT == kLogMel)
        for (int i = tid; i < kMelParTableDwords; i += kWave1024Threads)
            melpar[i] = mel_par[i];

    // LDS-DMA of `nfc` frames' samples starting at src into buffer `which`
    // (wave w moves the 1 KiB pieces w, w+4, ..)
    auto stage = [&](const float *src, int nfc, int which) {
        const int span4 = ((nfc - 1) * hop + 1024) >> 2;
        float *dst = smem + which * max_span;
        for (int qb = wave * 64; qb < span4; qb += kWave1024Threads) {
            const int q = qb + lane;
            if (q < span4)
                lds_dma_16(src + 4 * q, dst + 4 * qb);
        }
    };

    if (c0 < c1)
        stage(src_chunk, frames_left < frames_per_wg ? frames_left : frames_per_wg, 0);

    // per-lane constants as packed pairs
    v2f win[8], twA[8], twB[8], tws[4];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        win[j] = v2f{tab[(2 * j) * 64 + lane], tab[(2 * j + 1) * 64 + lane]};
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        twA[k] = v2f{tab[(16 + 2 * (k - 1)) * 64 + lane], tab[(17 + 2 * (k - 1)) * 64 + lane]};
        twB[k] = v2f{tab[(30 + 2 * (k - 1)) * 64 + lane], tab[(31 + 2 * (k - 1)) * 64 + lane]};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        tws[j] = v2f{tab[(44 + 2 * j) * 64 + lane], tab[(45 + 2 * j) * 64 + lane]};

    // a lane of the head / tail store is active iff tail_lo <= sh <= tail_lo + tail_span
    // (lanes 0-3: sh <= lane; lanes 4-7: sh >= lane - 4; the others never)
    const int tail_lo = lane < 4 ? 0 : lane < 8 ? lane - 4 : 4;
    const unsigned tail_span = lane < 4 ? (unsigned)lane : lane < 8 ? (unsigned)(7 - lane) : 0u;
    const unsigned smem_addr = lds_addr(smem);
    v2f *e1_wr = scr + 66 * (lane & 7) + (lane >> 3);                       // + 8 k1
    const v4f *e1_rd = reinterpret_cast<const v4f *>(scr + 8 * (lane & 7) + 66 * (lane >> 3));

    int it = 0;
    v2f va[8], vbv[8];
    bool dma_due = false;
    for (int c = c0; c < c1; ++c, ++it) {
        const int cur = it & 1;
        const int nf = frames_left < frames_per_wg ? frames_left : frames_per_wg;
        const int prio_step = (int)(blockIdx.x >> 8) + 4 * it;   // dispatch round + quarter chunks
        // The DMA loads of this chunk were issued before the previous chunk's row stores, and
        // vmcnt retires in order: leave exactly those stores in flight (a wave issued them for its
        // two frames of a full chunk) instead of waiting out their HBM latency every chunk.
        if (it == 0 || frames_per_wg != 8)
            wait_vmcnt_le<0>();
        else if (MEL && sch.n_mels > 64)
            wait_vmcnt_le<4>();                                  // 2 dword stores per mel row, 2 rows
        else if (MEL)
            wait_vmcnt_le<2>();
        else
            wait_vmcnt_le<kRowStoresPerChunk>();
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        w1024_prio<kPrioPattern>(0, prio_step);
        // AA_V_DMA_LATE: the LDS-DMA of the next chunk is issued BEHIND this chunk's frame reads (its scalar address
        // loop then runs under their LDS round trip); the order of the vector-memory operations - DMA, then the
        // chunk's stores - is the same.  Waves without a frame (ragged last chunk only) issue it here.
        dma_due = false;
        if (c + 1 < c1) {
            if (AA_V_DMA_LATE && frames_per_wg == 8 && (PAIR ? 2 * wave : wave) < nf &&
                (PAIR || OUT == kLogMel || AA_V_DMA_LATE == 2))   // (complex rows +0.4 % with it: left at the top)
                dma_due = true;
            else {
                const int nleft = frames_left - frames_step;
                stage(src_chunk + src_stride, nleft < frames_per_wg ? nleft : frames_per_wg, cur ^ 1);
            }
        }
        const int row0 = (c * frames_per_wg) & 31;           // row index mod 32 (alignment shift)
        const unsigned buf_addr = smem_addr + 4u * (unsigned)(cur * max_span + 2 * lane);

        // ---- per-frame phases (lambdas so that two frames can be interleaved) ----------
        auto read_frame = [&](int t, v2f (&v)[8]) {
            const unsigned fr = buf_addr + 4u * (unsigned)(t * hop);
            v[0] = lds_read_b64<0>(fr); v[1] = lds_read_b64<512>(fr); v[2] = lds_read_b64<1024>(fr);
            v[3] = lds_read_b64<1536>(fr); v[4] = lds_read_b64<2048>(fr); v[5] = lds_read_b64<2560>(fr);
            v[6] = lds_read_b64<3072>(fr); v[7] = lds_read_b64<3584>(fr);
        };
        auto pass_a = [&](v2f (&v)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = v[j] * win[j];
            pk_dft8(v);
            pk_twiddle7(v, twA);
        };
        auto exchange1 = [&](v2f (&v)[8]) {           // E1 through LDS
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                e1_wr[8 * kk] = v[kk];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v4f qv = e1_rd[i];
                v[2 * i] = v2f{qv.x, qv.y};
                v[2 * i + 1] = v2f{qv.z, qv.w};
            }
            __builtin_amdgcn_wave_barrier();
        };
        auto pass_bc = [&](v2f (&v)[8]) {
            pk_dft8(v);
            pk_twiddle7(v, twB);
            hazard_fence(v);   // packed asm -> DPP / permlane read
            exchange_high(v);
            pk_dft8(v);
        };
        auto strip = [&](v2f (&v)[8], v2f (&prt)[4]) {   // partner half through the LDS strip
#pragma unroll
            for (int j = 4; j < 8; ++j)
                scr[lane + 64 * (j - 4)] = v[j];
            if (lane == 0)
                scr[256] = v[0];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j)
                prt[j] = scr[256 - (lane + 64 * j)];
            __builtin_amdgcn_wave_barrier();
        };
        auto epilogue = [&](v2f (&v)[8], v2f (&prt)[4], int t, bool valid, int slot) {
            float *row = out_chunk + t * COLS;
            // stage by stage over the four pairs, so that no instruction reads the result of the one
            // just before it (a dependent packed pair back to back costs a wait state)
            v2f xk[4], xm[4], se[4], so[4], st[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                se[j] = pk_add_conj(v[j], prt[j]);
                so[j] = pk_sub_conj(v[j], prt[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                st[j] = pk_cmul_lo(so[j], tws[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                st[j] = pk_cmul_hi(so[j], tws[j], st[j]);
            if (OUT == kComplex || !AA_V_PKPOW) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xk[j] = se[j] + st[j];
                    xm[j] = se[j] - st[j];                       // X[512-k] = conj(xm)
                }
            }
            if (OUT == kComplex) {
                if (WIDE) {
                    // Row (4104 bytes, 8-byte aligned: the shift is one complex for odd rows) through
                    // the scratch, so that it leaves as 4 aligned nontemporal dwordx4 stores + one
                    // dwordx2 (complex 0 of an odd row, complex 512 of an even one) instead of 9
                    // scattered 8-byte stores: scr[0] <-> the 16-byte-aligned address at or below the row.
                    const int shc = (row0 + t) & 1;
                    v2f *lrow = scr + shc;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        lrow[lane + 64 * j] = xk[j];
                        lrow[512 - lane - 64 * j] = v2f{xm[j].x, -xm[j].y};
                    }
                    if (lane == 0)
                        lrow[256] = v2f{2.0f * v[4].x, -2.0f * v[4].y};
                    __builtin_amdgcn_wave_barrier();
                    const v4f *lw4 = reinterpret_cast<const v4f *>(scr) + shc + lane;
                    const v4f q0 = lw4[0], q1 = lw4[64], q2 = lw4[128], q3 = lw4[192];
                    const v2f edge = scr[shc ? 1 : 512];
                    __builtin_amdgcn_wave_barrier();
                    if (valid) {
                        v4f *gw4 = reinterpret_cast<v4f *>(row - 2 * shc) + shc + lane;
                        row_store(q0, gw4);
                        row_store(q1, gw4 + 64);
                        row_store(q2, gw4 + 128);
                        row_store(q3, gw4 + 192);
                        if (lane == 0)
                            reinterpret_cast<v2f *>(row)[shc ? 0 : 512] = edge;
                    }
                } else if (valid) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kk = lane + 64 * j;
                        reinterpret_cast<v2f *>(row)[kk] = xk[j];
                        reinterpret_cast<v2f *>(row)[512 - kk] = v2f{xm[j].x, -xm[j].y};
                    }
                    if (lane == 0)
                        reinterpret_cast<v2f *>(row)[256] = v2f{2.0f * v[4].x, -2.0f * v[4].y};
                }
                return;
            }
            float mk[4], mm[4];
            if (AA_V_PKPOW) {
                // the real parts of X[k] and X[512-k] in one pair, the imaginary parts in another:
                // both powers from one packed multiply and one packed FMA
                v2f re[4], im[4], pw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    re[j] = pk_sum_diff_re(se[j], st[j]);
                    im[j] = pk_sum_diff_im(se[j], st[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    pw[j] = re[j] * re[j];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    pw[j] = im[j] * im[j] + pw[j];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mk[j] = bin_from_power<OUT>(pw[j].x, log_floor);
                    mm[j] = bin_from_power<OUT>(pw[j].y, log_floor);
                }
            } else {
                // squares of all eight bins, then the sums, then the roots / logs (same reason)
                v2f qk[4], qm[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    qk[j] = xk[j] * xk[j];
                    qm[j] = xm[j] * xm[j];
                }
                float pk[4], pm[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pk[j] = qk[j].x + qk[j].y;
                    pm[j] = qm[j].x + qm[j].y;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mk[j] = bin_from_power<OUT>(pk[j], log_floor);
                    mm[j] = bin_from_power<OUT>(pm[j], log_floor);
                }
            }
            const float m256 = bin_value<OUT>(2.0f * v[4].x, 2.0f * v[4].y, log_floor);
            if (MEL) {
                // power row -> scratch slot (consumed by mel_reduce, never stored)
                float *lrow = reinterpret_cast<float *>(scr) + slot * NB;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lrow[lane + 64 * j] = mk[j];
                    lrow[512 - lane - 64 * j] = mm[j];
                }
                if (lane == 0)
                    lrow[256] = m256;
                return;
            }
            if (PAIR && frames_per_wg == 8) {
                // The wave's two frames are ADJACENT (t = 2 wave, 2 wave + 1): their rows are one
                // 4104-byte piece whose start is 0 or 2 floats past a 16-byte boundary (513 = 1 mod 4,
                // the chunk's first row is a multiple of 8).  Both rows are staged (slot 0, slot 1),
                // then the piece leaves as 4 full nontemporal dwordx4 stores + one dwordx2 (floats
                // 1024-1025 of the window when sh2 == 0, floats 2-3 when sh2 == 2).
                const int sh2 = (row0 + t - slot) & 3;                   // 0 or 2
                float *lrow = reinterpret_cast<float *>(scr) + sh2 + slot * NB;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lrow[lane + 64 * j] = mk[j];
                    lrow[512 - lane - 64 * j] = mm[j];
                }
                if (lane == 0)
                    lrow[256] = m256;
                if (slot == 1) {
                    __builtin_amdgcn_wave_barrier();
                    const float *lw = reinterpret_cast<const float *>(scr);
                    float *gw = row - NB - sh2;                          // window of row a
                    if (valid) {
                        const v4f *lw4 = reinterpret_cast<const v4f *>(lw + 2 * sh2) + lane;
                        const v4f q0 = lw4[0], q1 = lw4[64], q2 = lw4[128], q3 = lw4[192];
                        const int eo = sh2 ? 2 : 1024;
                        const v2f edge = *reinterpret_cast<const v2f *>(lw + eo);
                        __builtin_amdgcn_wave_barrier();
                        v4f *gw4 = reinterpret_cast<v4f *>(gw + 2 * sh2) + lane;
                        row_store(q0, gw4);
                        row_store(q1, gw4 + 64);
                        row_store(q2, gw4 + 128);
                        row_store(q3, gw4 + 192);
                        if (lane == 0)
                            *reinterpret_cast<v2f *>(gw + eo) = edge;
                    } else {
                        // ragged last chunk: only row a exists - the single-row scheme below
                        v4f b0 = *reinterpret_cast<const v4f *>(lw + 4 + 4 * lane);
                        v4f b1 = *reinterpret_cast<const v4f *>(lw + 260 + 4 * (lane < 63 ? lane : 62));
                        const int hi = lane < 4 ? lane : 508 + lane;
                        const float hv = lw[lane < 8 ? hi : 0];
                        __builtin_amdgcn_wave_barrier();
                        row_store(b0, reinterpret_cast<v4f *>(gw + 4 + 4 * lane));
                        if (lane < 63)
                            row_store(b1, reinterpret_cast<v4f *>(gw + 260 + 4 * lane));
                        if ((unsigned)(sh2 - tail_lo) <= tail_span)
                            gw[hi] = hv;
                    }
                }
            } else if (WIDE) {
                // Row through LDS so it leaves as 16-byte-aligned nontemporal dwordx4 stores.
                // The row starts at float 513 r of a 16-byte-aligned buffer, so its aligned
                // window begins sh = r & 3 floats earlier; LDS float i <-> global float
                // 513 r - sh + i.
                const int sh = (row0 + t) & 3;
                float *lrow = reinterpret_cast<float *>(scr) + sh;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lrow[lane + 64 * j] = mk[j];
                    lrow[512 - lane - 64 * j] = mm[j];
                }
                if (lane == 0)
                    lrow[256] = m256;
                __builtin_amdgcn_wave_barrier();
                const float *lw = reinterpret_cast<const float *>(scr);
                float *gw = row - sh;
                v4f b0 = *reinterpret_cast<const v4f *>(lw + 4 + 4 * lane);
                v4f b1 = *reinterpret_cast<const v4f *>(lw + 260 + 4 * (lane < 63 ? lane : 62));
                const int hi = lane < 4 ? lane : 508 + lane;
                const float hv = lw[lane < 8 ? hi : 0];
                __builtin_amdgcn_wave_barrier();
                if (valid) {
                    row_store(b0, reinterpret_cast<v4f *>(gw + 4 + 4 * lane));
                    if (lane < 63)
                        row_store(b1, reinterpret_cast<v4f *>(gw + 260 + 4 * lane));
                    // head floats sh..3 (lanes 0-3) and tail floats 512..512+sh (lanes 4-7)
                    const bool ok = lane < 4 ? (lane >= sh) : (lane < 8 && lane - 4 <= sh);
                    if (ok)
                        gw[hi] = hv;
                }
            } else if (valid) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    row[lane + 64 * j] = mk[j];
                    row[512 - lane - 64 * j] = mm[j];
                }
                if (lane == 0)
                    row[256] = m256;
            }
        };

        if (frames_per_wg == 8) {
            // Two frames of this wave (t = wave and wave + 4) interleaved phase by phase: the
            // independent work of one frame fills the LDS round trips and VALU dependency stalls
            // of the other.  The single scratch area is safe: DS operations of a wave execute in
            // program order, and each exchange's reads are issued before the next one's writes.
            const int ta = PAIR ? 2 * wave : wave, tb = PAIR ? 2 * wave + 1 : wave + 4;
            if (ta < nf) {
                const bool vb = tb < nf;   // ragged last chunk: B computes on stale LDS, stores nothing
                v2f pa[4], pb[4];
                read_frame(ta, va);
                read_frame(tb, vbv);
                if (dma_due) {
                    const int nleft = frames_left - frames_step;
                    stage(src_chunk + src_stride, nleft < frames_per_wg ? nleft : frames_per_wg, cur ^ 1);
                    dma_due = false;
                }
                lds_wait_8(va[0], va[1], va[2], va[3], va[4], va[5], va[6], va[7]);
                lds_wait_8(vbv[0], vbv[1], vbv[2], vbv[3], vbv[4], vbv[5], vbv[6], vbv[7]);
                pass_a(va);
                exchange1(va);
                w1024_prio<kPrioPattern>(1, prio_step);
                pass_a(vbv);
                exchange1(vbv);
                w1024_prio<kPrioPattern>(2, prio_step);
                pass_bc(va);
                strip(va, pa);
                pass_bc(vbv);
                strip(vbv, pb);
                w1024_prio<kPrioPattern>(3, prio_step);
                epilogue(va, pa, ta, true, 0);
                epilogue(vbv, pb, tb, vb, 1);
                if (OUT == kLogMel) {
                    __builtin_amdgcn_wave_barrier();
                    mel_reduce<2>(reinterpret_cast<const float *>(scr), NB, melpar, sch, melstage);
                    mel_store<2>(melstage, sch, log_floor, out_chunk + ta * COLS, out_chunk + tb * COLS, vb);
                }
            }
        } else {
            for (int t = wave; t < nf; t += 4) {
                v2f v[8], prt[4];
                read_frame(t, v);
                lds_wait_8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
                pass_a(v);
                exchange1(v);
                pass_bc(v);
                strip(v, prt);
                epilogue(v, prt, t, true, 0);
                if (OUT == kLogMel) {
                    __builtin_amdgcn_wave_barrier();
                    mel_reduce<1>(reinterpret_cast<const float *>(scr), NB, melpar, sch, melstage);
                    mel_store<1>(melstage, sch, log_floor, out_chunk + t * COLS, nullptr, false);
                }
            }
        }
        src_chunk += src_stride;
        out_chunk += out_stride;
        frames_left -= frames_step;
    }
}

// ------------------------------- n_fft = 1024, log-mel / MFCC rows (kLogMelM / kMfccM, r4) --
// stft_wave1024_mel_kernel: the FFT of stft_wave1024_kernel (same passes, same scratch images, 8 frames per
// chunk, wave w owns frames w and w + 4 interleaved phase by phase) with the mel contraction band-blocked on
// v_mfma_f32_4x4x1_16b_f32 (MelMfma above, aa_internal.hpp).  The contraction needs the eight power rows of the
// chunk, i.e. a workgroup barrier (B1) behind the epilogues, and the next chunk's exchanges overwrite the rows,
// i.e. a second one (T, also the barrier that says "chunk c + 1 has landed"):
//     loop over chunks c:   frame reads, window, pass A, exchange 1, passes B / C, split, power rows of chunk c
//                           wait for this wave's LDS-DMA pieces of chunk c + 1;  barrier B1
//                           row reads | 32 MFMAs | partial sums through LDS | logs | row pieces of chunk c (kMfccM:
//                           log-mel rows into the 16-frame block)
//                           barrier T;  LDS-DMA of chunk c + 2 into the buffer chunk c's samples and partial sums used
//                           kMfccM, every second chunk: DCT-II of the block on v_mfma_f32_16x16x4_f32 (behind T: no
//                           barrier of its own)
// Vector-memory order per wave: .. store(c - 1) | DMA(c + 1) | .. | store(c) | DMA(c + 2): the wait in front of B1
// is a plain vmcnt(0) - everything it covers was issued most of a chunk ago (timing-only ablation without the
// DMA: same launch time).  Measured and rejected (tools/experiments/r04_mel_kernel_pipelined.diff): the front half of
// chunk c + 1 (frame reads, window, pass A - registers only) inside the phase between B1 and T, to fill its chain of
// latencies: 85.4 us against 77.2 (the exchanges' LDS round trip is then exposed at the chunk top, and the LDS-DMA has
// half a chunk instead of a whole one to land: 82.7 us without any DMA).
// LDS: [2][span] samples | 4 x kMelMfmaScratchDwords | kMfccM: the log-mel block [16][kMfccRowStride], the coefficient
// tile [16][n_mfcc] | a zero float4 | the partial sums when a sample buffer is too small for them (hops below 174).
#ifndef AA_MELP_PRIO
#define AA_MELP_PRIO 2103  // r4: 74.7 us against 78.8 with the rotation (log-mel), 84.8 against 91.0 (MFCC); 3210 75.5, 3103 / 3203 / 2102 within 0.5 % of 2103, 2113 / 2003 / 1003 lose 5 %
#endif
#ifndef AA_MELP_PRIO2048
#define AA_MELP_PRIO2048 2103   // stft_wave2048_mel_kernel: level at the chunk top, behind the exchanges, from the power row on, in the mel phase
#endif
// wave priority of stft_wave1024_mel_kernel at its four points (chunk top, behind the exchanges, power rows, mel phase):
// 1 = the four-level rotation of stft_wave1024_kernel; abcd (decimal digits) = level by phase, the same in every workgroup
__device__ __forceinline__ void mel_prio(int point, int prio_step)
{
    if (AA_MELP_PRIO == 0)
        return;
    if (AA_MELP_PRIO == 1) {
        rotate_wave_priority(prio_step + point);
        return;
    }
    const int lvl = point == 0 ? (AA_MELP_PRIO / 1000) % 10 : point == 1 ? (AA_MELP_PRIO / 100) % 10
                  : point == 2 ? (AA_MELP_PRIO / 10) % 10 : AA_MELP_PRIO % 10;
    if (lvl == 0) __builtin_amdgcn_s_setprio(0);
    if (lvl == 1) __builtin_amdgcn_s_setprio(1);
    if (lvl == 2) __builtin_amdgcn_s_setprio(2);
    if (lvl == 3) __builtin_amdgcn_s_setprio(3);
}
template <int OUT>
__global__ __launch_bounds__(kWave1024Threads) __attribute__((amdgpu_waves_per_eu(OUT == kMfccM ? 3 : 1)))
void stft_wave1024_mel_kernel(
    const float *__restrict__ x, float *__restrict__ out, const float *__restrict__ tab,
    int64_t frame_begin, int64_t n_frames, int hop, int64_t n_chunks,
    float log_floor, const float *__restrict__ mel_tab, int n_mels, const float *__restrict__ dct_tab,
    int n_mfcc, int n_streams, int64_t stream_in_stride, int64_t stream_out_stride)
{
    static_assert(OUT == kLogMelM || OUT == kMfccM, "mel kinds only");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool MFCC = OUT == kMfccM;
    constexpr int kScr = kMelMfmaScratchDwords, FPW = 8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int vgrid, vblock;
    {
        int64_t in_off, out_off;
        stream_of_block(n_streams, stream_in_stride, stream_out_stride, in_off, out_off, vgrid, vblock);
        x += in_off;
        out += out_off;
    }
    int c0, c1;
    chunk_run(n_chunks, c0, c1, vgrid, vblock);
    if (c0 >= c1)
        return;                                              // (workgroup-uniform)
    const int max_span = (FPW - 1) * hop + 1024;
    float *scr0 = smem + 2 * max_span;
    v2f *scr = reinterpret_cast<v2f *>(scr0 + wave * kScr);
    float *lm = scr0 + 4 * kScr;                             // kMfccM: log-mel rows of two chunks
    float *ctile = lm + (MFCC ? kMfccLdsDwords : 0);         // kMfccM: coefficients of a 16-frame block, [16][n_mfcc]
    float *tail = ctile + (MFCC ? kMfccOutTileDwords : 0);   // [zero float4][partial sums, if not in a sample buffer]
    const int COLS = MFCC ? n_mfcc : n_mels;
    const int64_t first_frame = (int64_t)c0 * FPW;
    const float *src_chunk = x + (frame_begin + first_frame) * (int64_t)hop;   // samples of chunk c
    float *out_chunk = out + first_frame * COLS;                                // rows of chunk c
    int frames_left = (int)(n_frames - first_frame);                            // from chunk c on
    const int src_stride = FPW * hop, out_stride = FPW * COLS;

    // LDS-DMA of `nfc` frames' samples starting at src into buffer `which` (wave w moves the 1 KiB pieces w, w + 4, ..)
    auto stage = [&](const float *src, int nfc, int which) {
        const int span4 = ((nfc - 1) * hop + 1024) >> 2;
        float *dst = smem + which * max_span;
        for (int qb = wave * 64; qb < span4; qb += kWave1024Threads) {
            const int q = qb + lane;
            if (q < span4)
                lds_dma_16(src + 4 * q, dst + 4 * qb);
        }
    };
    stage(src_chunk, frames_left < FPW ? frames_left : FPW, 0);

    // per-lane constants as packed pairs (rows of the plan's table as in stft_wave1024_kernel)
    v2f win[8], twA[8], twB[8], tws[4];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        win[j] = v2f{tab[(2 * j) * 64 + lane], tab[(2 * j + 1) * 64 + lane]};
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        twA[k] = v2f{tab[(16 + 2 * (k - 1)) * 64 + lane], tab[(17 + 2 * (k - 1)) * 64 + lane]};
        twB[k] = v2f{tab[(30 + 2 * (k - 1)) * 64 + lane], tab[(31 + 2 * (k - 1)) * 64 + lane]};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        tws[j] = v2f{tab[(44 + 2 * j) * 64 + lane], tab[(45 + 2 * j) * 64 + lane]};
    MelMfma mmf;
    mel_mfma_load(mmf, mel_tab + wave * kMelMfmaWaveDwords);
    if (MFCC)
        for (int i = tid; i < kMfccLdsDwords; i += kWave1024Threads)
            lm[i] = 0.0f;                    // columns past n_mels are read by the DCT (zero weights): keep them finite
    if (tid < 4)
        tail[tid] = 0.0f;                    // the zero slot of the slice sums
    const unsigned smem_addr = lds_addr(smem);
    v2f *e1_wr = scr + 66 * (lane & 7) + (lane >> 3);                       // + 8 k1
    const v4f *e1_rd = reinterpret_cast<const v4f *>(scr + 8 * (lane & 7) + 66 * (lane >> 3));

    // ---- pieces of the transform (see stft_wave1024_kernel for the maps) ----
    auto read_frames = [&](int buf, v2f (&a)[8], v2f (&b)[8]) {             // frames wave and wave + 4 of the buffer
        const unsigned fa = smem_addr + 4u * (unsigned)(buf * max_span + 2 * lane + wave * hop);
        const unsigned fb = fa + 16u * (unsigned)hop;
        a[0] = lds_read_b64<0>(fa); a[1] = lds_read_b64<512>(fa); a[2] = lds_read_b64<1024>(fa); a[3] = lds_read_b64<1536>(fa);
        a[4] = lds_read_b64<2048>(fa); a[5] = lds_read_b64<2560>(fa); a[6] = lds_read_b64<3072>(fa); a[7] = lds_read_b64<3584>(fa);
        b[0] = lds_read_b64<0>(fb); b[1] = lds_read_b64<512>(fb); b[2] = lds_read_b64<1024>(fb); b[3] = lds_read_b64<1536>(fb);
        b[4] = lds_read_b64<2048>(fb); b[5] = lds_read_b64<2560>(fb); b[6] = lds_read_b64<3072>(fb); b[7] = lds_read_b64<3584>(fb);
    };
    auto pass_a = [&](v2f (&v)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = v[j] * win[j];
        pk_dft8(v);
        pk_twiddle7(v, twA);
    };
    auto exchange1 = [&](v2f (&v)[8]) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            e1_wr[8 * kk] = v[kk];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v4f qv = e1_rd[i];
            v[2 * i] = v2f{qv.x, qv.y};
            v[2 * i + 1] = v2f{qv.z, qv.w};
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto pass_bc = [&](v2f (&v)[8]) {
        pk_dft8(v);
        pk_twiddle7(v, twB);
        hazard_fence(v);   // packed asm -> DPP / permlane read
        exchange_high(v);
        pk_dft8(v);
    };
    auto strip = [&](v2f (&v)[8], v2f (&prt)[4]) {   // partner half through the LDS strip
#pragma unroll
        for (int j = 4; j < 8; ++j)
            scr[lane + 64 * (j - 4)] = v[j];
        if (lane == 0)
            scr[256] = v[0];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j)
            prt[j] = scr[256 - (lane + 64 * j)];
        __builtin_amdgcn_wave_barrier();
    };
    auto power_row = [&](v2f (&v)[8], v2f (&prt)[4], int slot) {           // |X[k]|^2, k = 0..512 -> the wave's row `slot`
        v2f se[4], so[4], st[4], re[4], im[4], pw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            se[j] = pk_add_conj(v[j], prt[j]);
            so[j] = pk_sub_conj(v[j], prt[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            st[j] = pk_cmul_lo(so[j], tws[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            st[j] = pk_cmul_hi(so[j], tws[j], st[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            re[j] = pk_sum_diff_re(se[j], st[j]);
            im[j] = pk_sum_diff_im(se[j], st[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            pw[j] = re[j] * re[j];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            pw[j] = im[j] * im[j] + pw[j];
        const float m256 = bin_value<kPower>(2.0f * v[4].x, 2.0f * v[4].y, log_floor);
        float *lrow = reinterpret_cast<float *>(scr) + slot * kMelMfmaRowB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lrow[lane + 64 * j] = pw[j].x;
            lrow[512 - lane - 64 * j] = pw[j].y;
        }
        if (lane == 0)
            lrow[256] = m256;
        // the row's last float4 piece is read whole (bin 512 + three floats with zero weights): keep those finite
        // whatever another frame left in the scratch
        if (lane < 3)
            lrow[513 + lane] = 0.0f;
    };

    // kMfccM: the accumulators of the last DCT block go to the coefficient tile at the top of the next chunk (the MFMAs
    // drain meanwhile); behind that chunk's barrier B1 the whole workgroup copies the tile - the block's 16 rows are ONE
    // contiguous span of the output - with coalesced stores (four scattered dwords per lane before: -5 us per launch)
    v4f pend0 = {0.0f, 0.0f, 0.0f, 0.0f}, pend1 = pend0;
    bool pending = false;                                    // accumulators not yet in the tile
    float *tile_dst = nullptr;                               // where the tile's block starts in the output, and
    int tile_rows = 0;                                       // how many of its rows exist (0: nothing to copy)
    const int tile_stride = COLS | 1;                        // odd: the accumulator writes spread over the banks (two-way at worst)
    auto flush_dct = [&]() {
        if (MFCC && pending) {
            const v4f d = pend0 + pend1;
            // D layout of the 16 x 16 x 4 MFMA: lane = (4-row group lane / 16, frame j = lane % 16), register r = row in the group
            const int coef0 = 2 * (16 * (wave & 1) + 4 * (lane >> 4)) + (wave >> 1);
            float *t = ctile + (lane & 15) * tile_stride + coef0;
            if (coef0 + 0 < n_mfcc) t[0] = d.x;
            if (coef0 + 2 < n_mfcc) t[2] = d.y;
            if (coef0 + 4 < n_mfcc) t[4] = d.z;
            if (coef0 + 6 < n_mfcc) t[6] = d.w;
            pending = false;
        }
    };
    auto store_tile = [&]() {                                // (behind a workgroup barrier that follows flush_dct)
        if (MFCC && tile_rows > 0) {                        // wave w: rows 4 w .. 4 w + 3, one coalesced piece of n_mfcc floats each
#pragma unroll
            for (int r = 4 * wave; r < 4 * wave + 4; ++r)
                if (r < tile_rows && lane < COLS)
                    tile_dst[r * COLS + lane] = ctile[r * tile_stride + lane];
            tile_rows = 0;
        }
    };
    // kMfccM: the A operand of the DCT (the wave's 16 coefficients x 64 folded mel bands) stays in registers: fetched per
    // block from the plan's table it cost 4 us per launch (its wait also covers whatever vector-memory work is in flight)
    v4f a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0, a2 = a0, a3 = a0;
    if (MFCC) {
        const v4f *at = reinterpret_cast<const v4f *>(dct_tab) + (wave * 4) * 64 + lane;
        a0 = at[0]; a1 = at[64]; a2 = at[128]; a3 = at[192];
    }

    // ---- prologue: chunk c0 has landed ----
    wait_vmcnt_le<0>();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    if (c0 + 1 < c1) {
        const int nleft = frames_left - FPW;
        stage(src_chunk + src_stride, nleft < FPW ? nleft : FPW, 1);
    }

    int it = 0;
    for (int c = c0; c < c1; ++c, ++it) {
        const int cur = it & 1;
        const int nf = frames_left < FPW ? frames_left : FPW;
        const bool has_next = c + 1 < c1;
        const int prio_step = (int)(blockIdx.x >> 8) + 4 * it;
        mel_prio(0, prio_step);
        flush_dct();
        // ---- the transform of the wave's two frames, phase by phase, and their power rows ----
        {
            v2f va[8], vbv[8], pa[4], pb[4];
            read_frames(cur, va, vbv);      // (a wave without a frame transforms stale samples: nothing of it is stored)
            lds_wait_8(va[0], va[1], va[2], va[3], va[4], va[5], va[6], va[7]);
            lds_wait_8(vbv[0], vbv[1], vbv[2], vbv[3], vbv[4], vbv[5], vbv[6], vbv[7]);
            pass_a(va);
            exchange1(va);
            pass_a(vbv);
            exchange1(vbv);
            mel_prio(1, prio_step);
            pass_bc(va);
            strip(va, pa);
            pass_bc(vbv);
            strip(vbv, pb);
            mel_prio(2, prio_step);
            power_row(va, pa, 0);
            power_row(vbv, pb, 1);
        }
        wait_vmcnt_le<0>();                  // this wave's pieces of chunk c + 1 (and whatever was stored before them)
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();        // B1: the eight power rows are written, chunk c + 1 has landed
        mel_prio(3, prio_step);
        store_tile();                        // kMfccM: the block the DCT finished a chunk ago

        // ---- the mel contraction of chunk c with the front half of chunk c + 1 inside it ----
        int ln = lane;
        asm volatile("" : "+v"(ln));         // (opaque: per-lane addresses are not hoisted out of the chunk loop)
        const v4f *q = reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(scr0 + (ln & 3) * kScr) + mmf.off);
        v4f b0[kMelMfmaSteps], b1[kMelMfmaSteps];
#pragma unroll
        for (int s = 0; s < kMelMfmaSteps; ++s) {
            b0[s] = q[s];
            b1[s] = q[s + kMelMfmaRowB / 4];
        }
        v4f d0 = {0.0f, 0.0f, 0.0f, 0.0f}, d1 = d0;
#pragma unroll
        for (int s = 0; s < kMelMfmaSteps; ++s) {            // (steps past the plan's carry zero weights)
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(mmf.w[s].x, b0[s].x, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(mmf.w[s].x, b1[s].x, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(mmf.w[s].y, b0[s].y, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(mmf.w[s].y, b1[s].y, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(mmf.w[s].z, b0[s].z, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(mmf.w[s].z, b1[s].z, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(mmf.w[s].w, b0[s].w, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(mmf.w[s].w, b1[s].w, d1, 0, 0, 0);
        }
        // partial sums of (block, frame) -> LDS: this chunk's sample buffer is dead (every wave read its frames a chunk
        // ago; the next LDS-DMA into it is issued behind barrier T)
        float *part = (max_span >= kMelMfmaPartDwords ? smem + cur * max_span : tail + 4) + wave * kMelMfmaPartWaveDwords;
        v4f *p4 = reinterpret_cast<v4f *>(part);
        unsigned em = mmf.emit;
        asm volatile("" : "+v"(em));
        const int set1 = 64 + 4 * __builtin_amdgcn_readfirstlane((int)(em >> 14) & 3);   // second frame set (aa_internal.hpp)
        p4[ln] = d0;
        p4[set1 + ln] = d1;
        __builtin_amdgcn_wave_barrier();
        // lanes re-labelled (quad slot, frame f = lane % 8): the slices of the quad (consecutive blocks), four per round
        const int ns = (int)(em & 31u), blk0 = (int)((em >> 5) & 15u);
        const int hmax = __builtin_amdgcn_readfirstlane((int)(em >> 24) & 31);
        const v4f *src = p4 + ((ln & 4) ? set1 : 0) + 4 * blk0 + (ln & 3);      // set f / 4, lane 4 blk0 + f % 4
        const v4f *z4 = reinterpret_cast<const v4f *>(tail);
        v4f m4 = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int h0 = 0; h0 < hmax; h0 += 4) {
            const v4f t0 = *(h0 + 0 < ns ? src + 4 * (h0 + 0) : z4);
            const v4f t1 = *(h0 + 1 < ns ? src + 4 * (h0 + 1) : z4);
            const v4f t2 = *(h0 + 2 < ns ? src + 4 * (h0 + 2) : z4);
            const v4f t3 = *(h0 + 3 < ns ? src + 4 * (h0 + 3) : z4);
            m4 += (t0 + t1) + (t2 + t3);
        }
        if (ns != 0) {                                       // (quad slot, frame): the quad's four mels of one frame
            const int q4 = (int)((em >> 9) & 31u) * 4, f = ln & 7;
            v4f l4;
            l4.x = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(m4.x, log_floor));
            l4.y = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(m4.y, log_floor));
            l4.z = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(m4.z, log_floor));
            l4.w = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(m4.w, log_floor));
            if (!MFCC) {
                // rows of n_mels floats (a multiple of 4, 16-byte-aligned buffer): one dwordx4 per (quad, frame)
                // (a plain store: nontemporal costs this kernel 5 % - 80.3 us against 75.9 - where the 2048 kernel gains 2 %)
                if (f < nf)
                    *reinterpret_cast<v4f *>(out_chunk + __mul24(f, COLS) + q4) = l4;
            } else {
                // row 8 (chunk parity) + frame of the log-mel block (rows 0..7 were read by the DCT a chunk pair ago)
                // (columns swizzled by the row's bit 2 - adjacent quads swapped in rows 4..7 and 12..15: with the 136-float
                // stride both this write and the DCT's B-operand reads are conflict free on the hardware's lane groups)
                *reinterpret_cast<v4f *>(lm + (8 * (it & 1) + f) * kMfccRowStride + (q4 ^ (f & 4))) = l4;
            }
        }
        const bool dct_now = MFCC && ((it & 1) || !has_next);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();        // T: every wave is done with the rows, the partial sums and the sample buffer
        if (dct_now) {
            // ---- DCT-II of the block of (up to) 16 frames on the f32 matrix cores (see stft_wave1024_kernel): wave w owns
            // the 16 coefficients c = 2 (16 (w & 1) + i) + (w >> 1); the fold about the middle of the mel axis happens in the
            // B operand: u[m] = L[m] + L[M-1-m] for the even coefficients' tiles, v[m] = L[m] - L[M-1-m] for the odd ones
            // (M % 4 == 0: the mirrored float4 is aligned; m >= M / 2 carries zero weights and reads finite values)
            const float *rowp = lm + (ln & 15) * kMfccRowStride;
            const int mq = 4 * (ln >> 4);
            v4f bb[4];
#pragma unroll
            for (int S = 0; S < 4; ++S) {
                const int ma = 16 * S + mq;
                int mb = n_mels - 4 - ma;
                mb = mb < 0 ? ma : mb;
                const v4f a = *reinterpret_cast<const v4f *>(rowp + (ma ^ (ln & 4)));      // (the row's column swizzle)
                const v4f r = *reinterpret_cast<const v4f *>(rowp + (mb ^ (ln & 4)));
                bb[S] = (wave >> 1) ? v4f{a.x - r.w, a.y - r.z, a.z - r.y, a.w - r.x}
                                    : v4f{a.x + r.w, a.y + r.z, a.z + r.y, a.w + r.x};
            }
            v4f acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bb[0].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bb[1].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bb[0].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bb[1].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bb[0].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, bb[1].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bb[0].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, bb[1].w, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, bb[2].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3.x, bb[3].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, bb[2].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3.y, bb[3].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, bb[2].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3.z, bb[3].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, bb[2].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3.w, bb[3].w, acc1, 0, 0, 0);
            // rows 0..7 of a full block are the previous chunk's frames, rows 8..15 this chunk's (a lone last chunk: rows 0..7)
            pend0 = acc0;
            pend1 = acc1;
            pending = true;
            tile_dst = (it & 1) ? out_chunk - out_stride : out_chunk;
            tile_rows = (it & 1) ? 8 + nf : nf;
        }
        if (c + 2 < c1) {
            const int nleft = frames_left - 2 * FPW;
            stage(src_chunk + 2 * src_stride, nleft < FPW ? nleft : FPW, cur);
        }
        src_chunk += src_stride;
        out_chunk += out_stride;
        frames_left -= FPW;
    }
    if (MFCC) {
        flush_dct();
        __syncthreads();
        store_tile();
    }
}

// ---------------------------------------------------- wave-per-frame, n_fft = 2048 --
// The tuner geometry of the reference (window 2048, hop 512: src/audio_io/stft.rs:169-170).
// 1024 complex points per frame = one radix-2 split over two 512-point transforms that one
// wave runs interleaved with the 8 x 8 x 8 machinery of the 1024 kernel:
//   z_even[n] = (x[4n], x[4n+1]), z_odd[n] = (x[4n+2], x[4n+3])  - one ds_read_b128 per lane
//   E = FFT512(z_even), O = FFT512(z_odd)            (passes A / E1 / B / E2 / C, twice)
//   Z[k] = E[k] + W_1024^k O[k],  Z[k+512] = E[k] - W_1024^k O[k]          (in-lane)
//   X[k], X[1024-k] from Z[k], conj(Z[1024-k]) with -i W_2048^k; the upper half of Z crosses
//   the wave-private LDS strip (8 ds_write_b64 + 8 ds_read_b64 per lane)
// Persistence, row staging and store scheme as in stft_wave1024_kernel; staging by LDS-DMA: the two
// chunk buffers of that kernel for any hop, a ring with the DMA two chunks ahead (RING == 2, below)
// for the tuner geometry itself.  Table rows ([92][64]): 0..31 0.5*w[4(64j+lane)+c] (row 4j+c),
// 32..45 W_512^(lane k), 46..59 W_64^((lane>>3) k), 60..75 W_1024^(lane+64kb),
// 76..91 -i W_2048^(lane+64kb); behind them [8][64] float4 of the window and [4][64] float4 each of the combine
// and the split twiddles (two per float4), which stft_wave2048_mel_kernel streams per frame.
__device__ __forceinline__ void fft512_pass_a(v2f (&v)[8], const v2f (&twA)[8])
{
    pk_dft8(v);
#pragma unroll
    for (int k = 1; k < 8; ++k)
        v[k] = pk_cmul(v[k], twA[k]);
}
__device__ __forceinline__ void fft512_exchange1(v2f (&v)[8], v2f *e1_wr, const v4f *e1_rd)
{
#pragma unroll
    for (int k = 0; k < 8; ++k)
        e1_wr[8 * k] = v[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v4f qv = e1_rd[i];
        v[2 * i] = v2f{qv.x, qv.y};
        v[2 * i + 1] = v2f{qv.z, qv.w};
    }
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void fft512_pass_bc(v2f (&v)[8], const v2f (&twB)[8])
{
    pk_dft8(v);
#pragma unroll
    for (int k = 1; k < 8; ++k)
        v[k] = pk_cmul(v[k], twB[k]);
    hazard_fence(v);   // packed asm -> DPP / permlane read
    exchange_high(v);
    pk_dft8(v);
}

#ifndef AA_W2048_DMA_LATE
#define AA_W2048_DMA_LATE 1  // r3: -0.4 % (75.0 -> 74.7 us), same order of the vector-memory operations (DMA, then the chunk's stores)
#endif
#ifndef AA_WSMALL_PRIO
#define AA_WSMALL_PRIO 0     // stft_wave_small_kernel: 0 one rotation step per frame group; ab: level a from the group's top, b behind its exchange
#endif
#ifndef AA_W2048_PRIO
#define AA_W2048_PRIO 210    // magnitude-type rows: 0 the four-level rotation; abc: priority by phase, the same in every workgroup - level a
                             // from the chunk top, b behind the first exchanges, c from the split on.  r3: three descending levels
                             // (210, 310, 320, 321 alike) take 4 % off the 2048 / 512 magnitude launch (77.8 -> 74.6 us; 81.7 -> 78.9 with
                             // HBM-sourced samples); two levels do nothing; complex rows and the fused mel rows lose with it (6 % at
                             // log-mel) and keep their rotations; the 1024 kernel (four waves per SIMD) is level with its rotation.
#endif
// RING == 2 (the tuner geometry: hop 512, 4 frames per chunk; magnitude-type and complex rows): the samples are
// staged in a ring of 30 blocks of 256 floats instead of two chunk buffers - a chunk is blocks [8 i, 8 i + 14) of
// the workgroup's sample run, block b lives in slot b mod 30, one LDS-DMA instruction moves one block, a wave
// moves exactly two per full chunk, a frame's eight ds_read_b128 are eight consecutive blocks - with the LDS-DMA
// running TWO chunks ahead.  With four frames per chunk and three workgroups per CU a chunk lasts ~1.7 us - about
// one HBM round trip under load - so the one-chunk-ahead DMA of the two-buffer variant (RING == 0) arrives late
// whenever the samples come from HBM rather than the Infinity Cache.  (The four-workgroup variant with a 22-block
// ring and streamed constants, RING == 1 of r2, lost everywhere: tools/experiments/r04_removed_variants.diff.)
template <int OUT, bool WIDE, int RING = 0>
__global__ __launch_bounds__(kWave1024Threads) __attribute__((amdgpu_waves_per_eu(3, 3)))
void stft_wave2048_kernel(
    const float *__restrict__ x, float *__restrict__ out, const float *__restrict__ tab,
    int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg, int64_t n_chunks,
    float log_floor, const float *__restrict__ mel_par, MelSched sch, int n_streams, int64_t stream_in_stride,
    int64_t stream_out_stride)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int vgrid, vblock;
    {
        int64_t in_off, out_off;
        stream_of_block(n_streams, stream_in_stride, stream_out_stride, in_off, out_off, vgrid, vblock);
        x += in_off;
        out += out_off;
    }
    static_assert(RING == 0 || RING == 2, "two chunk buffers or the 30-block ring");
    constexpr int RB = 30;                           // ring blocks
    constexpr int DEPTH = 2;                         // chunks the ring's DMA runs ahead
    if (RING) {
        hop = 512;
        frames_per_wg = 4;
    }

    int c0, c1;
    chunk_run(n_chunks, c0, c1, vgrid, vblock);
    const int max_span = (frames_per_wg - 1) * hop + 2048;
    const int stage_dwords = RING ? RB * 256 : 2 * max_span;
    v2f *scr = reinterpret_cast<v2f *>(smem + stage_dwords + wave * kWave1024ScratchDwords);
    constexpr int NB = 1025;
    constexpr bool MEL = OUT == kLogMel;
    static_assert(!(MEL && RING), "the log-mel rows of the tuner geometry are stft_wave2048_mel_kernel's");
    constexpr bool kPhasePrio = AA_W2048_PRIO != 0 && (OUT == kMag || OUT == kPower || OUT == kLogPower);
    // global stores a wave issues for its ONE row of a full 4-frame chunk (see stft_wave1024_kernel)
    constexpr int kRowStoresPerChunk = OUT == kComplex ? (WIDE ? 9 : 17)      // 8 dwordx4 + 1 dwordx2 | 17 dwordx2
                                       : WIDE          ? 5                    // 4 dwordx4 + head / tail dword
                                                       : 17;                  // 17 dwords
    const int COLS = (OUT == kComplex) ? 2 * NB : MEL ? sch.n_mels : NB;
    float *melpar = smem + stage_dwords + 4 * kWave1024ScratchDwords;
    float *melstage = melpar + kMelParTableDwords + wave * kMelStageDwords;
    if (OUT == kLogMel)
        for (int i = tid; i < kMelParTableDwords; i += kWave1024Threads)
            melpar[i] = mel_par[i];

    const int64_t first_frame = (int64_t)c0 * frames_per_wg;
    const float *src_chunk = x + (frame_begin + first_frame) * (int64_t)hop;
    float *out_chunk = out + first_frame * COLS;
    int frames_left = (int)(n_frames - first_frame);
    const int src_stride = frames_per_wg * hop, out_stride = frames_per_wg * COLS;

    auto stage = [&](const float *src, int nfc, int which) {
        const int span4 = ((nfc - 1) * hop + 2048) >> 2;
        float *dst = smem + which * max_span;
        for (int qb = wave * 64; qb < span4; qb += kWave1024Threads) {
            const int q = qb + lane;
            if (q < span4)
                lds_dma_16(src + 4 * q, dst + 4 * qb);
        }
    };
    // RING: blocks [have, need_end) of the workgroup's sample run into their ring slots; wave w moves
    // blocks have + w, have + w + 4, ..  Returns the number of DMA instructions this wave issued.
    int have = 0;
    const float *src_wg = src_chunk + 4 * lane;
    auto ring_dma = [&](int need_end) {
        int issued = 0;
        for (int b = have + wave; b < need_end; b += 4, ++issued)
            lds_dma_16(src_wg + 256 * b, smem + 256 * (b % RB));
        if (need_end > have)
            have = need_end;
        return issued;
    };
    // last block + 1 of chunk ci (relative to c0) of this workgroup's run; `have` when there is no such chunk
    const int run_frames = (c1 - c0) * 4 < frames_left ? (c1 - c0) * 4 : frames_left;
    auto blocks_end = [&](int ci) {
        const int left = run_frames - 4 * ci;
        return left <= 0 ? have : 8 * ci + 2 * ((left < 4 ? left : 4) - 1) + 8;
    };
    if (c0 < c1) {
        if (RING) {
            for (int k = 0; k < DEPTH; ++k)
                ring_dma(blocks_end(k));
        } else
            stage(src_chunk, frames_left < frames_per_wg ? frames_left : frames_per_wg, 0);
    }

    v4f win[8];
    v2f twA[8], twB[8], wc[8], tws[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        win[j] = v4f{tab[(4 * j) * 64 + lane], tab[(4 * j + 1) * 64 + lane], tab[(4 * j + 2) * 64 + lane],
                     tab[(4 * j + 3) * 64 + lane]};
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        twA[k] = v2f{tab[(32 + 2 * (k - 1)) * 64 + lane], tab[(33 + 2 * (k - 1)) * 64 + lane]};
        twB[k] = v2f{tab[(46 + 2 * (k - 1)) * 64 + lane], tab[(47 + 2 * (k - 1)) * 64 + lane]};
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        wc[j] = v2f{tab[(60 + 2 * j) * 64 + lane], tab[(61 + 2 * j) * 64 + lane]};
        tws[j] = v2f{tab[(76 + 2 * j) * 64 + lane], tab[(77 + 2 * j) * 64 + lane]};
    }

    v2f *e1_wr = scr + 66 * (lane & 7) + (lane >> 3);
    const v4f *e1_rd = reinterpret_cast<const v4f *>(scr + 8 * (lane & 7) + 66 * (lane >> 3));

    int dma_prev = 0;                                   // RING: DMA instructions of the last round
    v4f raw[8];                                         // RING: the wave's frame as it comes from the ring
    for (int c = c0; c < c1; ++c) {
        const int cur = (c - c0) & 1;
        // counted wait as in stft_wave1024_kernel: with 4 frames per chunk every wave stored one
        // row in the previous (full) chunk, after issuing this chunk's DMA loads
        if (c == c0 || frames_per_wg != 4)
            wait_vmcnt_le<0>();
        else if (RING) {
            // the blocks of this chunk were issued two DMA rounds ago: behind them sit the stores of
            // two (full) chunks and the two DMA instructions of the last round
            if (c - c0 < 2 || dma_prev != 2)
                wait_vmcnt_le<0>();
            else
                wait_vmcnt_le<2 * kRowStoresPerChunk + 2>();
        } else if (MEL && sch.n_mels > 64)
            wait_vmcnt_le<2>();
        else if (MEL)
            wait_vmcnt_le<1>();
        else
            wait_vmcnt_le<kRowStoresPerChunk>();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        const int prio_step = (int)(blockIdx.x >> 8) + (MEL ? 3 : 2) * (c - c0);   // see rotate_wave_priority[3]
        if (kPhasePrio)
            __builtin_amdgcn_s_setprio(AA_W2048_PRIO >= 1000 ? (AA_W2048_PRIO / 1000) % 10 : (AA_W2048_PRIO / 100) % 10);
        else if (MEL)
            rotate_wave_priority3(prio_step);
        else
            rotate_wave_priority(prio_step);
        const int nf = frames_left < frames_per_wg ? frames_left : frames_per_wg;
        // AA_W2048_DMA_LATE (RING): the DMA of the chunk after next is issued BEHIND the frame reads of this chunk (its
        // scalar address loop then runs under their LDS round trip) instead of in front
        constexpr bool kDmaLate = AA_W2048_DMA_LATE && RING;
        bool dma_due = false;
        if (RING) {
            if (kDmaLate && wave < nf)
                dma_due = true;
            else
                dma_prev = ring_dma(blocks_end(c - c0 + DEPTH));
        } else if (c + 1 < c1) {
            const int nleft = frames_left - frames_per_wg;
            stage(src_chunk + src_stride, nleft < frames_per_wg ? nleft : frames_per_wg, cur ^ 1);
        }
        const int row0 = (c * frames_per_wg) & 31;
        const v4f *buf = reinterpret_cast<const v4f *>(smem + cur * max_span) + lane;

        for (int t = wave; t < nf; t += 4) {
            v2f e[8], o[8];
            if (RING) {
                // frame t starts at block 8 (c - c0) + 2 t; read j is block + j
                const int s0 = (8 * (c - c0) + 2 * t) % RB;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int sj = s0 + j < RB ? s0 + j : s0 + j - RB;
                    raw[j] = reinterpret_cast<const v4f *>(smem + 256 * sj)[lane];
                }
                if (kDmaLate && dma_due) {
                    dma_prev = ring_dma(blocks_end(c - c0 + DEPTH));
                    dma_due = false;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v4f sx = raw[j] * win[j];
                    e[j] = v2f{sx.x, sx.y};
                    o[j] = v2f{sx.z, sx.w};
                }
            } else {
                const v4f *fr = buf + ((t * hop) >> 2);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v4f sx = fr[64 * j] * win[j];
                    e[j] = v2f{sx.x, sx.y};
                    o[j] = v2f{sx.z, sx.w};
                }
            }
            fft512_pass_a(e, twA);
            fft512_exchange1(e, e1_wr, e1_rd);
            if (kPhasePrio && AA_W2048_PRIO >= 1000)
                __builtin_amdgcn_s_setprio((AA_W2048_PRIO / 100) % 10);
            fft512_pass_a(o, twA);
            fft512_exchange1(o, e1_wr, e1_rd);
            if (kPhasePrio)
                __builtin_amdgcn_s_setprio((AA_W2048_PRIO / 10) % 10);
            else if (MEL)
                rotate_wave_priority3(prio_step + 1);
            else
                rotate_wave_priority(prio_step + 1);
            fft512_pass_bc(e, twB);
            fft512_pass_bc(o, twB);
            // radix-2 combine: zl[kb] = Z[lane + 64 kb], zh[kb] = Z[512 + lane + 64 kb]
            v2f zl[8], zh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v2f tt = pk_cmul(o[j], wc[j]);
                zl[j] = e[j] + tt;
                zh[j] = e[j] - tt;
            }
            // strip[i] = Z[512 + i], i = 0..511; strip[512] = Z[1024] := Z[0]
#pragma unroll
            for (int j = 0; j < 8; ++j)
                scr[lane + 64 * j] = zh[j];
            if (lane == 0)
                scr[512] = zl[0];
            __builtin_amdgcn_wave_barrier();
            v2f prt[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                prt[j] = scr[512 - (lane + 64 * j)];
            __builtin_amdgcn_wave_barrier();
            if (kPhasePrio)
                __builtin_amdgcn_s_setprio(AA_W2048_PRIO % 10);

            float *row = out_chunk + t * COLS;
            // complex rows: xk / xm = X[k], conj(X[1024-k]); the other kinds: pw = (|X[k]|^2, |X[1024-k]|^2)
            // from the real and the imaginary parts of both bins side by side (see pk_sum_diff_re)
            v2f xk[8], xm[8], pw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v2f ee = pk_add_conj(zl[j], prt[j]), oo = pk_sub_conj(zl[j], prt[j]);
                v2f tt = pk_cmul(oo, tws[j]);
                if (OUT == kComplex || !AA_V_PKPOW2048) {
                    xk[j] = ee + tt;
                    xm[j] = ee - tt;                   // X[1024-k] = conj(xm)
                } else {
                    const v2f re = pk_sum_diff_re(ee, tt), im = pk_sum_diff_im(ee, tt);
                    pw[j] = im * im + re * re;
                }
            }
            const v2f z512 = zh[0];                    // lane 0: Z[512]; X[512] = 2 conj(Z[512])
            if (OUT == kComplex && WIDE) {
                // Row (8200 bytes, 8-byte aligned: odd rows start one complex past a 16-byte boundary)
                // through the scratch in two halves, so that it leaves as 8 aligned nontemporal
                // dwordx4 stores + one dwordx2 (complex 0 of an odd row, complex 1024 of an even one).
                // scr[0] <-> the aligned address at or below complex 0 (first half) / complex 512.
                const int shc = (row0 + t) & 1;
                const v2f x512 = v2f{2.0f * z512.x, -2.0f * z512.y};
                v2f *lrow = scr + shc;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    lrow[lane + 64 * j] = xk[j];
                if (lane == 0)
                    lrow[512] = x512;
                __builtin_amdgcn_wave_barrier();
                const v4f *lw4 = reinterpret_cast<const v4f *>(scr) + shc + lane;
                v4f *gw4 = reinterpret_cast<v4f *>(row - 2 * shc) + shc + lane;
                {
                    const v4f q0 = lw4[0], q1 = lw4[64], q2 = lw4[128], q3 = lw4[192];
                    const v2f x0 = scr[1];                         // complex 0 of an odd row
                    __builtin_amdgcn_wave_barrier();
                    row_store(q0, gw4);
                    row_store(q1, gw4 + 64);
                    row_store(q2, gw4 + 128);
                    row_store(q3, gw4 + 192);
                    if (lane == 0 && shc)
                        reinterpret_cast<v2f *>(row)[0] = x0;
                }
                // second half: complex k (512..1024) at scr[k - 512 + shc]
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    lrow[512 - lane - 64 * j] = v2f{xm[j].x, -xm[j].y};
                if (lane == 0)
                    lrow[0] = x512;
                __builtin_amdgcn_wave_barrier();
                {
                    const v4f q0 = lw4[0], q1 = lw4[64], q2 = lw4[128], q3 = lw4[192];
                    const v2f x1024 = scr[512];                    // complex 1024 of an even row
                    __builtin_amdgcn_wave_barrier();
                    row_store(q0, gw4 + 256);
                    row_store(q1, gw4 + 320);
                    row_store(q2, gw4 + 384);
                    row_store(q3, gw4 + 448);
                    if (lane == 0 && !shc)
                        reinterpret_cast<v2f *>(row)[1024] = x1024;
                }
            } else if (OUT == kComplex) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = lane + 64 * j;
                    reinterpret_cast<v2f *>(row)[k] = xk[j];
                    reinterpret_cast<v2f *>(row)[1024 - k] = v2f{xm[j].x, -xm[j].y};
                }
                if (lane == 0)
                    reinterpret_cast<v2f *>(row)[512] = v2f{2.0f * z512.x, -2.0f * z512.y};
            } else {
                float mk[8], mm[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (AA_V_PKPOW2048) {
                        mk[j] = bin_from_power<OUT>(pw[j].x, log_floor);
                        mm[j] = bin_from_power<OUT>(pw[j].y, log_floor);
                    } else {
                        mk[j] = bin_value<OUT>(xk[j].x, xk[j].y, log_floor);
                        mm[j] = bin_value<OUT>(xm[j].x, xm[j].y, log_floor);
                    }
                }
                const float m512 = bin_value<OUT>(2.0f * z512.x, 2.0f * z512.y, log_floor);
                if (MEL) {
                    float *lrow = reinterpret_cast<float *>(scr);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        lrow[lane + 64 * j] = mk[j];
                        lrow[1024 - lane - 64 * j] = mm[j];
                    }
                    if (lane == 0)
                        lrow[512] = m512;
                    __builtin_amdgcn_wave_barrier();
                    rotate_wave_priority3(prio_step + 2);
                    mel_reduce<1>(lrow, NB, melpar, sch, melstage);
                    mel_store<1>(melstage, sch, log_floor, row, nullptr, false);
                } else if (WIDE) {
                    const int sh = (row0 + t) & 3;     // 1025 = 1 (mod 4): same shift rule as 513
                    float *lrow = reinterpret_cast<float *>(scr) + sh;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        lrow[lane + 64 * j] = mk[j];
                        lrow[1024 - lane - 64 * j] = mm[j];
                    }
                    if (lane == 0)
                        lrow[512] = m512;
                    __builtin_amdgcn_wave_barrier();
                    const float *lw = reinterpret_cast<const float *>(scr);
                    float *gw = row - sh;
                    // window floats 4..1023 as 4 dwordx4 pieces (the last one 63 lanes), head
                    // floats sh..3 and tail floats 1024..1024+sh by single dwords
#pragma unroll
                    for (int pce = 0; pce < 4; ++pce) {
                        const int ls = (pce == 3 && lane == 63) ? 62 : lane;   // keep the last read in range
                        v4f bb = *reinterpret_cast<const v4f *>(lw + 4 + 256 * pce + 4 * ls);
                        if (pce < 3 || lane < 63)
                            row_store(bb, reinterpret_cast<v4f *>(gw + 4 + 256 * pce + 4 * lane));
                    }
                    const int hi = lane < 4 ? lane : 1020 + lane;
                    const float hv = lw[lane < 8 ? hi : 0];
                    const bool ok = lane < 4 ? (lane >= sh) : (lane < 8 && lane - 4 <= sh);
                    if (ok)
                        gw[hi] = hv;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        row[lane + 64 * j] = mk[j];
                        row[1024 - lane - 64 * j] = mm[j];
                    }
                    if (lane == 0)
                        row[512] = m512;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        src_chunk += src_stride;
        out_chunk += out_stride;
        frames_left -= frames_per_wg;
    }
}

// ------------------------ n_fft = 2048 at hop 512, log-mel rows (kLogMelM; BASELINE configs[2]) --
// stft_wave2048_mel_kernel: the transform and the ring staging of stft_wave2048_kernel<.., RING == 2> (one wave per
// frame, four frames per chunk, LDS-DMA two chunks ahead) with the mel contraction of stft_wave1024_mel_kernel:
// behind a workgroup barrier (B1) the four power rows of the chunk are visible and every wave contracts a quarter
// of the filterbank with all four frames on v_mfma_f32_4x4x1_16b_f32 (block = quad of mels x the chunk's 4 frames).
// Differences to the 1024 kernel: one frame set (4 frames = the 4 columns of a block), 8 emission slots of 4 frames x 2 half quads,
// up to kMelMfmaSteps2048 steps with the weights RESIDENT (28 registers: the combine and split twiddles make room for
// them, see below), partial sums in an LDS area of their own (the ring has no dead space).  Vector-memory order per wave:
// .. twiddle fetches of frame c | DMA(c + 2) | store(c) | fetches of frame c + 1 ..: every fetch is waited for (vmcnt retires in
// order) before the next DMA round goes out; the chunk-top wait leaves that round and the row store in flight.
// LDS: 30 blocks of 256 samples | 4 x kMelMfmaScratchDwords | zero float4 | [4][64] float4 of partial sums.
template <int OUT>
__global__ __launch_bounds__(kWave1024Threads) __attribute__((amdgpu_waves_per_eu(3, 3)))
void stft_wave2048_mel_kernel(
    const float *__restrict__ x, float *__restrict__ out, const float *__restrict__ tab,
    int64_t frame_begin, int64_t n_frames, int64_t n_chunks, float log_floor,
    const float *__restrict__ mel_tab, int n_mels, int n_steps, int n_streams, int64_t stream_in_stride,
    int64_t stream_out_stride)
{
    static_assert(OUT == kLogMelM, "log-mel rows only");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RB = 30, DEPTH = 2, HOP = 512, FPW = 4, kScr = kMelMfmaScratchDwords, SMAX = kMelMfmaSteps2048;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int vgrid, vblock;
    {
        int64_t in_off, out_off;
        stream_of_block(n_streams, stream_in_stride, stream_out_stride, in_off, out_off, vgrid, vblock);
        x += in_off;
        out += out_off;
    }
    int c0, c1;
    chunk_run(n_chunks, c0, c1, vgrid, vblock);
    if (c0 >= c1)
        return;                                              // (workgroup-uniform)
    float *scr0 = smem + RB * 256;
    v2f *scr = reinterpret_cast<v2f *>(scr0 + wave * kScr);
    float *tail = scr0 + 4 * kScr;                           // [zero float4][partial sums [4][64] float4]
    const int COLS = n_mels;
    const int64_t first_frame = (int64_t)c0 * FPW;
    const float *src_chunk = x + (frame_begin + first_frame) * (int64_t)HOP;   // the workgroup's sample run (ring blocks count from here)
    float *out_chunk = out + first_frame * COLS;
    int frames_left = (int)(n_frames - first_frame);
    const int out_stride = FPW * COLS;

    // blocks [have, need_end) of the workgroup's sample run into their ring slots; wave w moves blocks have + w,
    // have + w + 4, ..  Returns the number of DMA instructions this wave issued.
    int have = 0;
    // (per-lane addresses from opaque copies of the lane index: as loop invariants they would hold registers the
    // resident weights need)
    auto ring_dma = [&](int need_end) {
        int issued = 0;
        int ld = lane;
        asm volatile("" : "+v"(ld));
        const float *src_wg = src_chunk + 4 * ld;
        for (int b = have + wave; b < need_end; b += 4, ++issued)
            lds_dma_16(src_wg + 256 * b, smem + 256 * (b % RB));
        if (need_end > have)
            have = need_end;
        return issued;
    };
    const int run_frames = (c1 - c0) * 4 < frames_left ? (c1 - c0) * 4 : frames_left;
    auto blocks_end = [&](int ci) {
        const int left = run_frames - 4 * ci;
        return left <= 0 ? have : 8 * ci + 2 * ((left < 4 ? left : 4) - 1) + 8;
    };
    for (int k = 0; k < DEPTH; ++k)
        ring_dma(blocks_end(k));

    // resident per-lane constants: the window, the pass-A / pass-B twiddles and the mel weights of this wave's quarter of
    // the filterbank; the combine and split twiddles (2 x 16 registers) are fetched per frame from the plan's float4
    // tables (the same 8 KB for every wave of the chip: L1 resident) - with them resident the weights do not fit 168
    // registers, and a per-wave weight table fetched per chunk instead costs 18 us per launch (28 KB per workgroup and
    // chunk out of the L2)
    v4f win[8];
    v2f twA[8], twB[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        win[j] = v4f{tab[(4 * j) * 64 + lane], tab[(4 * j + 1) * 64 + lane], tab[(4 * j + 2) * 64 + lane],
                     tab[(4 * j + 3) * 64 + lane]};
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        twA[k] = v2f{tab[(32 + 2 * (k - 1)) * 64 + lane], tab[(33 + 2 * (k - 1)) * 64 + lane]};
        twB[k] = v2f{tab[(46 + 2 * (k - 1)) * 64 + lane], tab[(47 + 2 * (k - 1)) * 64 + lane]};
    }
    const float *ct4 = tab + kWave2048Rows * 64 + 8 * 64 * 4;     // [4][64] float4 of combine twiddle pairs, [4][64] of split pairs
    const float *wtab = mel_tab + wave * kMelMfmaWaveDwords2048;
    v4f w[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; ++s)
        w[s] = reinterpret_cast<const v4f *>(wtab)[64 * s + lane];
    const unsigned *words = reinterpret_cast<const unsigned *>(wtab + SMAX * 64 * 4);
    if (tid < 4)
        tail[tid] = 0.0f;                    // the zero slot of the slice sums (visible behind the first barrier)

    int dma_prev = 0;                        // DMA instructions of the last round
    for (int c = c0; c < c1; ++c) {
        // the blocks of this chunk were issued two DMA rounds ago and have landed (the fetch waits of the previous chunk's
        // frame cover them: vmcnt retires in order); behind them sit the two DMA instructions of the last round and the row
        // store of the previous chunk, which stay in flight (fewer DMA instructions: a plain wait)
        if (c - c0 < 2 || dma_prev != 2)
            wait_vmcnt_le<0>();
        else
            wait_vmcnt_le<1 + 2>();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();        // T: chunk c has landed; every wave is done with the rows of chunk c - 1
        __builtin_amdgcn_s_setprio((AA_MELP_PRIO2048 / 1000) % 10);
        const int nf = frames_left < FPW ? frames_left : FPW;
        if (wave < nf) {
            const int t = wave;
            int le = lane;
            asm volatile("" : "+v"(le));
            v2f *e1_wr = scr + 66 * (le & 7) + (le >> 3);
            const v4f *e1_rd = reinterpret_cast<const v4f *>(scr + 8 * (le & 7) + 66 * (le >> 3));
            v2f e[8], o[8];
            {
                // frame t starts at block 8 (c - c0) + 2 t; read j is block + j
                const int s0 = (8 * (c - c0) + 2 * t) % RB;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int sj = s0 + j < RB ? s0 + j : s0 + j - RB;
                    const v4f sx = reinterpret_cast<const v4f *>(smem + 256 * sj)[lane] * win[j];
                    e[j] = v2f{sx.x, sx.y};
                    o[j] = v2f{sx.z, sx.w};
                }
            }
            fft512_pass_a(e, twA);
            fft512_exchange1(e, e1_wr, e1_rd);
            fft512_pass_a(o, twA);
            fft512_exchange1(o, e1_wr, e1_rd);
            __builtin_amdgcn_s_setprio((AA_MELP_PRIO2048 / 100) % 10);
            fft512_pass_bc(e, twB);
            int lt = lane;
            asm volatile("" : "+v"(lt));     // (opaque: the table addresses are not hoisted out of the chunk loop)
            v2f wc[8], tws[8];
            {
                // combine twiddles of this frame: in flight during passes B and C of the odd half
                const v4f *cp = reinterpret_cast<const v4f *>(ct4) + lt;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v4f q = cp[64 * j];
                    wc[2 * j] = v2f{q.x, q.y};
                    wc[2 * j + 1] = v2f{q.z, q.w};
                }
            }
            fft512_pass_bc(o, twB);
            // radix-2 combine: zl[kb] = Z[lane + 64 kb], zh[kb] = Z[512 + lane + 64 kb]
            v2f zl[8], zh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v2f tt = pk_cmul(o[j], wc[j]);
                zl[j] = e[j] + tt;
                zh[j] = e[j] - tt;
            }
            {
                // split twiddles: in flight during the strip round trip
                asm volatile("" : "+v"(lt));
                const v4f *sp = reinterpret_cast<const v4f *>(ct4) + 4 * 64 + lt;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v4f q = sp[64 * j];
                    tws[2 * j] = v2f{q.x, q.y};
                    tws[2 * j + 1] = v2f{q.z, q.w};
                }
            }
            // strip[i] = Z[512 + i], i = 0..511; strip[512] = Z[1024] := Z[0]
#pragma unroll
            for (int j = 0; j < 8; ++j)
                scr[lane + 64 * j] = zh[j];
            if (lane == 0)
                scr[512] = zl[0];
            __builtin_amdgcn_wave_barrier();
            v2f prt[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                prt[j] = scr[512 - (lane + 64 * j)];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_setprio((AA_MELP_PRIO2048 / 10) % 10);
            v2f pw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v2f ee = pk_add_conj(zl[j], prt[j]), oo = pk_sub_conj(zl[j], prt[j]);
                v2f tt = pk_cmul(oo, tws[j]);
                const v2f re = pk_sum_diff_re(ee, tt), im = pk_sum_diff_im(ee, tt);
                pw[j] = im * im + re * re;
            }
            const v2f z512 = zh[0];                    // lane 0: Z[512]; X[512] = 2 conj(Z[512])
            const float m512 = bin_value<kPower>(2.0f * z512.x, 2.0f * z512.y, log_floor);
            float *lrow = reinterpret_cast<float *>(scr);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lrow[lane + 64 * j] = pw[j].x;
                lrow[1024 - lane - 64 * j] = pw[j].y;
            }
            if (lane == 0)
                lrow[512] = m512;
            // the row's last float4 piece is read whole (bin 1024 + three floats with zero weights): keep those finite
            if (lane < 3)
                lrow[1025 + lane] = 0.0f;
        }
        // ---- the mel contraction of the chunk's four frames ----
        int ln = lane;
        asm volatile("" : "+v"(ln));         // (opaque: per-lane addresses are not hoisted out of the chunk loop)
        unsigned m_off = words[ln], m_emit = words[64 + ln];         // (two L1 hits per chunk instead of two registers)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(m_off), "+v"(m_emit)::"memory");
        // The LDS-DMA of the chunk after next goes out HERE, behind every fetch of this chunk (twiddles, the two words above):
        // vmcnt retires in order, so a wait for a fetch issued behind a DMA would wait for the DMA's HBM round trip
        dma_prev = ring_dma(blocks_end(c - c0 + DEPTH));
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();        // B1: the four power rows are written
        __builtin_amdgcn_s_setprio(AA_MELP_PRIO2048 % 10);
        const v4f *q = reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(scr0 + (ln & 3) * kScr) + m_off);
        v4f b[SMAX];
#pragma unroll
        for (int s = 0; s < SMAX; ++s)
            b[s] = q[s];
        v4f d0 = {0.0f, 0.0f, 0.0f, 0.0f}, d1 = d0;
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {     // (a step past n_steps has zero weights: it is skipped, wave-uniform)
            if (s < n_steps) {
                d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[s].x, b[s].x, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[s].y, b[s].y, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[s].z, b[s].z, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[s].w, b[s].w, d1, 0, 0, 0);
            }
        }
        // partial sums of (block, frame): lane = (block, frame j)
        v4f *p4 = reinterpret_cast<v4f *>(tail + 4) + wave * 64;
        p4[ln] = d0 + d1;
        __builtin_amdgcn_wave_barrier();
        // lanes re-labelled (quad slot = lane / 8, frame f = (lane / 2) % 4, half = lane % 2): each lane adds the slices
        // (consecutive blocks) of two of its quad's mels - all 64 lanes busy for 8 quads, two logs per lane
        unsigned em = m_emit;
        asm volatile("" : "+v"(em));
        const int ns = (int)(em & 31u), blk0 = (int)((em >> 5) & 15u);
        const int hmax = __builtin_amdgcn_readfirstlane((int)(em >> 24) & 31);
        const int f = (ln >> 1) & 3;
        const v2f *src = reinterpret_cast<const v2f *>(p4 + 4 * blk0 + f) + (ln & 1);
        const v2f *z2 = reinterpret_cast<const v2f *>(tail);
        v2f m2 = {0.0f, 0.0f};
        for (int h0 = 0; h0 < hmax; h0 += 4) {
            const v2f t0 = *(h0 + 0 < ns ? src + 8 * (h0 + 0) : z2);
            const v2f t1 = *(h0 + 1 < ns ? src + 8 * (h0 + 1) : z2);
            const v2f t2 = *(h0 + 2 < ns ? src + 8 * (h0 + 2) : z2);
            const v2f t3 = *(h0 + 3 < ns ? src + 8 * (h0 + 3) : z2);
            m2 += (t0 + t1) + (t2 + t3);
        }
        if (ns != 0) {
            const int q2 = (int)((em >> 9) & 31u) * 4 + 2 * (ln & 1);
            v2f l2;
            l2.x = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(m2.x, log_floor));
            l2.y = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(m2.y, log_floor));
            // rows of n_mels floats (a multiple of 4, 16-byte-aligned buffer): one dwordx2 per (quad half, frame).  (Measured
            // and rejected: the pairs parked in LDS and stored half a chunk later as one coalesced row per wave, so that no
            // fetch wait of the next frame covers a fresh store - 85.0 us against 82.6; whatever the store's position, a
            // timing-only build without it runs at 75.5.)
            if (f < nf)                                      // (nontemporal: 81.6 us against 83.0)
                __builtin_nontemporal_store(l2, reinterpret_cast<v2f *>(out_chunk + __mul24(f, COLS) + q2));
        }
        out_chunk += out_stride;
        frames_left -= FPW;
    }
}

// -------------------------------- several frames per wave, n_fft = 128, 256 and 512 --
// 256/64 is the onset geometry of the reference (src/analysis/onset.rs:122-123); 512 shares the
// structure (and 128, with no DPP stage at all).  M = n_fft/2 complex points per frame on LPF = M/8
// lanes (8 / 16 / 32) with 8 points per lane, so a wave transforms FPW = 64/LPF frames (8 / 4 / 2):
//   z[m], m = LPF j + q:  lane holds q, register j (0..7)
//   pass A   radix-8 over j in registers, x W_M^(q k1)                    -> Y'[q][k1]
//   stages   the LPF-point transform over q = q' + 8 c: log2(LPF) - 3 radix-2 DIF butterflies
//            (1 / 2) across lane bits 0 (and 1) by DPP quad_perm - the lane map puts the partners
//            in the same quad - each followed by its twiddle on the odd half
//   E        registers <-> lanes through the wave's LDS scratch, the 8 x 8 transposes of
//            stft_wave1024_kernel's E1 (same addresses, conflict free): q' <-> k1
//   pass C   radix-8 over q' in registers -> lane l (0..LPF-1 of its frame) holds Z[l + LPF mu]
//   split    Z of the FPW frames goes to an LDS strip (8 x ds_write_b64 per lane); the (k, M-k)
//            pairs (260 / 258) are spread over the 64 lanes, 5 per lane, each pair yields X[k]
//            and X[M-k]; magnitudes land in a dense FPW-row LDS image
//   store    256: 4 rows x 129 floats = 129 float4 (group starts are 16-byte aligned): 3
//            nontemporal dwordx4 stores per FOUR frames; 512: 2 rows x 257 floats = 257 float2.
// Lane map before E: lane = 8 q' + gamma, gamma = (frame, c) with c = the DIF output residue
// (kappa = (LPF/8) mu + c) whose bits, reversed, are the high bits of q; after E: lane = LPF
// frame + l.  (First 256 version: all four radix-2 stages of the 16-point transform as DPP
// butterflies - 55 % more VALU instructions per sample than the 1024 kernel.)
// Table rows ([50][64]): 0..15 0.5*w[2(LPF j+q)+c] (row 2j+c), 16..29 W_M^(q k1) k1=1..7,
// 30..33 stage twiddles, 36..37 stage signs, 40..49 -i W_N^k of the lane's five pairs.
template <int N>
struct WaveSmall {
    static constexpr int M = N / 2, LPF = M / 8, FPW = 64 / LPF, STAGES = (LPF == 8) ? 0 : (LPF == 16) ? 1 : 2;
    static constexpr int NB = M + 1, PPF = M / 2 + 1, PAIRS = FPW * PPF;
    // complex per frame in the strip (>= NB, even; 256: 2*144 = 32 mod 64 keeps two frames apart)
    static constexpr int STRIP = (N == 128) ? 72 : (N == 256) ? 144 : 264;
    static constexpr int ROW_FLOATS = FPW * NB;            // 520 / 516 / 514: the dense row image
    static constexpr int GROUP_STORES = (N == 512) ? 5 : 3;
};

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ v2f dpp_pair(v2f old, v2f src)
{
    return v2f{dpp_update<CTRL, BANK_MASK>(old.x, src.x), dpp_update<CTRL, BANK_MASK>(old.y, src.y)};
}
template <int N, int BYTE_OFF>
__device__ __forceinline__ v2f small_frame_read(unsigned fr)
{
    return lds_read_b64<BYTE_OFF * N / 256>(fr);
}
template <int N, int OUT, bool WIDE>
__global__ __launch_bounds__(kWave1024Threads) void stft_wave_small_kernel(
    const float *__restrict__ x, float *__restrict__ out, const float *__restrict__ tab,
    int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg, int64_t n_chunks,
    float log_floor)
{
    using G = WaveSmall<N>;
    constexpr int M = G::M, LPF = G::LPF, FPW = G::FPW, NB = G::NB, PPF = G::PPF, SS = G::STRIP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // lane map before the exchange E (frame read, pass A, stages) and after it (split, store)
    const int gam = lane & 7, cA = gam & ((1 << G::STAGES) - 1);
    const int crev = G::STAGES < 2 ? cA : (((cA & 1) << 1) | (cA >> 1));
    // q and the block rotation come from the plan's table: build_wave_small_table (aa_abi.cpp) relabels
    // frame slots (q' ^ 4, block j + 1) so that the FPW frames of one read land in different banks
    const int fqA = gam >> G::STAGES, qA = (int)tab[38 * 64 + lane], rotA = (int)tab[39 * 64 + lane];
    (void)crev;
    const int fqB = lane / LPF, lB = lane % LPF;

    int c0, c1;
    chunk_run(n_chunks, c0, c1);
    const int max_span = (frames_per_wg - 1) * hop + N;
    float *scrf = smem + 2 * max_span + wave * kWave256ScratchDwords;
    v2f *scr = reinterpret_cast<v2f *>(scrf);
    constexpr int COLS = (OUT == kComplex) ? 2 * NB : NB;
    const int64_t first_frame = (int64_t)c0 * frames_per_wg;
    const float *src_chunk = x + (frame_begin + first_frame) * (int64_t)hop;
    float *out_chunk = out + first_frame * COLS;
    int frames_left = (int)(n_frames - first_frame);
    const int src_stride = frames_per_wg * hop, out_stride = frames_per_wg * COLS;

    auto stage = [&](const float *src, int nfc, int which) {
        const int span4 = ((nfc - 1) * hop + N) >> 2;
        float *dst = smem + which * max_span;
        for (int qb = wave * 64; qb < span4; qb += kWave1024Threads) {
            const int qq = qb + lane;
            if (qq < span4)
                lds_dma_16(src + 4 * qq, dst + 4 * qb);
        }
    };
    if (c0 < c1)
        stage(src_chunk, frames_left < frames_per_wg ? frames_left : frames_per_wg, 0);

    v2f win[8], twA[8], wst[2], sig[2], tws[5];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        win[j] = v2f{tab[(2 * j) * 64 + lane], tab[(2 * j + 1) * 64 + lane]};
#pragma unroll
    for (int k = 1; k < 8; ++k)
        twA[k] = v2f{tab[(16 + 2 * (k - 1)) * 64 + lane], tab[(17 + 2 * (k - 1)) * 64 + lane]};
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        wst[st] = v2f{tab[(30 + 2 * st) * 64 + lane], tab[(31 + 2 * st) * 64 + lane]};
        sig[st] = v2f{tab[(36 + st) * 64 + lane], tab[(36 + st) * 64 + lane]};
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
        tws[i] = v2f{tab[(40 + 2 * i) * 64 + lane], tab[(41 + 2 * i) * 64 + lane]};

    // pair bookkeeping: pair p = lane + 64 i -> frame p / PPF, bin k = p % PPF
    int pa[5], pb[5], rlo[5], rhi[5], pfr[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int p = lane + 64 * i;
        const int fr = p / PPF, k = p - PPF * fr;
        pfr[i] = p < G::PAIRS ? fr : FPW;              // FPW = no such pair
        pa[i] = (fr % FPW) * SS + k;
        pb[i] = (fr % FPW) * SS + M - k;
        rlo[i] = (fr % FPW) * NB + k;
        rhi[i] = (fr % FPW) * NB + M - k;
    }
    v2f *e_wr = scr + 66 * (lane & 7) + (qA & 7);                           // + 8 k1   (E, as E1 at 1024)
    const v4f *e_rd = reinterpret_cast<const v4f *>(scr + 8 * (lane & 7) + 66 * (lane >> 3));
    v2f *strip_wr = scr + fqB * SS + lB;                                    // + LPF mu
    const unsigned smem_addr = lds_addr(smem);
    const int groups_per_wave = frames_per_wg / (4 * FPW);   // in a full chunk

    for (int c = c0; c < c1; ++c) {
        const int cur = (c - c0) & 1;
        // counted wait as in stft_wave1024_kernel: in the previous (full) chunk every wave stored
        // two groups of FPW frames (GROUP_STORES stores each) after issuing this chunk's DMA loads
        if (c == c0 || groups_per_wave != 2 || frames_per_wg != 8 * FPW || OUT == kComplex || !WIDE)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (G::GROUP_STORES == 3)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        int prio_step = (int)(blockIdx.x >> 8) + 2 * (c - c0);   // see rotate_wave_priority; + 1 per group
        const int nf = frames_left < frames_per_wg ? frames_left : frames_per_wg;
        // (the next chunk's LDS-DMA behind the frame reads of the wave's first group, as in stft_wave1024_kernel, measured +-0
        // here - 68.4 against 68.7 us at 256 / 64)
        if (c + 1 < c1) {
            const int nleft = frames_left - frames_per_wg;
            stage(src_chunk + src_stride, nleft < frames_per_wg ? nleft : frames_per_wg, cur ^ 1);
        }
        // register j <- block (j + rotA) mod 8: blocks are N / 2 bytes apart; one base for j = 0..6,
        // the wrapped one for j = 7
        const unsigned buf_addr = smem_addr + 4u * (unsigned)(cur * max_span + 2 * qA) + (unsigned)(rotA * (N / 2));
        const unsigned wrap7 = rotA ? (unsigned)(8 * (N / 2)) : 0u;

        for (int g = wave; FPW * g < nf; g += 4) {
            if (AA_WSMALL_PRIO)
                __builtin_amdgcn_s_setprio((AA_WSMALL_PRIO / 10) % 10);
            else
                rotate_wave_priority(prio_step++);
            const int t = FPW * g + fqA;               // this lane's frame within the chunk
            const int nvalid = nf - FPW * g < FPW ? nf - FPW * g : FPW;
            // a lane of a frame past the end reads the last valid frame instead (stays in the
            // staged span); its results are never stored
            const int tr = t < nf ? t : nf - 1;
            const unsigned fr = buf_addr + 4u * (unsigned)(tr * hop);
            v2f v[8];
            v[0] = small_frame_read<N, 0>(fr); v[1] = small_frame_read<N, 128>(fr);
            v[2] = small_frame_read<N, 256>(fr); v[3] = small_frame_read<N, 384>(fr);
            v[4] = small_frame_read<N, 512>(fr); v[5] = small_frame_read<N, 640>(fr);
            v[6] = small_frame_read<N, 768>(fr); v[7] = small_frame_read<N, 896>(fr - wrap7);
            lds_wait_8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = v[j] * win[j];
            pk_dft8(v);
#pragma unroll
            for (int k = 1; k < 8; ++k)
                v[k] = pk_cmul(v[k], twA[k]);

            // ---- radix-2 DIF stages across lane bits 0 (and 1): partner in the same quad ----
            if (G::STAGES >= 1) {
                hazard_fence(v);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v2f p = dpp_pair<0xB1, 0xF>(v[k], v[k]);    // quad_perm [1,0,3,2]: lane ^ 1
                    v[k] = pk_cmul(v[k] * sig[0] + p, wst[0]);
                }
            }
            if (G::STAGES == 2) {
                hazard_fence(v);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v2f p = dpp_pair<0x4E, 0xF>(v[k], v[k]);    // quad_perm [2,3,0,1]: lane ^ 2
                    v[k] = pk_cmul(v[k] * sig[1] + p, wst[1]);
                }
            }
            // ---- E: q' <-> k1 through LDS (addresses of stft_wave1024_kernel's E1) ----
            fft512_exchange1(v, e_wr, e_rd);
            if (AA_WSMALL_PRIO)
                __builtin_amdgcn_s_setprio(AA_WSMALL_PRIO % 10);
            // ---- pass C: lane l of its frame ends with Z[l + LPF mu] in register mu ----
            pk_dft8(v);

            float *grp = out_chunk + (FPW * g) * COLS;   // first row of the group
            if (LPF == 16) {
                // ---- 256: the split partner without LDS.  Lane l of a frame (= one DPP row) holds
                // Z[l + 16 mu]; the partner of k = l + 16 mu (mu < 4) is Z[128 - k] = register 7 - mu
                // of lane 16 - l: a row_mirror followed by a row_shr:1 whose out-of-range lane 0 keeps
                // its `old` operand - lane 0's partner is its own register 8 - mu (Z[128] := Z[0]).
                v2f prt[4];
                hazard_fence(v);   // packed asm -> DPP read
#pragma unroll
                for (int mu = 0; mu < 4; ++mu) {
                    const v2f own = v[mu == 0 ? 0 : 8 - mu], src = v[7 - mu];
                    const float mx = dpp_update<0x140, 0xF>(src.x, src.x), my = dpp_update<0x140, 0xF>(src.y, src.y);
                    prt[mu] = v2f{dpp_update<0x111, 0xF>(own.x, mx), dpp_update<0x111, 0xF>(own.y, my)};
                }
                const bool live = fqB < nvalid;
                float mk[4], mm[4];
#pragma unroll
                for (int mu = 0; mu < 4; ++mu) {
                    const v2f e = pk_add_conj(v[mu], prt[mu]), o = pk_sub_conj(v[mu], prt[mu]);
                    const v2f tt = pk_cmul(o, tws[mu]);
                    const int k = lB + 16 * mu;
                    if (OUT == kComplex) {
                        const v2f xk = e + tt, xm = e - tt;    // X[128-k] = conj(xm)
                        if (live) {
                            reinterpret_cast<v2f *>(grp)[fqB * NB + k] = xk;
                            reinterpret_cast<v2f *>(grp)[fqB * NB + M - k] = v2f{xm.x, -xm.y};
                        }
                    } else if (AA_V_PKPOW) {
                        // both powers from one packed multiply and one packed FMA (see pk_sum_diff_re)
                        const v2f re = pk_sum_diff_re(e, tt), im = pk_sum_diff_im(e, tt);
                        const v2f pw = im * im + re * re;
                        mk[mu] = bin_from_power<OUT>(pw.x, log_floor);
                        mm[mu] = bin_from_power<OUT>(pw.y, log_floor);
                    } else {
                        const v2f xk = e + tt, xm = e - tt;
                        mk[mu] = bin_value<OUT>(xk.x, xk.y, log_floor);
                        mm[mu] = bin_value<OUT>(xm.x, xm.y, log_floor);
                    }
                }
                // bin 64 = Z[64], its own partner: lane 0 of the frame, register 4
                if (OUT == kComplex) {
                    if (live && lB == 0)
                        reinterpret_cast<v2f *>(grp)[fqB * NB + 64] = v2f{2.0f * v[4].x, -2.0f * v[4].y};
                } else {
                    float *img = scrf + fqB * NB;
#pragma unroll
                    for (int mu = 0; mu < 4; ++mu) {
                        img[lB + 16 * mu] = mk[mu];
                        img[M - lB - 16 * mu] = mm[mu];
                    }
                    if (lB == 0)
                        img[64] = bin_value<OUT>(2.0f * v[4].x, 2.0f * v[4].y, log_floor);
                }
            } else {
            // ---- strip: Z of the FPW frames, Z[M] := Z[0] ----
#pragma unroll
            for (int mu = 0; mu < 8; ++mu)
                strip_wr[LPF * mu] = v[mu];
            if (lB == 0)
                scr[fqB * SS + M] = v[0];
            __builtin_amdgcn_wave_barrier();
            v2f za[5], zb[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                za[i] = scr[pa[i]];
                zb[i] = scr[pb[i]];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                v2f e = pk_add_conj(za[i], zb[i]), o = pk_sub_conj(za[i], zb[i]);
                v2f tt = pk_cmul(o, tws[i]);
                const bool live = pfr[i] < nvalid;
                if (OUT == kComplex) {
                    v2f xk = e + tt, xm = e - tt;      // X[M-k] = conj(xm)
                    if (live) {
                        reinterpret_cast<v2f *>(grp)[rlo[i]] = xk;
                        reinterpret_cast<v2f *>(grp)[rhi[i]] = v2f{xm.x, -xm.y};
                    }
                } else {
                    float mk, mm;
                    if (AA_V_PKPOW) {
                        const v2f re = pk_sum_diff_re(e, tt), im = pk_sum_diff_im(e, tt);
                        const v2f pw = im * im + re * re;
                        mk = bin_from_power<OUT>(pw.x, log_floor);
                        mm = bin_from_power<OUT>(pw.y, log_floor);
                    } else {
                        v2f xk = e + tt, xm = e - tt;
                        mk = bin_value<OUT>(xk.x, xk.y, log_floor);
                        mm = bin_value<OUT>(xm.x, xm.y, log_floor);
                    }
                    if (pfr[i] < FPW) {
                        scrf[rlo[i]] = mk;
                        scrf[rhi[i]] = mm;
                    }
                }
            }
            }   // LPF != 16
            if (OUT != kComplex) {
                __builtin_amdgcn_wave_barrier();
                if (WIDE && nvalid == FPW && G::ROW_FLOATS % 4 == 0) {
                    // 520 / 516 floats = 130 / 129 float4, group start 16-byte aligned
                    v4f b0 = *reinterpret_cast<const v4f *>(scrf + 4 * lane);
                    v4f b1 = *reinterpret_cast<const v4f *>(scrf + 256 + 4 * lane);
                    v4f b2 = *reinterpret_cast<const v4f *>(scrf + 512 + 4 * (lane & 1));
                    row_store(b0, reinterpret_cast<v4f *>(grp + 4 * lane));
                    row_store(b1, reinterpret_cast<v4f *>(grp + 256 + 4 * lane));
                    if (4 * lane < G::ROW_FLOATS - 512)
                        row_store(b2, reinterpret_cast<v4f *>(grp + 512 + 4 * lane));
                } else if (WIDE && nvalid == FPW) {
                    // 514 floats = 257 float2, group start 8-byte aligned
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        v2f b = *reinterpret_cast<const v2f *>(scrf + 128 * u + 2 * lane);
                        row_store(b, reinterpret_cast<v2f *>(grp + 128 * u + 2 * lane));
                    }
                    v2f bt = *reinterpret_cast<const v2f *>(scrf + 512);
                    if (lane == 0)
                        row_store(bt, reinterpret_cast<v2f *>(grp + 512));
                } else {
                    const int nfl = NB * nvalid;
#pragma unroll
                    for (int u = 0; u < 9; ++u) {
                        const int idx = lane + 64 * u;
                        if (idx < nfl)
                            grp[idx] = scrf[idx];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        src_chunk += src_stride;
        out_chunk += out_stride;
        frames_left -= frames_per_wg;
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
// PAD: Z is stored at index i + (i >> 3) (the bank-conflict padding of stft_wg_kernel).
template <int OUT, bool PAD = false>
__device__ __forceinline__ void split_and_store(const float2 *Z, const float2 *tw_split, int m,
                                                float *row, float log_floor)
{
    auto at = [&](int i) { return Z[PAD ? i + (i >> 3) : i]; };
    for (int k = threadIdx.x; 2 * k <= m; k += blockDim.x) {
        if (k == 0) {
            float2 z = at(0);
            store_bin<OUT>(row, 0, z.x + z.y, 0.0f, log_floor);
            store_bin<OUT>(row, m, z.x - z.y, 0.0f, log_floor);
            continue;
        }
        const int kk = m - k;
        float2 a = at(k), p = at(kk);
        float2 bc = make_float2(p.x, -p.y);
        float2 e = make_float2(0.5f * (a.x + bc.x), 0.5f * (a.y + bc.y));
        float2 o = make_float2(0.5f * (a.x - bc.x), 0.5f * (a.y - bc.y));
        float2 wo = cmul(o, tw_split[k]);
        float2 tt = make_float2(wo.y, -wo.x);          // -i * W * O
        store_bin<OUT>(row, k, e.x + tt.x, e.y + tt.y, log_floor);
        store_bin<OUT>(row, kk, e.x - tt.x, -(e.y - tt.y), log_floor);
    }
}

// R-point forward DFT in place, natural order (R = 2, 4, 8), plain float2.
template <int R>
__device__ __forceinline__ void small_dft(float2 (&v)[R])
{
    if (R == 2) {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    } else if (R == 4) {
        const float2 a0 = cadd(v[0], v[2]), b0 = csub(v[0], v[2]);
        const float2 a1 = cadd(v[1], v[3]), b1 = csub(v[1], v[3]);
        const float2 jb1 = make_float2(b1.y, -b1.x);            // -i b1
        v[0] = cadd(a0, a1);
        v[2] = csub(a0, a1);
        v[1] = cadd(b0, jb1);
        v[3] = csub(b0, jb1);
    } else {
        const float h = 0.70710678118654752440f;
        float2 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
        small_dft<4>(e);
        small_dft<4>(o);
        const float2 o1 = make_float2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));    // W8^1 o1
        const float2 o2 = make_float2(o[2].y, -o[2].x);                                 // W8^2 o2
        const float2 o3 = make_float2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));   // W8^3 o3
        v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
        v[1] = cadd(e[1], o1);   v[5] = csub(e[1], o1);
        v[2] = cadd(e[2], o2);   v[6] = csub(e[2], o2);
        v[3] = cadd(e[3], o3);   v[7] = csub(e[3], o3);
    }
}
// One Stockham (autosort) pass of radix R over m points: sub-transforms of size ns -> ns R.
// tw_m holds W_m^q for q < m.
template <int R>
__device__ __forceinline__ void stockham_pass(const float2 *A, float2 *B, const float2 *__restrict__ tw_m,
                                              int m, int ns)
{
    const int mr = m / R, tstep = m / (ns * R);
    for (int j = threadIdx.x; j < mr; j += blockDim.x) {
        const int k = j & (ns - 1);
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
            v[r] = A[j + r * mr];
        if (ns > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r)
                v[r] = cmul(v[r], tw_m[r * k * tstep]);
        }
        small_dft<R>(v);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r)
            B[j0 + r * ns] = v[r];
    }
}

// LDS Stockham FFT, one workgroup per frame, any power-of-two n >= 4: radix-8 passes, then one
// radix-4 or radix-2 pass for what is left (2048 complex points: 4 passes instead of 11).
template <int OUT>
__global__ void stft_lds_stockham_kernel(const float *__restrict__ x, float *__restrict__ out,
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
    int ns = 1;
    while (m / ns >= 8) {
        stockham_pass<8>(A, B, tw_m, m, ns);
        ns *= 8;
        __syncthreads();
        float2 *tmp = A;
        A = B;
        B = tmp;
    }
    if (m / ns == 4) {
        stockham_pass<4>(A, B, tw_m, m, ns);
        __syncthreads();
        A = B;
    } else if (m / ns == 2) {
        stockham_pass<2>(A, B, tw_m, m, ns);
        __syncthreads();
        A = B;
    }
    const int cols = (OUT == kComplex) ? 2 * (m + 1) : (m + 1);
    split_and_store<OUT>(A, tw_split, m, out + t * (int64_t)cols, log_floor);
}

// Frames the wave kernels do not take (n_fft 4096 / 8192; 1024 / 2048 with an odd hop or an
// unaligned start): one persistent workgroup of M/8 <= 256 threads per frame in flight, the same Stockham passes as stft_lds_stockham_kernel, but
//   * every thread keeps the twiddles and window values of ITS butterflies in registers for the
//     kernel's lifetime (they do not depend on the frame; the generic kernel gathers them from
//     global memory for every butterfly, which is what it is bound by),
//   * the first pass reads the samples straight from global memory (a wave reads 64 consecutive
//     complex points per load) and the next frame's samples are fetched into registers before
//     the passes of the current frame,
//   * radix-8 passes: 512 = 8 x 8 x 8, 1024 = 8 x 8 x 8 x 2, 2048 = 8 x 8 x 8 x 4, 4096 = 8^4.
template <int LOGM>
struct WgFft {
    static constexpr int M = 1 << LOGM, T = (M / 8 < 256) ? M / 8 : 256;  // threads per frame: 64 / 128 / 256 / 256
    static constexpr int PTS = M / T;                                      // 8 (16 at 8192) points per thread
    static constexpr int LAST = (LOGM % 3 == 0) ? 8 : (LOGM % 3 == 2 ? 4 : 2);   // radix of the last pass
    static constexpr int NPASS8 = (LOGM % 3 == 0) ? LOGM / 3 - 1 : LOGM / 3;     // radix-8 passes before it
};
template <int LOGM, int OUT>
__global__ __launch_bounds__(256) void stft_wg_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                      const float *__restrict__ window,
                                                      const float2 *__restrict__ tw_m,
                                                      const float2 *__restrict__ tw_split, int64_t frame_begin,
                                                      int64_t n_frames, int64_t hop, float log_floor)
{
    using G = WgFft<LOGM>;
    constexpr int M = G::M, T = G::T, PTS = G::PTS, B8 = PTS / 8;   // radix-8 butterflies per thread and pass
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // The Stockham scatter of the first two passes strides 8 / 64 points between neighbouring
    // threads (8- and 4-way bank conflicts); storing point i at i + (i >> 3) spreads both the
    // scatters and the unit-stride gathers over all banks.  9/8 of 2 x M points must fit 64 KiB.
    constexpr bool PAD = (2 * (M + M / 8) * 8 <= 65536);
    constexpr int MP = PAD ? M + M / 8 : M;
    auto ix = [](int i) { return PAD ? i + (i >> 3) : i; };
    float2 *bufA = reinterpret_cast<float2 *>(smem), *bufB = bufA + MP;
    const int tid = threadIdx.x;
    constexpr int cols = (OUT == kComplex) ? 2 * (M + 1) : (M + 1);

    // ---- per-thread constants: window of its first-pass points, twiddles of its butterflies ----
    float2 win[PTS];
#pragma unroll
    for (int u = 0; u < B8; ++u)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = tid + T * u + r * (M / 8);
            win[8 * u + r] = window ? make_float2(window[2 * i], window[2 * i + 1]) : make_float2(1.0f, 1.0f);
        }
    float2 tw8[G::NPASS8 > 1 ? G::NPASS8 - 1 : 1][B8][7];       // radix-8 passes 2.. (the first has none)
    {
        int ns = 8;
#pragma unroll
        for (int p = 0; p < G::NPASS8 - 1; ++p) {
#pragma unroll
            for (int u = 0; u < B8; ++u) {
                const int k = (tid + T * u) & (ns - 1), tstep = M / (ns * 8);
#pragma unroll
                for (int r = 1; r < 8; ++r)
                    tw8[p][u][r - 1] = tw_m[r * k * tstep];
            }
            ns *= 8;
        }
    }
    constexpr int NS_LAST = M / G::LAST, BL = (M / G::LAST) / T;  // last pass: ns and butterflies per thread
    float2 twl[BL][G::LAST - 1];
#pragma unroll
    for (int u = 0; u < BL; ++u) {
        const int k = (tid + T * u) & (NS_LAST - 1);
#pragma unroll
        for (int r = 1; r < G::LAST; ++r)
            twl[u][r - 1] = tw_m[r * k];                          // tstep = M / (ns R) = 1
    }

    auto fetch = [&](int64_t t, float2 (&v)[PTS]) {
        const float *src = x + (frame_begin + t) * hop;
        const bool al8 = (reinterpret_cast<uintptr_t>(src) & 7u) == 0;
#pragma unroll
        for (int u = 0; u < B8; ++u)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = tid + T * u + r * (M / 8);
                if (al8)
                    v[8 * u + r] = *reinterpret_cast<const float2 *>(src + 2 * i);
                else
                    v[8 * u + r] = make_float2(src[2 * i], src[2 * i + 1]);
            }
    };

    float2 nx[PTS];
    int64_t t = blockIdx.x;
    if (t < n_frames)
        fetch(t, nx);
    for (; t < n_frames; t += gridDim.x) {
        float2 v[PTS];
#pragma unroll
        for (int i = 0; i < PTS; ++i)
            v[i] = make_float2(nx[i].x * win[i].x, nx[i].y * win[i].y);
        if (t + gridDim.x < n_frames)
            fetch(t + gridDim.x, nx);                            // in flight during the passes below
        // ---- pass 1 (radix 8, ns = 1): registers -> bufA, no twiddles ----
#pragma unroll
        for (int u = 0; u < B8; ++u) {
            float2 b[8];
#pragma unroll
            for (int r = 0; r < 8; ++r)
                b[r] = v[8 * u + r];
            small_dft<8>(b);
            const int j = tid + T * u;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                bufA[ix(j * 8 + r)] = b[r];
        }
        __syncthreads();
        float2 *A = bufA, *B = bufB;
        // ---- radix-8 passes 2.. ----
        int ns = 8;
#pragma unroll
        for (int p = 0; p < G::NPASS8 - 1; ++p) {
#pragma unroll
            for (int u = 0; u < B8; ++u) {
                const int j = tid + T * u, k = j & (ns - 1);
                float2 b[8];
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    b[r] = A[ix(j + r * (M / 8))];
#pragma unroll
                for (int r = 1; r < 8; ++r)
                    b[r] = cmul(b[r], tw8[p][u][r - 1]);
                small_dft<8>(b);
                const int j0 = (j - k) * 8 + k;
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    B[ix(j0 + r * ns)] = b[r];
            }
            __syncthreads();
            float2 *tmp = A;
            A = B;
            B = tmp;
            ns *= 8;
        }
        // ---- last pass (radix LAST, ns = M / LAST) ----
#pragma unroll
        for (int u = 0; u < BL; ++u) {
            const int j = tid + T * u, k = j & (NS_LAST - 1);
            float2 b[G::LAST];
#pragma unroll
            for (int r = 0; r < G::LAST; ++r)
                b[r] = A[ix(j + r * (M / G::LAST))];
#pragma unroll
            for (int r = 1; r < G::LAST; ++r)
                b[r] = cmul(b[r], twl[u][r - 1]);
            small_dft<G::LAST>(b);
            const int j0 = (j - k) * G::LAST + k;
#pragma unroll
            for (int r = 0; r < G::LAST; ++r)
                B[ix(j0 + r * NS_LAST)] = b[r];
        }
        __syncthreads();
        split_and_store<OUT, PAD>(B, tw_split, M, out + t * (int64_t)cols, log_floor);
        __syncthreads();                                         // B is read above, written next frame
    }
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
// Persistent grid: resident workgroups per CU (occupancy API) x CUs.  There is no
// inter-workgroup dependency, so an over- or under-estimate only changes speed.
// Experiment knobs of the launchers (AA_GRID_MULT, AA_GRID_SET, AA_W2048_RING, AA_LDS_PAD: experiment_knob, aa_internal.hpp)
// exist only in libraries built with -DAA_EXPERIMENTS=1 (tools/probe_bimodal_grid.py, tools/profile_ring.sh).
template <typename K>
static int persistent_grid(K kernel, size_t lds)
{
    int dev = 0, cus = 256, per_cu = 1;
    if (hipGetDevice(&dev) == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kWave1024Threads, lds) != hipSuccess ||
        per_cu < 1)
        per_cu = 1;
    // AA_GRID_MULT (experiment builds): that many dispatch generations per resident slot instead of one
    // persistent workgroup (the hardware dispatcher then balances the tail)
    return cus * per_cu * experiment_knob("AA_GRID_MULT", 1, 1, 16);
}

// Grid of a launch over ch channels: ch equal groups of workgroups (stream_of_block), together at most
// the persistent grid of the device.
static unsigned channel_grid(int64_t persistent, int64_t chunks, int n_channels)
{
    int64_t per = persistent / (n_channels > 0 ? n_channels : 1);
    if (per < 1)
        per = 1;
    if (per > chunks)
        per = chunks;
    return (unsigned)(per * (n_channels > 0 ? n_channels : 1));
}

template <int K>
static hipError_t launch_wave1024_kind(bool wide, const float *x, float *out, const float *table,
                                       int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg,
                                       float log_floor, hipStream_t stream, const float *mel_par,
                                       const MelSched &sch, const Channels &ch)
{
    const int64_t chunks = (n_frames + frames_per_wg - 1) / frames_per_wg;
    const size_t lds = wave1024_lds_bytes(frames_per_wg, hop, K == kLogMel ? 1 : 0);
    static thread_local int cached_grid = 0;      // per thread: plans of different threads are independent
    static thread_local size_t cached_lds = 0;
    if (cached_grid == 0 || cached_lds != lds) {
        cached_grid = persistent_grid(stft_wave1024_kernel<K, true>, lds);
        cached_lds = lds;
    }
    // AA_GRID_SET (experiment builds): another number of persistent workgroups
    const int64_t want = experiment_knob("AA_GRID_SET", cached_grid, 1, 1 << 16);
    const dim3 grid(channel_grid(want, chunks, ch.n)), block(kWave1024Threads);
    if (wide)
        hipLaunchKernelGGL((stft_wave1024_kernel<K, true>), grid, block, lds, stream, x, out, table,
                           frame_begin, n_frames, hop, frames_per_wg, chunks, log_floor, mel_par, sch,
                           ch.n, ch.in_stride, ch.out_stride);
    else
        hipLaunchKernelGGL((stft_wave1024_kernel<K, false>), grid, block, lds, stream, x, out, table,
                           frame_begin, n_frames, hop, frames_per_wg, chunks, log_floor, mel_par, sch,
                           ch.n, ch.in_stride, ch.out_stride);
    return hipGetLastError();
}

// kLogMelM / kMfccM: stft_wave1024_mel_kernel (8 frames per chunk; arguments validated by launch_wave1024)
template <int K>
static hipError_t launch_wave1024_mel_kind(const float *x, float *out, const float *table, int64_t frame_begin,
                                           int64_t n_frames, int hop, float log_floor, hipStream_t stream,
                                           const float *mel_tab, const MelSched &sch, const float *dct_tab, int n_mfcc,
                                           const Channels &ch)
{
    const int64_t chunks = (n_frames + 7) / 8;
    const size_t lds = wave1024_lds_bytes(8, hop, K == kMfccM ? 4 : 3);
    static thread_local int cached_grid = 0;      // per thread: plans of different threads are independent
    static thread_local size_t cached_lds = 0;
    if (cached_grid == 0 || cached_lds != lds) {
        cached_grid = persistent_grid(stft_wave1024_mel_kernel<K>, lds);
        cached_lds = lds;
    }
    const int64_t want = experiment_knob("AA_GRID_SET", cached_grid, 1, 1 << 16);
    const dim3 grid(channel_grid(want, chunks, ch.n)), block(kWave1024Threads);
    hipLaunchKernelGGL((stft_wave1024_mel_kernel<K>), grid, block, lds, stream, x, out, table, frame_begin, n_frames, hop,
                       chunks, log_floor, mel_tab, sch.n_mels, dct_tab, n_mfcc, ch.n, ch.in_stride, ch.out_stride);
    return hipGetLastError();
}

#ifdef AA_TU_MFCC
// aa_kernels_mfcc.hip: this file once more, for the fused MFCC instantiation of stft_wave1024_mel_kernel alone, compiled with
// -mllvm -amdgpu-sched-strategy=max-ilp (r3: -2.5 % for the fused MFCC kernel of that round; as a flag of the whole library it makes
// fallback instantiations of the 2048 kernel spill and costs the 256 kernel a wave per SIMD, csrc/Makefile).
hipError_t launch_wave1024_mfcc(const float *x, float *out, const float *table, int64_t frame_begin, int64_t n_frames, int hop,
                                float log_floor, hipStream_t stream, const float *mel_par, const MelSched &sch,
                                const float *dct_tab, int n_mfcc, const Channels &ch)
{
    return launch_wave1024_mel_kind<kMfccM>(x, out, table, frame_begin, n_frames, hop, log_floor, stream, mel_par, sch, dct_tab,
                                            n_mfcc, ch);
}
#else
hipError_t launch_wave1024(int out_kind, const float *x, float *out, const float *table,
                           int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg,
                           float log_floor, hipStream_t stream, const float *mel_par, const MelSched *sched,
                           const float *dct_tab, int n_mfcc, const Channels *channels)
{
    const Channels ch = channels ? *channels : Channels();
    if (ch.n < 1 || (ch.n > 1 && ((ch.in_stride | ch.out_stride) & 3)))
        return hipErrorInvalidValue;
    if (n_frames <= 0)
        return hipSuccess;
    const MelSched sch = sched ? *sched : MelSched();
    if (out_kind == kLogMel && (!mel_par || sch.n_groups < 1 || sch.n_groups > kMelMaxGroups))
        return hipErrorInvalidValue;
    if ((out_kind == kLogMelM || out_kind == kMfccM) &&
        (!mel_par || sch.n_steps < 1 || sch.n_steps > kMelMfmaSteps || frames_per_wg != 8 || sch.n_mels < 16 ||
         sch.n_mels > kMelMaxMels || (sch.n_mels & 3)))
        return hipErrorInvalidValue;
    if (out_kind == kLogMelM && (reinterpret_cast<uintptr_t>(out) & 15u))   // dwordx4 row pieces
        return hipErrorInvalidValue;
    if (out_kind == kMfccM && (!dct_tab || n_mfcc < 1 || n_mfcc > 16 * kMfccTiles))
        return hipErrorInvalidValue;
    // aligned nontemporal row stores need a 16-byte-aligned output buffer
    const bool wide = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    switch (out_kind) {
    case kComplex:
        return launch_wave1024_kind<kComplex>(wide, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream, mel_par, sch, ch);
    case kMag:
        return launch_wave1024_kind<kMag>(wide, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream, mel_par, sch, ch);
    case kPower:
        return launch_wave1024_kind<kPower>(wide, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream, mel_par, sch, ch);
    case kLogPower:
        return launch_wave1024_kind<kLogPower>(wide, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream, mel_par, sch, ch);
    case kLogMel:
        return launch_wave1024_kind<kLogMel>(false, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream, mel_par, sch, ch);
    case kLogMelM:
        return launch_wave1024_mel_kind<kLogMelM>(x, out, table, frame_begin, n_frames, hop, log_floor, stream, mel_par, sch, nullptr, 0, ch);
    case kMfccM:   // built in a translation unit of its own (aa_kernels_mfcc.hip: another scheduler strategy)
        return launch_wave1024_mfcc(x, out, table, frame_begin, n_frames, hop, log_floor, stream, mel_par, sch, dct_tab, n_mfcc, ch);
    default:
        return hipErrorInvalidValue;
    }
}

template <int K>
static hipError_t launch_wave2048_kind(bool wide, const float *x, float *out, const float *table,
                                       int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg,
                                       float log_floor, hipStream_t stream, const float *mel_par,
                                       const MelSched &sch, const Channels &ch)
{
    const int64_t chunks = (n_frames + frames_per_wg - 1) / frames_per_wg;
    constexpr int kMelKind = K == kLogMel ? 1 : 0;
    // The tuner geometry (hop 512, four frames per chunk) stages through the ring with the DMA two chunks ahead
    // (RING == 2; r2: magnitudes 82.9 -> 80.1 us, complex rows 96.3 -> 91.7 with the samples coming from HBM).  The
    // fallback log-mel rows with arithmetic weights keep the two chunk buffers (their LDS has no room for the ring).
    // AA_W2048_RING=0: experiment builds.
    constexpr int kRingKind = K == kLogMel ? 0 : 2;   // built for this output kind
    int ring = hop == 512 && frames_per_wg == 4 ? kRingKind : 0;
    if (experiment_knob("AA_W2048_RING", 1, 0, 1) == 0)     // experiment builds: the two chunk buffers
        ring = 0;
    size_t lds = ring ? wave2048_ring_lds_bytes(30, kMelKind)
                      : wave2048_lds_bytes(frames_per_wg, hop, kMelKind != 0);
    // AA_LDS_PAD (experiment builds): fewer resident workgroups per CU; never past the 64 KiB a plan validated
    const size_t pad = (size_t)experiment_knob("AA_LDS_PAD", 0, 0, 65536);
    if (lds + pad <= 65536)
        lds += pad;
    const int ri = ring ? 1 : 0;
    static thread_local int cached_grid[2] = {0, 0};   // per thread: plans of different threads are independent
    static thread_local size_t cached_lds[2] = {0, 0};
    if (cached_grid[ri] == 0 || cached_lds[ri] != lds) {
        cached_grid[ri] = ring ? persistent_grid(stft_wave2048_kernel<K, true, kRingKind>, lds)
                               : persistent_grid(stft_wave2048_kernel<K, true, 0>, lds);
        cached_lds[ri] = lds;
    }
    const dim3 grid(channel_grid(cached_grid[ri], chunks, ch.n)), block(kWave1024Threads);
    if (ring && wide)
        hipLaunchKernelGGL((stft_wave2048_kernel<K, true, kRingKind>), grid, block, lds, stream, x, out, table,
                           frame_begin, n_frames, hop, frames_per_wg, chunks, log_floor, mel_par, sch, ch.n, ch.in_stride,
                           ch.out_stride);
    else if (ring)
        hipLaunchKernelGGL((stft_wave2048_kernel<K, false, kRingKind>), grid, block, lds, stream, x, out, table,
                           frame_begin, n_frames, hop, frames_per_wg, chunks, log_floor, mel_par, sch, ch.n, ch.in_stride,
                           ch.out_stride);
    else if (wide)
        hipLaunchKernelGGL((stft_wave2048_kernel<K, true, 0>), grid, block, lds, stream, x, out, table,
                           frame_begin, n_frames, hop, frames_per_wg, chunks, log_floor, mel_par, sch, ch.n, ch.in_stride,
                           ch.out_stride);
    else
        hipLaunchKernelGGL((stft_wave2048_kernel<K, false, 0>), grid, block, lds, stream, x, out, table,
                           frame_begin, n_frames, hop, frames_per_wg, chunks, log_floor, mel_par, sch, ch.n, ch.in_stride,
                           ch.out_stride);
    return hipGetLastError();
}

// kLogMelM at hop 512: stft_wave2048_mel_kernel (arguments validated by launch_wave2048)
static hipError_t launch_wave2048_mel(const float *x, float *out, const float *table, int64_t frame_begin, int64_t n_frames,
                                      float log_floor, hipStream_t stream, const float *mel_tab, const MelSched &sch,
                                      const Channels &ch)
{
    const int64_t chunks = (n_frames + 3) / 4;
    const size_t lds = wave2048_mel_lds_bytes();
    static thread_local int cached_grid = 0;      // per thread: plans of different threads are independent
    if (cached_grid == 0)
        cached_grid = persistent_grid(stft_wave2048_mel_kernel<kLogMelM>, lds);
    const dim3 grid(channel_grid(cached_grid, chunks, ch.n)), block(kWave1024Threads);
    hipLaunchKernelGGL((stft_wave2048_mel_kernel<kLogMelM>), grid, block, lds, stream, x, out, table, frame_begin, n_frames,
                       chunks, log_floor, mel_tab, sch.n_mels, sch.n_steps, ch.n, ch.in_stride, ch.out_stride);
    return hipGetLastError();
}

hipError_t launch_wave2048(int out_kind, const float *x, float *out, const float *table,
                           int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg,
                           float log_floor, hipStream_t stream, const float *mel_par, const MelSched *sched,
                           const Channels *channels)
{
    const Channels ch = channels ? *channels : Channels();
    if (ch.n < 1 || (ch.n > 1 && ((ch.in_stride | ch.out_stride) & 3)))
        return hipErrorInvalidValue;
    if (n_frames <= 0)
        return hipSuccess;
    const MelSched sch = sched ? *sched : MelSched();
    if (out_kind == kLogMel && (!mel_par || sch.n_groups < 1 || sch.n_groups > kMelMaxGroups))
        return hipErrorInvalidValue;
    if (out_kind == kLogMelM) {
        if (!mel_par || hop != 512 || frames_per_wg != 4 || sch.n_steps < 1 || sch.n_steps > kMelMfmaSteps2048 ||
            sch.n_mels < 16 || sch.n_mels > kMelMaxMels || (sch.n_mels & 3) || (reinterpret_cast<uintptr_t>(out) & 15u))
            return hipErrorInvalidValue;
        return launch_wave2048_mel(x, out, table, frame_begin, n_frames, log_floor, stream, mel_par, sch, ch);
    }
    const bool wide = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    switch (out_kind) {
    case kComplex:
        return launch_wave2048_kind<kComplex>(wide, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream, mel_par, sch, ch);
    case kMag:
        return launch_wave2048_kind<kMag>(wide, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream, mel_par, sch, ch);
    case kPower:
        return launch_wave2048_kind<kPower>(wide, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream, mel_par, sch, ch);
    case kLogPower:
        return launch_wave2048_kind<kLogPower>(wide, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream, mel_par, sch, ch);
    case kLogMel:
        return launch_wave2048_kind<kLogMel>(false, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream, mel_par, sch, ch);
    default:
        return hipErrorInvalidValue;
    }
}

template <int N, int K>
static hipError_t launch_wave_small_kind(bool wide, const float *x, float *out, const float *table,
                                      int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg,
                                      float log_floor, hipStream_t stream)
{
    const int64_t chunks = (n_frames + frames_per_wg - 1) / frames_per_wg;
    const size_t lds = wave_small_lds_bytes(N, frames_per_wg, hop);
    static thread_local int cached_grid = 0;      // per thread: plans of different threads are independent
    static thread_local size_t cached_lds = 0;
    if (cached_grid == 0 || cached_lds != lds) {
        cached_grid = persistent_grid(stft_wave_small_kernel<N, K, true>, lds);
        cached_lds = lds;
    }
    const dim3 grid((unsigned)(cached_grid < chunks ? cached_grid : chunks)), block(kWave1024Threads);
    if (wide)
        hipLaunchKernelGGL((stft_wave_small_kernel<N, K, true>), grid, block, lds, stream, x, out, table,
                           frame_begin, n_frames, hop, frames_per_wg, chunks, log_floor);
    else
        hipLaunchKernelGGL((stft_wave_small_kernel<N, K, false>), grid, block, lds, stream, x, out, table,
                           frame_begin, n_frames, hop, frames_per_wg, chunks, log_floor);
    return hipGetLastError();
}

template <int N>
static hipError_t launch_wave_small_n(int out_kind, const float *x, float *out, const float *table,
                                      int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg,
                                      float log_floor, hipStream_t stream)
{
    if (n_frames <= 0)
        return hipSuccess;
    const bool wide = out_kind != kComplex && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    switch (out_kind) {
    case kComplex:
        return launch_wave_small_kind<N, kComplex>(wide, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream);
    case kMag:
        return launch_wave_small_kind<N, kMag>(wide, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream);
    case kPower:
        return launch_wave_small_kind<N, kPower>(wide, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream);
    case kLogPower:
        return launch_wave_small_kind<N, kLogPower>(wide, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream);
    default:
        return hipErrorInvalidValue;
    }
}

hipError_t launch_wave_small(int n_fft, int out_kind, const float *x, float *out, const float *table,
                             int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg,
                             float log_floor, hipStream_t stream)
{
    if (n_fft == 128)
        return launch_wave_small_n<128>(out_kind, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream);
    if (n_fft == 256)
        return launch_wave_small_n<256>(out_kind, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream);
    if (n_fft == 512)
        return launch_wave_small_n<512>(out_kind, x, out, table, frame_begin, n_frames, hop, frames_per_wg, log_floor, stream);
    return hipErrorInvalidValue;
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
    if (pow2 && n >= 1024 && n <= 8192) {
        // persistent workgroups with register-resident twiddles; buffers padded when they fit
        const size_t lds = n <= 4096 ? sizeof(float2) * 2 * (size_t)(m + m / 8) : sizeof(float2) * 2 * (size_t)m;
        const int threads = m / 8 < 256 ? m / 8 : 256;
#define AA_LAUNCH_WG(LOGM, K)                                                                              \
    {                                                                                                      \
        static int per_cu = 0;                                                                             \
        if (!per_cu && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stft_wg_kernel<LOGM, K>,      \
                                                                    threads, lds) != hipSuccess)            \
            per_cu = 1;                                                                                    \
        int dev = 0, cus = 256;                                                                            \
        (void)hipGetDevice(&dev);                                                                          \
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);                     \
        int64_t g = (int64_t)cus * (per_cu > 0 ? per_cu : 1);                                              \
        if (g > n_frames)                                                                                  \
            g = n_frames;                                                                                  \
        hipLaunchKernelGGL((stft_wg_kernel<LOGM, K>), dim3((unsigned)g), dim3(threads), lds, stream, x,    \
                           out, window, twm, tws, frame_begin, n_frames, hop, log_floor);                  \
    }
#define AA_SWITCH_WG(LOGM)                                                                     \
    switch (out_kind) {                                                                        \
    case kComplex: AA_LAUNCH_WG(LOGM, kComplex) break;                                         \
    case kMag: AA_LAUNCH_WG(LOGM, kMag) break;                                                 \
    case kPower: AA_LAUNCH_WG(LOGM, kPower) break;                                             \
    case kLogPower: AA_LAUNCH_WG(LOGM, kLogPower) break;                                       \
    default: return hipErrorInvalidValue;                                                      \
    }
        if (n == 1024) {
            AA_SWITCH_WG(9)
        } else if (n == 2048) {
            AA_SWITCH_WG(10)
        } else if (n == 4096) {
            AA_SWITCH_WG(11)
        } else {
            AA_SWITCH_WG(12)
        }
#undef AA_SWITCH_WG
#undef AA_LAUNCH_WG
        return hipGetLastError();
    }
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
        AA_SWITCH_GEN(stft_lds_stockham_kernel)
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

#endif  // AA_TU_MFCC

}  // namespace aa
