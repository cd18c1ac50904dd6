// Design aid (never shipped): what one frame of the onset gates' two recurrences costs a LONE wave on gfx950 - the per-bin
// floor update (onset.rs:311-332: an IEEE f32 division inside the loop-carried chain) and the scalar chains (energy EMA,
// FluxTracker, frames-since) - and what the candidates for shortening the loop-carried chains buy.  One workgroup, one
// wave per variant run; cycles from s_memtime, time from wall_clock64 (100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/bin/ubench_gate_chain tools/ubench_gate_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int V>
__device__ __forceinline__ float bin_step(float &fl, float mag, float gf)
{
    const float FLOOR_OVERCOMPENSATE = 1.3f, FLOOR_RISE = 0.1f, FLOOR_DECAY = 0.04f, BIN_BURST_RATIO = 2.5f;
    const float floor_eps = fmaxf(gf, 0.01f);
    const float den = fmaxf(fl, floor_eps);
    const float d = mag - fl;
    const float burst_f = mag * FLOOR_OVERCOMPENSATE, rise_f = fl + FLOOR_RISE * d, decay_f = fl + FLOOR_DECAY * d;
    const float slow = mag > fl ? rise_f : decay_f;
    float r;
    bool burst;
    if (V == 0) {                       // as shipped: the quotient decides
        r = mag / den;
        burst = r > BIN_BURST_RATIO;
    } else if (V == 1) {                // the decision from an exact f64 product, the quotient off the chain
        // fl32(mag / den) > 2.5  <=>  mag / den > 2.5 + 2^-23 (the midpoint above 2.5; ties go to 2.5, even mantissa)
        // <=>  mag > den * (2.5 + 2^-23): 24 x 25 significant bits, exact in f64
        burst = (double)mag > (double)den * 2.50000011920928955078125;
        r = mag / den;
    } else if (V == 2) {                // no quotient at all (lower bound of the chain)
        burst = (double)mag > (double)den * 2.50000011920928955078125;
        r = den;
    } else {                            // f32 only: two FMAs decide, exact (see the text in DESIGN.md)
        const float e = __builtin_fmaf(-den, 2.5f, mag);       // fl(mag - 2.5 den): sign exact
        // mag - 2.5 den > den 2^-23 ?  e is the rounded difference; the residual of that rounding is exact in one more FMA
        const float t = den * 1.1920928955078125e-07f;         // den 2^-23, exact (power of two)
        burst = e > t;                   // (approximate variant: timing only)
        r = mag / den;
    }
    fl = burst ? burst_f : slow;
    return r;
}

template <int V>
__global__ void k_bins(const float *mags, float *out, long long *t, int blocks)
{
    __shared__ float img[32 * 131];
    const int lane = threadIdx.x;
    float cur[32];
#pragma unroll
    for (int i = 0; i < 32; ++i)
        cur[i] = mags[(i * 64 + lane) & 4095];
    float fl = cur[0];
    const long long w0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < blocks; ++b) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float gf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur[3]), i));
            img[i * 131 + lane] = bin_step<V>(fl, cur[i], gf);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i)
            cur[i] = cur[i] * 1.0001f + 0.001f;          // (keeps the inputs moving; off the chain)
        __builtin_amdgcn_s_barrier();
    }
    const long long w1 = wall_clock64(), c1 = __builtin_amdgcn_s_memtime();
    out[lane] = fl + img[lane];
    if (lane == 0) {
        t[0] = w1 - w0;
        t[1] = c1 - c0;
    }
}

// the scalar chains, one frame per step; inputs of the block's 32 frames sit in lanes 0..31.  V = 0: as shipped (v_readlane
// per operand); V = 1: the operands of a frame packed into LDS rows and read back as one wave-uniform ds_read_b128 pair
template <int V>
__global__ void k_chain(const float *in, float *out, long long *t, int blocks)
{
    __shared__ __attribute__((aligned(16))) float ops[32 * 8];
    const int lane = threadIdx.x;
    float v_flux = in[lane & 31], v_energy = in[32 + (lane & 31)];
    const int bo_l = lane & 1, v_sup = 0;
    float ema = 0.1f, thr = 0.9f;
    int since = 4;
    float o_ema = 0.0f;
    int o_flags = 0;
    const float ENERGY_EMA_RISE = 0.84f, ENERGY_EMA_DECAY = 0.95f, multiplier = 1.5f, rise_memory = 0.84f, decay_memory = 0.89f;
    const long long w0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < blocks; ++b) {
        const float e_rise = v_energy * (1.0f - ENERGY_EMA_RISE), e_dec = v_energy * (1.0f - ENERGY_EMA_DECAY);
        const float f_rise = v_flux * (1.0f - rise_memory), f_dec = v_flux * (1.0f - decay_memory);
        if (V == 1) {
            if (lane < 32) {
                float4 *p = reinterpret_cast<float4 *>(ops + 8 * lane);
                p[0] = make_float4(v_flux, v_energy, e_rise, e_dec);
                p[1] = make_float4(f_rise, f_dec, __int_as_float(bo_l), __int_as_float(v_sup));
            }
            __builtin_amdgcn_wave_barrier();
        }
        auto lane_f = [&](float v, int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)); };
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            float current_flux, frame_energy, er, ed, fr, fd;
            bool bo, sup;
            if (V == 0) {
                current_flux = lane_f(v_flux, i); frame_energy = lane_f(v_energy, i);
                er = lane_f(e_rise, i); ed = lane_f(e_dec, i); fr = lane_f(f_rise, i); fd = lane_f(f_dec, i);
                bo = __builtin_amdgcn_readlane(bo_l, i) != 0;
                sup = __builtin_amdgcn_readlane(v_sup, i) != 0;
            } else {
                const float4 a = reinterpret_cast<const float4 *>(ops + 8 * i)[0], c = reinterpret_cast<const float4 *>(ops + 8 * i)[1];
                current_flux = a.x; frame_energy = a.y; er = a.z; ed = a.w; fr = c.x; fd = c.y;
                bo = __float_as_int(c.z) != 0; sup = __float_as_int(c.w) != 0;
            }
            const bool up = frame_energy > ema;
            ema = ema * (up ? ENERGY_EMA_RISE : ENERGY_EMA_DECAY) + (up ? er : ed);
            const bool is_onset = current_flux > thr;
            thr = thr * (is_onset ? rise_memory : decay_memory) + (is_onset ? fr : fd);
            thr = thr < 0.9f ? 0.9f : thr;
            const bool flux_onset = is_onset && current_flux > thr * multiplier;
            const bool detected = flux_onset && bo;
            const bool rising = frame_energy > ema * 1.5f;
            const bool tick = detected && sup;
            const bool fired = detected && !tick && rising && since >= 3;
            since = (fired || (detected && since < 3)) ? 0 : (since < 0x7fffffff ? since + 1 : since);
            const int fl_bits = (int)flux_onset | ((int)bo << 1) | ((int)detected << 2) | ((int)rising << 3) | ((int)fired << 4) | ((int)tick << 5);
            o_ema = lane == i ? ema : o_ema;
            o_flags = lane == i ? fl_bits : o_flags;
        }
        v_flux = v_flux * 1.0001f + o_ema * 1e-6f;
        v_energy = v_energy * 0.9999f + (float)o_flags * 1e-9f;
        __builtin_amdgcn_s_barrier();
    }
    const long long w1 = wall_clock64(), c1 = __builtin_amdgcn_s_memtime();
    out[lane] = ema + thr + (float)since + o_ema + (float)o_flags;
    if (lane == 0) {
        t[0] = w1 - w0;
        t[1] = c1 - c0;
    }
}

int main()
{
    float *m, *o, h[4096];
    long long *t, ht[2];
    for (int i = 0; i < 4096; ++i)
        h[i] = 0.05f + 0.9f * (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f * ((i % 37) == 0 ? 8.0f : 1.0f);
    hipMalloc(&m, sizeof h);
    hipMalloc(&o, 4096);
    hipMalloc(&t, 16);
    hipMemcpy(m, h, sizeof h, hipMemcpyHostToDevice);
    const int blocks = 14063;                 // x 32 = 450 016 frames: the single-stream onset job of 10 min
    auto report = [&](const char *what) {
        hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost);
        const double sec = ht[0] / 1e8, frames = 32.0 * blocks;
        printf("%-78s %7.2f ms  %6.1f ns / frame  %6.1f cycles / frame (%.0f MHz)\n", what, sec * 1e3, sec / frames * 1e9,
               ht[1] / frames, ht[1] / sec / 1e6);
    };
    for (int rep = 0; rep < 2; ++rep) {
        k_bins<0><<<1, 64>>>(m, o, t, blocks); report("bin floors, as shipped (IEEE quotient in the chain)");
        k_bins<1><<<1, 64>>>(m, o, t, blocks); report("bin floors, decision from an exact f64 product, quotient off the chain");
        k_bins<2><<<1, 64>>>(m, o, t, blocks); report("bin floors, no quotient (chain only)");
        k_bins<3><<<1, 64>>>(m, o, t, blocks); report("bin floors, f32 FMA decision (approximate; timing only)");
        k_chain<0><<<1, 64>>>(m, o, t, blocks); report("scalar chains, operands by v_readlane (as shipped)");
        k_chain<1><<<1, 64>>>(m, o, t, blocks); report("scalar chains, operands by wave-uniform ds_read_b128");
    }
    return 0;
}
