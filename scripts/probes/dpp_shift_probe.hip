// Round 5 probe: does a wave-wide DPP shift (v_mov_b32_dpp wave_shr:1 / wave_shl:1) deliver the neighbour lane's value while ANOTHER PROCESS's
// waves share the CU?  (The neighbour-lane variants of out_conv / fir_down2 / fir_up2 were bit-identical to their load versions alone and failed the
// two-rank drop-in test in 5 of 7 runs: profiles/r05_small_kernels.txt.)  A streaming kernel in the FIR kernels' shape: every lane loads 16 bytes,
// takes its neighbours' edge words by DPP, and compares them with the same words loaded directly; mismatches are counted per (lane mod 16).
// Usage: dpp_shift_probe <seconds> [mode] -- prints launches, compared values, mismatches.  Build: hipcc --offload-arch=gfx950 -O3 (scripts/jobs/j352.sh).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(256) void probe(const float* __restrict__ x, long n4, unsigned long long* __restrict__ bad, float* __restrict__ sink, int mode) {
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (long base = blockIdx.x * 256L; base < n4; base += (long)gridDim.x * 256) {
        const long i0 = base + threadIdx.x;
        const long i = i0 < n4 ? i0 : n4 - 1;  // (whole waves in the body)
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        const float l = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[3]), 0x138, 0xf, 0xf, false));  // lane - 1's last word
        const float r = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[0]), 0x130, 0xf, 0xf, false));  // lane + 1's first word
        const bool has_l = lane > 0 && i0 < n4 && i0 - 1 >= 0, has_r = lane < 63 && i0 + 1 < n4;
        const float wl = has_l ? x[4 * i - 1] : 0.f, wr = has_r ? x[4 * i + 4] : 0.f;
        if (has_l && __float_as_int(l) != __float_as_int(wl)) atomicAdd(bad + (lane & 15), 1ull);
        if (has_r && __float_as_int(r) != __float_as_int(wr)) atomicAdd(bad + 16 + (lane & 15), 1ull);
        acc += l + r + v[1];
        if (mode == 2) {
            // the FIR variants' exact shape: the DPP result is OVERWRITTEN in the seam lanes by a load executed under a partial EXEC mask
            // (every 16th lane and the wave's ends play the seams here), then every lane's value is checked against a full-wave load
            const bool seam_l = lane == 0 || (lane & 15) == 0, seam_r = lane == 63 || (lane & 15) == 15;
            float l2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[3]), 0x138, 0xf, 0xf, false));
            float r2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[0]), 0x130, 0xf, 0xf, false));
            const long il = 4 * i - 1 >= 0 ? 4 * i - 1 : 0, ir = 4 * i + 4 < 4 * n4 ? 4 * i + 4 : 0;
            if (seam_l) l2 = x[il];
            if (seam_r) r2 = x[ir];
            const float fl = x[il], fr = x[ir];
            asm volatile("" : "+v"(l2), "+v"(r2));
            if (i0 < n4 && i0 > 0 && __float_as_int(l2) != __float_as_int(fl)) atomicAdd(bad + (lane & 15), 1ull);
            if (i0 + 1 < n4 && __float_as_int(r2) != __float_as_int(fr)) atomicAdd(bad + 16 + (lane & 15), 1ull);
            acc += l2 + r2;
        }
    }
    if (acc == 12345.678f) *sink = acc;  // (keeps the arithmetic)
}

// mode 3: is the SCALAR data cache isolated between processes?  Every wave re-reads, through scalar loads (constant address space: s_load), a table this
// process filled with ITS tag; a second instance of this program holds another tag in a table that very likely sits at the same virtual address.  A scalar
// load that returns the other process's words is counted.  (Kernel arguments travel the same way: s_load from the kernarg segment.)
__global__ __launch_bounds__(256) void kprobe(const unsigned* __restrict__ tab, int n, unsigned tag, unsigned long long* __restrict__ bad, int reps) {
    using cu = const unsigned __attribute__((address_space(4)))*;
    const cu t = (cu)tab;
    unsigned long long wrong = 0;
    for (int r = 0; r < reps; ++r) {
        const int i = (int)((blockIdx.x * 7u + r * 13u) % (unsigned)n);
        const unsigned v = t[i];  // wave-uniform index: a scalar load
        if (v != (tag ^ (unsigned)i)) ++wrong;
        __builtin_amdgcn_s_sleep(2);
        asm volatile("s_dcache_inv" ::: "memory");  // (each repetition a fresh fetch through the shared scalar cache hierarchy)
    }
    if (wrong && (threadIdx.x & 63) == 0) atomicAdd(bad, wrong);
}

// mode 4: v_cndmask_b32 with an SGPR-PAIR lane mask (VOP3) in a hot loop -- what the branch-free fir_up2 body is made of and the committed one is not.  Every lane
// checks the select against the same choice made with integer arithmetic; mismatches are counted per 16-lane quarter of the wave (the corrupted outputs of the
// rewritten kernel sit in lanes 48-63).
__global__ __launch_bounds__(256) void cprobe(const float* __restrict__ x, long n, unsigned long long* __restrict__ bad, float* __restrict__ sink, int reps) {
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (long base = blockIdx.x * 256L; base < n; base += (long)gridDim.x * 256) {
        const long i = base + threadIdx.x < n ? base + threadIdx.x : n - 1;
        const float a = x[i], b = x[(i * 7 + 3) % n];
        for (int r = 0; r < reps; ++r) {
            const bool c = ((__float_as_uint(a) >> (r & 15)) ^ (unsigned)(lane >> 2) ^ (unsigned)r) & 1u;
            const unsigned long long m = __ballot(c);  // the condition as an SGPR pair
            float sel;
            asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(sel) : "v"(a), "v"(b), "s"(m));
            const unsigned want = c ? __float_as_uint(b) : __float_as_uint(a);
            if (__float_as_uint(sel) != want) atomicAdd(bad + (lane >> 4), 1ull);
            acc += sel;
        }
    }
    if (acc == 12345.678f) *sink = acc;
}

// mode 5: the branch-free fir_up2 body's LOADS (three rows from a clamped row index: one 8-byte and two 4-byte loads each) checked word by word against the
// pattern the buffer holds (x[k] = pat(k)); mode 6: the same loads, then the body's ARITHMETIC done twice from the same registers (selects on the row-inside-the-
// image predicate, the [1,3]/4 chains) and the two results compared.  Mismatches by 16-lane quarter of the wave.
__device__ __forceinline__ unsigned pat(long k) { return (unsigned)(k * 2654435761u) >> 9 | 0x3f800000u; }  // a float in [1, 2)
__global__ __launch_bounds__(256) void fprobe(const float* __restrict__ x, int C, int H, int W, unsigned long long* __restrict__ bad, float* __restrict__ sink, int mode) {
    const int Wh = W >> 1, lane = threadIdx.x & 63;
    const long per_plane = (long)H * Wh, total = per_plane * C;
    float keep = 0.f;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx / per_plane;
        const long rem = idx % per_plane;
        const int i = rem / Wh, t = rem % Wh;
        const float* xp = x + (long)c * H * W;
        const int cl = 2 * t - 1 < 0 ? W - 1 : 2 * t - 1, cr = 2 * t + 2 >= W ? 0 : 2 * t + 2;
        float h[2][3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int r = i + a - 1;
            const bool ok = r >= 0 && r < H;
            const float* row = xp + (long)(ok ? r : i) * W;
            const float2 m = *reinterpret_cast<const float2*>(row + 2 * t);
            const float l = row[cl], rr = row[cr];
            if (mode == 5) {
                const long k0 = (row - x) + 2 * t;
                int w = 0;
                w += __float_as_uint(m.x) != pat(k0);
                w += __float_as_uint(m.y) != pat(k0 + 1);
                w += __float_as_uint(l) != pat((row - x) + cl);
                w += __float_as_uint(rr) != pat((row - x) + cr);
                if (w) atomicAdd(bad + (lane >> 4), (unsigned long long)w);
            }
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                float mx = m.x, my = m.y, ll = l, rq = rr;
                asm volatile("" : "+v"(mx), "+v"(my), "+v"(ll), "+v"(rq));  // (two independent evaluations of the same registers)
                h[rep][a][0] = ok ? ll * 0.25f + mx * 0.75f : 0.f;
                h[rep][a][1] = ok ? mx * 0.75f + my * 0.25f : 0.f;
                h[rep][a][2] = ok ? mx * 0.25f + my * 0.75f : 0.f;
                h[rep][a][3] = ok ? my * 0.75f + rq * 0.25f : 0.f;
            }
        }
        int w2 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float e[2], o[2];
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                e[rep] = h[rep][0][j] * 0.25f + h[rep][1][j] * 0.75f;
                o[rep] = h[rep][1][j] * 0.75f + h[rep][2][j] * 0.25f;
            }
            w2 += __float_as_uint(e[0]) != __float_as_uint(e[1]);
            w2 += __float_as_uint(o[0]) != __float_as_uint(o[1]);
            keep += e[0] + o[1];
        }
        if (mode == 6 && w2) atomicAdd(bad + 4 + (lane >> 4), (unsigned long long)w2);
    }
    if (keep == 12345.678f) *sink = keep;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 10.0;
    const int mode = argc > 2 ? atoi(argv[2]) : 1;  // 2: + the seam-overwrite shape
    const long n = 32L << 20;  // 128 MB of floats: a level-1 tensor
    std::vector<float> h(n);
    unsigned s = 12345u;
    for (long i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) * (1.0f / 16777216.0f); }
    float *x, *sink;
    unsigned long long* bad;
    hipMalloc(&x, n * 4); hipMalloc(&sink, 4); hipMalloc(&bad, 32 * 8);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 32 * 8);
    if (mode == 3) {
        const unsigned tag = argc > 3 ? (unsigned)strtoul(argv[3], nullptr, 0) : 0x12345678u;
        const int nt = 1 << 16;
        std::vector<unsigned> ht(nt);
        for (int i = 0; i < nt; ++i) ht[i] = tag ^ (unsigned)i;
        unsigned* tab;
        hipMalloc(&tab, nt * 4);
        hipMemcpy(tab, ht.data(), nt * 4, hipMemcpyHostToDevice);
        long kl = 0;
        const auto k0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - k0).count() < secs) {
            for (int k = 0; k < 20; ++k) kprobe<<<4096, 256>>>(tab, nt, tag, bad, 64);
            if (hipDeviceSynchronize() != hipSuccess) { printf("dpp_shift_probe: launch failed\n"); return 2; }
            kl += 20;
        }
        unsigned long long hb0 = 0;
        hipMemcpy(&hb0, bad, 8, hipMemcpyDeviceToHost);
        printf("scalar_cache_probe: tag 0x%08x table at %p, %ld launches x 4096 blocks x 4 waves x 64 scalar loads, wrong words %llu\n", tag, (void*)tab, kl, hb0);
        return hb0 ? 1 : 0;
    }
    if (mode == 5 || mode == 6) {
        const int C = 128, H = 32, W = 512;  // batch 2 x 64 planes of the 32 x 512 up-sampler as 128 planes
        for (long k = 0; k < (long)C * H * W; ++k) { const unsigned u = (unsigned)(k * 2654435761u) >> 9 | 0x3f800000u; memcpy(&h[k], &u, 4); }
        hipMemcpy(x, h.data(), (long)C * H * W * 4, hipMemcpyHostToDevice);
        long kl = 0;
        const auto k0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - k0).count() < secs) {
            for (int k = 0; k < 20; ++k) fprobe<<<4096, 256>>>(x, C, H, W, bad, sink, mode);
            if (hipDeviceSynchronize() != hipSuccess) { printf("dpp_shift_probe: launch failed\n"); return 2; }
            kl += 20;
        }
        unsigned long long hq[8];
        hipMemcpy(hq, bad, sizeof hq, hipMemcpyDeviceToHost);
        printf("fir_body_probe mode %d: %ld launches of %d items; wrong LOADED words by lane quarter: %llu %llu %llu %llu; evaluations that DISAGREE by lane quarter: %llu %llu %llu %llu\n", mode, kl,
               C * H * (W / 2), hq[0], hq[1], hq[2], hq[3], hq[4], hq[5], hq[6], hq[7]);
        return 0;
    }
    if (mode == 4) {
        long kl = 0;
        const auto k0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - k0).count() < secs) {
            for (int k = 0; k < 20; ++k) cprobe<<<4096, 256>>>(x, 1L << 22, bad, sink, 32);
            if (hipDeviceSynchronize() != hipSuccess) { printf("dpp_shift_probe: launch failed\n"); return 2; }
            kl += 20;
        }
        unsigned long long hq[4];
        hipMemcpy(hq, bad, sizeof hq, hipMemcpyDeviceToHost);
        printf("cndmask_probe: %ld launches, %.3g selects checked, mismatches by lane quarter: %llu %llu %llu %llu\n", kl, (double)kl * (1L << 22) * 32, hq[0], hq[1], hq[2], hq[3]);
        return (hq[0] + hq[1] + hq[2] + hq[3]) ? 1 : 0;
    }
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int k = 0; k < 20; ++k) probe<<<8192, 256>>>(x, n / 4, bad, sink, mode);
        if (hipDeviceSynchronize() != hipSuccess) { printf("dpp_shift_probe: launch failed\n"); return 2; }
        launches += 20;
    }
    unsigned long long hb[32];
    hipMemcpy(hb, bad, sizeof hb, hipMemcpyDeviceToHost);
    unsigned long long tl = 0, tr = 0;
    for (int i = 0; i < 16; ++i) { tl += hb[i]; tr += hb[16 + i]; }
    printf("dpp_shift_probe: %ld launches, %.3g neighbour words compared, mismatches wave_shr %llu wave_shl %llu  by lane %% 16 (shr):", launches, (double)launches * (n / 4) * 2, tl, tr);
    for (int i = 0; i < 16; ++i) printf(" %llu", hb[i]);
    printf("\n");
    return (tl + tr) ? 1 : 0;
}
