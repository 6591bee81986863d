// Calibration probe (not product): cycles each extra instruction adds to a saturated v_mfma_f32_32x32x2_f32 stream
// of the SAME wave (1 wave per SIMD), for the instruction kinds the conv staging/epilogue uses.  gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int OP>
__device__ __forceinline__ void op(float (&v)[8], double (&d)[4], f32x2 (&pk)[4], f32x4 (&q)[2], const float* gp, float a0, float b0, int j, float* lds, int tid) {
    int sdummy;
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(a0), "v"(b0));
    if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[j & 3]) : "v"(pk[(j + 1) & 3]), "v"(pk[(j + 2) & 3]));
    if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
    if (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[j & 7]));
    if (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[j & 3]) : "v"(d[(j + 1) & 3]), "v"(d[(j + 2) & 3]));
    if (OP == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[j & 3]) : "v"(v[j & 7]));
    if (OP == 6) asm volatile("ds_write_b32 %0, %1" ::"v"(tid * 4), "v"(v[j & 7]) : "memory");
    if (OP == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j & 7]) : "v"(a0));
    if (OP == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(a0));
    if (OP == 9) asm volatile("s_nop 0");
    if (OP == 10) asm volatile("ds_read_b32 %0, %1" : "=v"(v[j & 7]) : "v"(tid * 4) : "memory");
    if (OP == 11) asm volatile("ds_read_b64 %0, %1" : "=v"(pk[j & 3]) : "v"(tid * 8) : "memory");
    if (OP == 12) asm volatile("ds_read_b128 %0, %1" : "=v"(q[j & 1]) : "v"(tid * 16) : "memory");
    if (OP == 13) asm volatile("ds_write_b128 %0, %1" ::"v"(tid * 16), "v"(q[j & 1]) : "memory");
    if (OP == 14) asm volatile("ds_write_b64 %0, %1" ::"v"(tid * 8), "v"(pk[j & 3]) : "memory");
    if (OP == 15) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(a0));
    if (OP == 16) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[j & 3]) : "v"(pk[(j + 1) & 3]));
    if (OP == 17) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[j & 1]) : "v"(gp) : "memory");
    if (OP == 18) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sdummy) : "v"(v[j & 7]));
    if (OP == 19) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v[j & 7]) : "v"(tid * 4), "v"(v[(j + 1) & 7]) : "memory");
    if (OP == 20) asm volatile("v_exp_f32 %0, %1" : "=v"(v[j & 7]) : "v"(a0));
}

template <int NV, int OP>
__global__ __launch_bounds__(256) void probe(const float* in, float* out, int iters) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a0 = in[tid], a1 = in[tid + 256], b0 = in[tid + 512], b1 = in[tid + 768];
    float v[8]; double d[4]; f32x2 pk[4]; f32x4 q[2] = {f32x4{a0, a1, b0, b1}, f32x4{a1, a0, b1, b0}}; const float* gp = in + tid * 4;
    for (int j = 0; j < 8; ++j) v[j] = in[tid + j];
    for (int j = 0; j < 4; ++j) { d[j] = in[tid + j]; pk[j] = f32x2{in[tid + j], in[tid + j + 1]}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) op<OP>(v, d, pk, q, gp, a0, b0, j, lds, tid);
            if (OP >= 10) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
    }
    float s = lds[tid];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    for (int j = 0; j < 4; ++j) s += (float)d[j] + pk[j][0] + pk[j][1];
    s += q[0][0] + q[1][3];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NV, int OP>
float run(const float* in, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256, iters = 256;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NV, OP>), dim3(grid), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    return best;
}

template <int OP>
void line(const char* name, const float* in, float* out, float base) {
    const float t8 = run<8, OP>(in, out), t16 = run<16, OP>(in, out);
    // per iteration-step: 4 MFMA (256 cycles) + NV ops; 18*256 steps
    const double steps = 18.0 * 256;
    const double cyc_per_ms = 256.0 * steps / base;   // calibrate the clock from the bare stream (64 cycles/MFMA)
    printf("%-16s +8: %.1f cycles/op   +16: %.1f cycles/op   (bare %.3f ms, +8 %.3f, +16 %.3f)\n", name,
           (t8 - base) * cyc_per_ms / steps / 8, (t16 - base) * cyc_per_ms / steps / 16, base, t8, t16);
}

int main() {
    float *in, *out; hipMalloc(&in, 1 << 20); hipMalloc(&out, 64 << 20); hipMemset(in, 0, 1 << 20);
    const float base = run<0, 0>(in, out);
    line<0>("v_fma_f32", in, out, base); line<1>("v_pk_fma_f32", in, out, base); line<2>("v_exp_f32", in, out, base);
    line<3>("v_rcp_f32", in, out, base); line<4>("v_fma_f64", in, out, base); line<5>("v_cvt_f64_f32", in, out, base);
    line<6>("ds_write_b32", in, out, base); line<7>("v_mov_b32", in, out, base); line<8>("v_add_u32", in, out, base);
    line<9>("s_nop", in, out, base);
    line<10>("ds_read_b32", in, out, base); line<11>("ds_read_b64", in, out, base); line<12>("ds_read_b128", in, out, base);
    line<13>("ds_write_b128", in, out, base); line<14>("ds_write_b64", in, out, base); line<15>("v_mul_f32", in, out, base);
    line<16>("v_pk_mul_f32", in, out, base); line<17>("global_load_x4", in, out, base); line<18>("v_readlane", in, out, base);
    line<19>("ds_bpermute", in, out, base); line<20>("v_exp (indep)", in, out, base);
    return 0;
}
