// K2f: 3x3 ring convolution on the fp16 matrix pipe with fp32-class accuracy ("f16x2 split"), warp-specialised and
// persistent: 4 MFMA waves + 4 staging waves per CU.
//
// Contract and data layout are those of conv_bf16x3.hip (reference ops.Conv2d + ops.Pad,
// /root/reference/models/ops.py:32-49,149-173; fused GroupNorm-affine + SiLU prologue and bias / residual / scale /
// GroupNorm-statistics epilogue of /root/reference/models/efficient_unet.py:95-110).  The arithmetic is new in round 2:
//
//   every fp32 operand v is split EXACTLY to 22 bits as  v = h + 2^-11 l,  h = RNE_f16(v),  l = RNE_f16(2^11 (v - h)),
//   and a product becomes  x w = xh wh + 2^-11 (xh wl + xl wh) + O(2^-22):  THREE v_mfma_f32_32x32x16_f16 instead of the
//   six bf16 products of the three-piece bf16 split.  The xh wh products go to one fp32 accumulator, the two cross
//   products to a second one that is scaled by 2^-11 once, in the epilogue.
//
// Measured on the matrix pipe (scripts/probes/f16x2_probe.hip, profiles/r02_f16x2_probe.txt): relative rms error
// 1.7e-7 / 2.3e-7 / 4.3e-7 at K = 576 / 1152 / 4608 against 3.6e-7 / 5.4e-7 / 1.1e-6 for the bf16 three-piece split and
// 4.4e-7 / 6.0e-7 / 1.2e-6 for the fp32 MFMA -- the big accumulator takes one (truncating) update per tap and chunk
// instead of six, and the mean signed error is ten times smaller, so the sign-alternation trick of conv_bf16x3.hip is
// not needed.  Why it matters: the bf16x3 kernels run into the BOARD POWER LIMIT (1400 W, shader clock 1.83-1.89 GHz,
// profiles/r02_power_clock.txt) -- as does hipBLASLt's bf16 GEMM -- so the only way to more images per second is fewer
// matrix-pipe operations per product.
//
// Range: fp16 tops out at 65504.  The kernel runs with MODE.FP16_OVFL set (conversions saturate instead of producing
// inf), and the engine only routes convolutions here whose input range is guarded from the data: GroupNorm-normalised
// inputs by gn_finalize's bound |a| M + |d| (M >= max|x| from the statistics slots, norm.hip), raw inputs by their producer's
// recorded max|output|; a violation raises the engine's range flag and the forward fails loudly (engine.hip).  Weights are
// scaled per layer by a power of two when they are packed (max|w| into [2^9, 2^10): tiny weights keep their 22 bits, none can
// leave the range); the inverse scale (ConvParams::wscale) multiplies the matrix product in the epilogue -- exact.
//
// Who does what (as in the round-2 timeline analysis: beside a running MFMA stream every other instruction costs issue
// slots in whichever wave it sits, so the multiplying waves carry nothing else):
//   waves 0..3  ("multipliers", one per SIMD): a pure MFMA stream -- per tap 12 MFMAs and the 8 ds_read_b128 of the next
//               tap's fragments; no global memory instruction, no transform, no LDS-DMA.  The tile's epilogue.
//   waves 4..7  ("stagers", one per SIMD): raw pixel loads two chunks ahead (inline assembly, explicit vmcnt), affine +
//               SiLU + the f16 split of the next chunk into the other x buffer, the weight stages by LDS-DMA into a
//               4-slot ring (three segments of flight per stage), the addresses of the block's next tile.
// Three block-wide barriers per chunk (one per weight stage) couple the two groups.  The block is persistent (one per
// CU) and walks its share of the launch's (output channel tile, pixel tile) pairs; the stagers run ahead across tile
// boundaries, so a tile's epilogue overlaps the staging of the next tile's first chunks.
//
// Round 4: a third template parameter.  MRK = 2 is the kernel described above (64-channel output tiles, two accumulators);
// MRK = 4 gives a multiplying wave four 32-channel row blocks -- 128-channel output tiles, 24 MFMAs and 12 fragment reads per tap,
// half the staging work per MFMA -- with ONE accumulator for all three products (residual planes at their true scale, a weight
// packing of its own): below an fp32 fmaf chain's rms error at every depth; conv_f16x2_pick_co_tile uses it up to Cin = 256.
// PRO_PRESPLIT: the input arrives pre-split (presplit.hip) and the stagers only issue LDS-DMA -- an experiment, off by default.
#include "common.h"
#include "conv_bf16x3.h"
#include "conv_epilogue.h"
#include "f16x2.h"
#include "gn_math.h"
#include <stdlib.h>
#include <string.h>

namespace r2dm {

namespace f2 {
using namespace x3;  // CO_T 64, TW 64, NG 2, CK 16 (TH / XR / MR / NR of x3 are the (2, 2) tile's: the kernel shadows them per tile)
constexpr int NPL = 2;                                      // planes: h, l
constexpr int XS = 67;                                      // 66 columns + 1 dump column (never read)
constexpr int ADTAB_BYTES = 4096;                           // (a, d) of the current tile's sample, all Cin channels: Cin <= 512
// A multiplying wave owns MRK 32-channel row blocks x NRK 32-pixel segments (two segments per image row, four waves):
//   (MRK, NRK) = (2, 2): 64 channels x 4 rows x 64 pixels, two accumulators per MFMA tile (round 2)
//                (4, 2): 128 channels x 4 rows, ONE accumulator (round 4: half the staging work and x-fragment reads per MFMA)
//                (2, 4): 64 channels x 8 rows, ONE accumulator (round 5, the "tall" tile: what the 128-channel tile is to layers with
//                        >= 128 output channels, for the layers with 64: 24 MFMAs per tap between two barriers, a halo of 10 / 8 instead
//                        of 6 / 4 rows, one tile end per 512 pixels)
//                (1, 2): 32 channels x 4 rows, two accumulators (round 5: twice the tiles where a launch has fewer 64-channel tiles than the chip
//                        has CUs -- u_block4 ran on 128 of 256 CUs; same x tile and stagers, half the MFMAs per chunk: stager-bound)
template <int MRK, int NRK>
struct Geo {
    static constexpr int COT = 32 * MRK;                        // output channels per tile
    static constexpr int TH = 2 * NRK;                          // image rows per tile
    static constexpr int XR = TH + 2;                           // ... and of its input tile
    static constexpr int XPL = NG * XR * XS;                    // 16-byte entries per plane
    static constexpr int XBYTES = NPL * XPL * 16;               // 25728 (four rows) | 42880 (eight) | 17152 (two)
    static constexpr int NQ = MRK * NRK;                        // 32 x 32 quarters of a multiplying wave's tile
    static constexpr bool ONEACC = NQ == 8;                     // all three products into one accumulator (l planes at their true scale)
    static constexpr int WSTAGE = NPL * 3 * NG * COT * 16;      // 12288 / 24576: one kernel row of one chunk
    static constexpr int RING = ONEACC ? 3 : 4;                 // weight stages in LDS
    static constexpr int WB0 = 2 * XBYTES;                      // [x buffer 0][x buffer 1][weight ring][epilogue patches][finished tile][(a, d) table]
    static constexpr int PATCH0 = WB0 + RING * WSTAGE;
    static constexpr int RESQ = ONEACC ? 2 : NQ - 1;            // quarters of a finished tile that wait in LDS for their deferred epilogue (per wave)
    static constexpr int RES0 = PATCH0 + (ONEACC ? 0 : 4 * 1024);  // four waves x RESQ x 4 KiB (one-accumulator tiles: no separate patches -- the turn of the
                                                                // immediate quarters goes through the wave's first waiting slot, which is empty at a tile's end)
    static constexpr int ADTAB0 = RES0 + 4 * RESQ * 4096;
    static constexpr int BIAS0 = ADTAB0 + ADTAB_BYTES;          // one-accumulator tiles: the biases of the block's current tiles, two slots of COT floats (by tile parity)
    static constexpr int LDS_TOTAL = BIAS0 + (ONEACC ? 2 * COT * 4 : 0);  // 157952 (2, 2) | 163072 (4, 2) | 160000 (2, 4)
    static constexpr int UNITS_X = NG * XR * 18;                // staging units of a chunk: 8 channels x 4 pixels of one row (16 interior quads + 2 halo columns per row)
    static constexpr int NU = (UNITS_X + 255) / 256;            // ... per staging thread (1 | 2)
    static_assert(LDS_TOTAL <= 160 * 1024, "LDS");
};
}  // namespace f2

// Reciprocals for the tile decode (host: f2_magic): x / d == __umulhi(x, m) exactly for x * d < 2^32, m = floor(2^32 / d) + 1
// (d == 1: m = 0, x / 1 = x).  The compiler's own signed 32-bit division is ~25 scalar instructions per quotient; a tile is
// decoded four times (load cursor, weight cursor, table, epilogue), three of them by the stagers that bound the chunk time.
struct F2Div {
    unsigned co, tw, th;
};
static unsigned f2_magic(int d) { return d == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned long long)d) + 1u; }

// NPLK = operand planes in use: 2 = the split arithmetic above (three products per fp32 product: the parity path);
// 1 = the h plane alone, ONE fp16 product per MAC with fp32 accumulation -- the reduced-precision bulk mode that mirrors the
// reference's fp16 autocast sampler (/root/reference/sample_and_save.py:70, utils/option.py:49): same packing, same tiles,
// the l plane is neither fetched, computed nor multiplied.
// IOM (round 5, the one-plane mode only): activations stored as fp16 in HBM, as the reference's autocast stores its convolution outputs
// (/root/reference/sample_and_save.py:45,70) -- bit 0: the input tensor is fp16, bit 1: the output and the residual are.  Everything in
// between stays what it is: fp32 GroupNorm affine + SiLU on the way in, fp32 accumulation, fp32 bias / residual / scale, statistics of
// the values as stored.
template <int PRO, int NPLK, int MRK, int NRK, int IOM = 0>
__global__ __launch_bounds__(512, 2) void conv_f16x2_kernel(const ConvParams p, const int total_tiles, const F2Div dv) {
    using namespace f2;
    static_assert(NPLK == 1 || NPLK == 2, "planes");
    static_assert(IOM == 0 || (NPLK == 1 && PRO != PRO_PRESPLIT), "fp16 storage: the one-plane mode");
    constexpr bool X16 = (IOM & 1) != 0, Y16 = (IOM & 2) != 0;
    constexpr int XE = X16 ? 2 : 4, YE = Y16 ? 2 : 4;  // bytes per stored element
    using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
    static_assert((MRK == 2 && (NRK == 2 || NRK == 4)) || (MRK == 4 && NRK == 2) || (MRK == 1 && NRK == 2), "tile family");
    using GEO = Geo<MRK, NRK>;
    // (the tile's own geometry shadows x3's constants of the same names)
    constexpr int MR = MRK, NR = NRK, TH = GEO::TH, XR = GEO::XR, XPL = GEO::XPL, XBYTES = GEO::XBYTES, NQ = GEO::NQ, NU = GEO::NU;
    constexpr int COT = GEO::COT, WSTAGE = GEO::WSTAGE, RING = GEO::RING, WB0 = GEO::WB0, PATCH0 = GEO::PATCH0, RESQ = GEO::RESQ,
                  RES0 = GEO::RES0, ADTAB0 = GEO::ADTAB0, BIAS0 = GEO::BIAS0;
    constexpr bool ONEACC = GEO::ONEACC;
    // Round 5, the tile end of the one-accumulator tiles (-DF2_EPI_V1: round 4's, for A/B): biases wait in an LDS table the stagers fill
    // three chunks ahead (requested at the tile's end they cost its first quarter ~1.9 k cycles), the residual of the first two quarters is
    // prefetched by LDS-DMA during the tile's last chunk (no registers: requested at the tile's end the first quarters waited for HBM, 10 k
    // cycles per tile of the eight-row tile), and all eight quarters are finished at the tile's end (no row block waits in LDS: the
    // multipliers bound a chunk of these tiles, a deferred quarter costs what it takes wherever it runs).
#ifdef F2_EPI_V1
    constexpr bool EPI2 = false;
#else
    constexpr bool EPI2 = ONEACC;
#endif
    // Round 6, the tile end of the one-accumulator tiles SHARED between the two wave groups (-DF2_NO_EPI3: round 5's, for A/B).  In-kernel timelines
    // (profiles/r06_level1_timeline.txt): a tile end is 7 k cycles of the four multiplying waves' own instruction stream (8 quarters x ~0.85 k: turn, bias /
    // residual / scale, store, statistics) -- 11-14 k with a residual, whose loads a wave can only keep four quarters of in flight -- while the four staging
    // waves, a tile ahead, sleep at their next barrier (16 k cycles per tile end in the level-1 residual launches).  Now a multiplying wave turns its last
    // NSQ quarters into LDS, the block meets at a barrier, and its staging partner (same wave index) finishes those -- bias from the LDS table, its own
    // residual loads, stores, statistics -- while the multiplier finishes the first NQ - NSQ; a second barrier ends the tile.  LDS: the wave's two waiting
    // slots (their DMA-prefetched residual is in registers by then) and, for the eight-row tile's third and fourth quarter, x buffer 1, which is dead from
    // the tile's last barrier until the stagers' next transform -- behind the second barrier; the multipliers' 1 KiB turn patches move there too.
#ifdef F2_NO_EPI3
    constexpr bool EPI3 = false;
#else
    constexpr bool EPI3 = EPI2 && IOM == 0 && PRO != PRO_PRESPLIT;
#endif
    constexpr int NSQ = !EPI3 ? 0 : NR == 4 ? 4 : 2;  // quarters (the last ones: whole statistics pairs) the staging partner finishes
    constexpr int XDUMP = NSQ > 2 ? NSQ - 2 : 0;      // ... of which so many wait in x buffer 1: [4 x 1 KiB patches][4 waves x XDUMP x 4 KiB]
    static_assert(4096 + 4 * XDUMP * 4096 <= XBYTES, "tile-end scratch inside x buffer 1");
    // the cross products go to a second accumulator (scaled l planes) or, in the one-accumulator tiles, to the same one (l planes at their true scale)
    constexpr bool ACC2 = NPLK == 2 && !ONEACC;
    constexpr bool LSCALED = !ONEACC;
    // MRK == 2 -- NPLK == 2: 12 DMA pieces of 1 KiB per stage, three per stager.  NPLK == 1: the h
    // plane is the first 6 pieces of a stage: stager w fetches pieces w and w + 2 (pieces 2 and 3 twice: harmless).
    // MRK == 4 -- NPLK == 2: 24 pieces per stage, six per stager; NPLK == 1: the h plane = 12 pieces, three per stager
    // MFMAs per tap: 3 MR NR with both planes (6 | 12 | 24), MR NR with one
    constexpr int UNITS = (NPLK == 2 ? 3 : 1) * MR * NR, PPW = MRK == 1 ? (NPLK == 2 ? 2 : 1) : MRK == 2 ? (NPLK == 2 ? 3 : 2) : (NPLK == 2 ? 6 : 3);
    // (MRK == 1: a stage is 6 pieces -- both planes of 32 channels --, fetched like the h plane of the 64-channel tile: pieces w and w + 2)
    // A weight stage must have landed when the barrier in front of its first read is reached; so many younger operations of the
    // stager may still be in flight then (vmcnt retires in order; the queue holds loads only).  Per iteration the queue is
    //   D0 [PPW] | #1 | D1 [PPW] | #2 | D2 [PPW], raw [NL] | #3        NL = 8 global loads per staging unit of the iteration
    // RING == 4: the stage due at #1 is D1 of the previous iteration, at #2 its D2, at #3 this iteration's D0 -- 2 PPW + NL' younger each time
    // (NL': the previous iteration's pixel loads).
    // RING == 3 (one segment less of flight): due at #1 is the previous D2 (raw', D0 younger), at #2 D0 (D1), at #3 D1 (D2, raw).
    static_assert(RING == 4 || RING == 3, "ring depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)smem;

    // A block OWNS its CU (round 5): every wave allocates all 256 registers (a clobber of v255 costs nothing else -- occupancy is one block
    // per CU either way) and the launcher pads the LDS request to >= 156 KiB, so no wave of any other kernel or process is ever resident on
    // the same CU.  Why: next to ANOTHER PROCESS's waves on its CU the one-plane 32-channel tile (181 registers) ended 12-17 of 20 runs of
    // 600 forwards in a GPU memory fault at a wild address (never a wrong result); with the CU to itself 0 of 70, all modes
    // (profiles/r05_coresidency.txt: what was ruled out, and that the cause is NOT identified).  -DF2_SHARE_CU: the allocation hipcc computes.
#ifndef F2_SHARE_CU
    asm volatile("" ::: "v255");
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool helper = wave8 >= 4;  // (the multipliers are the older waves: instruction issue is arbitrated by age)
    const int wave = wave8 & 3, t4 = tid & 255;

    const int H = p.H, W = p.W;
    const int HW = H * W;
    const int nTw = W / TW, nTh = H / TH;  // whole tiles only (launcher)
    const int nCoT = p.Cout / COT;
    const int nchunks = p.Cin / CK, nst = 3 * nchunks;
    const int G = gridDim.x;
    const int nIt = (total_tiles - (int)blockIdx.x + G - 1) / G;  // tiles of this block (grid <= total_tiles)
    const int Q = nIt * nchunks;
    const int c0 = p.x.c0;
    if (p.stagger > 0 && nIt >= 3 && ((blockIdx.x >> 3) & 1)) {  // (experiment: half of the blocks out of phase with the other half)
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)p.stagger) __builtin_amdgcn_s_sleep(32);
    }

    // tile `it` of this block -> output channel tile, sample, tile row / column (all wave-uniform: scalar ALU, F2Div)
    auto udiv = [&](unsigned x, int d, unsigned m) __attribute__((always_inline)) { return d == 1 ? x : __umulhi(x, m); };
    auto decode = [&](int it, int& cot, int& b, int& th, int& tw) __attribute__((always_inline)) {
        unsigned L = (unsigned)__builtin_amdgcn_readfirstlane(xcd_remap((int)blockIdx.x + it * G, total_tiles));
        if (p.reverse) L = (unsigned)total_tiles - 1u - L;
        const unsigned t1 = udiv(L, nCoT, dv.co), t2 = udiv(t1, nTw, dv.tw), t3 = udiv(t2, nTh, dv.th);
        cot = (int)(L - t1 * (unsigned)nCoT);
        tw = (int)(t1 - t2 * (unsigned)nTw);
        th = (int)(t2 - t3 * (unsigned)nTh);
        b = (int)t3;
    };
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.w);
#ifdef F2_PROF  // timeline probe (scripts/f2_timeline.py): multiplier wave 0 and stager wave 4 of block 0 stamp s_memtime
    int prof_i = 0;
    auto stamp = [&](int code) __attribute__((always_inline)) {
        if (p.prof && blockIdx.x == 0 && wave == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0 && prof_i < 1020) p.prof[(helper ? 1024 : 0) + prof_i] = (t << 8) | (unsigned)code;
            ++prof_i;
        }
    };
    auto stamp_real = [&](int code) __attribute__((always_inline)) {  // constant 100 MHz clock next to the shader clock
        if (p.prof && blockIdx.x == 0 && wave == 0 && !helper) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime(), u = __builtin_amdgcn_s_memtime();
            if (lane == 0) { p.prof[2048 + 2 * code] = t; p.prof[2048 + 2 * code + 1] = u; }
        }
    };
    stamp_real(0);
#else
    auto stamp = [&](int) __attribute__((always_inline)) {};
    auto stamp_real = [&](int) __attribute__((always_inline)) {};
#endif

    if (helper) {
        // ============================================= staging waves =============================================
        __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);  // MODE.FP16_OVFL: f16 conversions saturate at +-65504
        // ---- weights: LDS-DMA cursor (stage within the tile's co tile; tiles may change co tile) ----
        // 12 pieces of 1 KiB per stage; stager w issues pieces 3w, 3w+1, 3w+2.  Everything but the lane offset is wave-uniform:
        // source and LDS base go through SGPRs.
        const unsigned lane16 = (unsigned)lane * 16;
        int d_item = 0, d_s = 0;
        const unsigned char* d_base = nullptr;
        auto set_dma_item = [&](int it) __attribute__((always_inline)) {
            int cot, b, th, tw;
            decode(it, cot, b, th, tw);
            d_base = wsrc + (size_t)cot * nst * WSTAGE;
        };
        // the cursor's stage -> its ring slot (stage number q3 + K, q3 a multiple of 3: with RING == 3 the slot is K % 3 statically); this
        // wave's PPW pieces; advances the cursor (past the end: the last stage again -- harmless, keeps the vmcnt bookkeeping uniform)
        auto dma_stage = [&](int q3, auto KK) __attribute__((always_inline)) {
            constexpr int K = decltype(KK)::value;
            const unsigned l16 = lane16, l0 = lds0;  // (odr-used here: clang does not capture variables that a generic lambda only names inside if constexpr)
            const int wv = wave;
            const int slot = RING == 4 ? ((q3 + K) & 3) : K % 3;
            const unsigned long long sv = (unsigned long long)(d_base + (size_t)d_s * WSTAGE);  // wave-uniform: say so (SGPR operand)
            const unsigned char* src = (const unsigned char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sv >> 32)) << 32) |
                                                              (unsigned)__builtin_amdgcn_readfirstlane((int)sv));
            if (d_s + 1 < nst)
                ++d_s;
            else if (d_item + 1 < nIt) {
                d_s = 0;
                set_dma_item(++d_item);
            }
#ifdef F2_NO_DMA  // timing ablation (wrong results)
            if (slot >= 0) return;
#endif
            // one M0 set-up per group of pieces, the instruction offset advances the global and the LDS address together
            if constexpr (PPW == 3) {  // pieces 3w, 3w+1, 3w+2 (64-channel tiles: the whole stage; 128-channel tiles, one plane: the h plane)
                const unsigned char* s3 = src + wv * (PPW * 1024);
                const unsigned d3 = l0 + WB0 + (unsigned)(slot * WSTAGE + wv * (PPW * 1024));
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, %1\n\t"
                             "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                             "global_load_lds_dwordx4 %0, %1 offset:2048"
                             :
                             : "v"(l16), "s"(s3), "s"(d3)
                             : "memory", "m0");
            } else if constexpr (PPW == 6) {  // 128-channel tiles, both planes: pieces 6w .. 6w+5 (the instruction offset ends at 4095)
                const unsigned char* s3 = src + wv * (PPW * 1024);
                const unsigned d3 = l0 + WB0 + (unsigned)(slot * WSTAGE + wv * (PPW * 1024));
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, %1\n\t"
                             "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                             "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                             "global_load_lds_dwordx4 %0, %1 offset:3072"
                             :
                             : "v"(l16), "s"(s3), "s"(d3)
                             : "memory", "m0");
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, %1\n\t"
                             "global_load_lds_dwordx4 %0, %1 offset:1024"
                             :
                             : "v"(l16), "s"(s3 + 4096), "s"(d3 + 4096u)
                             : "memory", "m0");
            } else if constexpr (PPW == 1) {  // 32-channel tiles, the h plane only (3 KiB): piece w (stager 3: piece 2 again -- harmless, uniform bookkeeping)
                const int pc = wv < 3 ? wv : 2;
                const unsigned char* s3 = src + pc * 1024;
                const unsigned d3 = l0 + WB0 + (unsigned)(slot * WSTAGE + pc * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, %1"
                             :
                             : "v"(l16), "s"(s3), "s"(d3)
                             : "memory", "m0");
            } else {  // 64-channel tiles, the h plane only (6 KiB at the start of the stage): pieces w and w + 2; 32-channel tiles, both planes (6 KiB)
                const unsigned char* s3 = src + wv * 1024;
                const unsigned d3 = l0 + WB0 + (unsigned)(slot * WSTAGE + wv * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, %1\n\t"
                             "global_load_lds_dwordx4 %0, %1 offset:2048"
                             :
                             : "v"(l16), "s"(s3), "s"(d3)
                             : "memory", "m0");
            }
        };


        if constexpr (PRO == PRO_PRESPLIT) {
            // ---- pre-split input (presplit.hip): affine, SiLU and the f16 split were applied ONCE, by a pre-pass that wrote the two
            // planes as [b][chunk][plane][group][H + 2][W][8 ch]; the x tile of a chunk is NPIECE LDS-DMA pieces of 1 KiB whose lanes
            // address their own entry (tile row outside the image: the layout's zero rows; azimuth wrap: a lane address).  No pixel
            // loads, no transform, no LDS writes: ~16 DMA instructions per chunk and wave.  Queue of one iteration:
            //   D0 [PPW], x(q+1) [PPX] | #1 | D1 [PPW] | #2 | D2 [PPW] | #3        (x(q+1) has three segments to land)
            static_assert(MRK <= 2 && NRK == 2 && RING == 4, "pre-split input: 64 x 4 and 32 x 4 tiles");
            constexpr int NENT = NPLK * XPL, NPIECE = (NENT + 63) / 64, PPX = (NPIECE + 3) / 4;
            const int plane_px = (H + 2) * W;  // 16-byte entries per (plane, group)
            const unsigned chunk_bytes = (unsigned)(NPL * NG * plane_px) * 16u;
            int rowent[PPX], colx[PPX];
            bool live[PPX];
            unsigned ldsoff[PPX], voff[PPX];
#pragma unroll
            for (int i = 0; i < PPX; ++i) {
                const int pc = wave + 4 * i < NPIECE ? wave + 4 * i : NPIECE - 1;  // (past the end: the last piece again -- uniform vmcnt bookkeeping)
                const int e = pc * 64 + lane;
                live[i] = e < NENT;
                const int ee = live[i] ? e : 0;
                const int plg = ee / (XR * XS), rem = ee - plg * (XR * XS), r = rem / XS, c = rem - r * XS;
                rowent[i] = (plg * (H + 2) + r) * W;  // (plane, group, tile row) part; plg = plane * NG + group as in the layout
                colx[i] = c == XS - 1 ? 0 : c - 1;    // (the dump column: any address)
                ldsoff[i] = (unsigned)pc * 1024u;
            }
            int x_item = 0, x_c = 0;
            const unsigned char* x_base = nullptr;
            auto set_x_item = [&](int it) __attribute__((always_inline)) {
                int cot, b, th, tw;
                decode(it, cot, b, th, tw);
                x_base = reinterpret_cast<const unsigned char*>(p.x.p0) + (long)b * p.x.bs0 * 4;
#pragma unroll
                for (int i = 0; i < PPX; ++i) {
                    int gc = tw * TW + colx[i];
                    if (gc < 0) gc += W;
                    if (gc >= W) gc -= W;  // azimuth is periodic
                    voff[i] = (unsigned)(rowent[i] + th * TH * W + gc) * 16u;
                }
            };
            auto dma_x = [&](int buf) __attribute__((always_inline)) {  // the cursor's chunk -> x buffer `buf`; advances the cursor
                const unsigned long long sv = (unsigned long long)(x_base + (size_t)x_c * chunk_bytes);
                const unsigned char* src = (const unsigned char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sv >> 32)) << 32) |
                                                                  (unsigned)__builtin_amdgcn_readfirstlane((int)sv));
                const unsigned l0 = lds0 + (unsigned)buf * XBYTES;
#pragma unroll
                for (int i = 0; i < PPX; ++i)
                    if (live[i])
                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff[i]), "s"(src), "s"(l0 + ldsoff[i]) : "memory", "m0");
                if (x_c + 1 < nchunks)
                    ++x_c;
                else if (x_item + 1 < nIt) {
                    x_c = 0;
                    set_x_item(++x_item);
                }  // (past the end: the last chunk again, into the buffer nobody reads)
            };
            set_x_item(0);
            set_dma_item(0);
            dma_stage(0, ic<0>{});
            dma_stage(0, ic<1>{});
            dma_stage(0, ic<2>{});
            dma_x(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // P
            asm volatile("" ::: "memory");
            constexpr int N12 = 2 * PPW + PPX, N3 = 2 * PPW;  // younger operations when the stage due at #1 / #2, at #3 (with x(q+1)) must have landed
            for (int q = 0; q < Q; ++q) {
                dma_stage(3 * q, ic<3>{});  // D0: stage 3q+3
                dma_x((q + 1) & 1);         // x(q+1): its buffer was last read in chunk q-1
                stamp(10);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N12) : "memory");  // stage 3q+1 landed
                stamp(11);
                __builtin_amdgcn_s_barrier();                               // #1
                stamp(12);
                asm volatile("" ::: "memory");
                dma_stage(3 * q, ic<4>{});  // D1
                stamp(13);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N12) : "memory");  // stage 3q+2 landed
                stamp(14);
                __builtin_amdgcn_s_barrier();                               // #2
                stamp(15);
                asm volatile("" ::: "memory");
                dma_stage(3 * q, ic<5>{});  // D2
                stamp(16);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N3) : "memory");   // stage 3q+3 and x(q+1) landed
                stamp(17);
                __builtin_amdgcn_s_barrier();                               // #3: publishes x(q+1)
                stamp(18);
                asm volatile("" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // nothing may outlive the block
            return;
        }
        // Staging units: one aligned quad (8 channels x 4 pixels) of one tile row.  Units 0 .. 32 XR - 1 are the interior quads, the next
        // 4 XR the quads holding a halo column (the three pixels they do not need go to the dump column), anything beyond repeats the
        // last one -- see conv_bf16x3_stream_kernel.  Four-row tiles have 216 units, one per staging thread; the eight-row tile has 360:
        // unit t4 for every thread, and a second one -- rows 8 and 9 (waves 4 / 6: one each per lane) or a halo column (waves 5 / 7) --
        // that waves 4, 5 stage in the chunks of odd index and waves 6, 7 in those of even index: three units per stager and pair of chunks.
        constexpr int U_INT = 32 * XR, U_HALO = 4 * XR;
        int s_row[NU], s_g[NU], s_col[NU];
        unsigned dsto[NU][4];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int v = u == 0 ? t4 : 256 + (wave & 1) * 64 + lane;
            if (v < U_INT) {
                // 16 consecutive lanes = 4 quads x 2 channel groups x 2 rows: their 16-byte LDS entries of one pixel fall into 16
                // different bank quartets (quad stride 64 B, group stride XR * 1072 = 32 | 224 mod 256, row stride 1072 = 48 mod 256), so the
                // ds_write_b128 of the transform is conflict-free; with 16 lanes on 16 consecutive quads it was a 4-way conflict
                // that took LDS cycles from the multipliers' fragment reads (round-2 timeline: +20 % on two of three segments)
                const int qd = ((v >> 2) & 3) | (((v >> 4) & 3) << 2);
                s_g[u] = v & 1;
                s_row[u] = ((v >> 6) << 1) | ((v >> 1) & 1);
                s_col[u] = qd * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) dsto[u][e] = (unsigned)(((s_g[u] * XR + s_row[u]) * XS + 1 + qd * 4 + e) * 16);
            } else {
                const int w = v - U_INT < U_HALO ? v - U_INT : U_HALO - 1;
                s_row[u] = w >> 2;
                s_g[u] = (w >> 1) & 1;
                const bool right = w & 1;
                s_col[u] = right ? TW : -4;
                const unsigned rowb = (unsigned)((s_g[u] * XR + s_row[u]) * XS);
#pragma unroll
                for (int e = 0; e < 4; ++e) dsto[u][e] = (rowb + (right ? (e == 0 ? XS - 2 : XS - 1) : (e == 3 ? 0 : XS - 1))) * 16;
            }
        }

        // ---- load cursor: the chunk whose pixels are fetched next (two to three chunks ahead of the multipliers) ----
        int l_item = 0, l_c = 0;
        const unsigned char* l_x0 = nullptr;  // (byte pointers: the input is fp32 or, IOM bit 0, fp16)
        const unsigned char* l_x1 = nullptr;
        unsigned l_voff[NU];  // this lane's byte offset inside a channel group's planes (per unit)
        bool l_ok[NU];
        bool l_edge = false;  // (wave-uniform) the tile touches the top or bottom image row: some lanes' rows are zero padding
        auto set_load_item = [&](int it) __attribute__((always_inline)) {
            int cot, b, th, tw;
            decode(it, cot, b, th, tw);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                int gc = tw * TW + s_col[u];
                if (gc < 0) gc += W;
                if (gc >= W) gc -= W;  // azimuth is periodic
                const int gr = th * TH + s_row[u] - 1;
                l_ok[u] = gr >= 0 && gr < H;  // rows outside [0,H) are zero padding (of the ACTIVATED tensor)
                l_voff[u] = (unsigned)(s_g[u] * 8 * HW + (l_ok[u] ? gr * W + gc : 0)) * (unsigned)XE;  // (16 HW floats < 2^31: launcher)
            }
            l_edge = th == 0 || th == nTh - 1;
            l_x0 = reinterpret_cast<const unsigned char*>(p.x.p0) + (long)b * p.x.bs0 * XE;
            l_x1 = p.x.p1 ? reinterpret_cast<const unsigned char*>(p.x.p1) + (long)b * p.x.bs1 * XE : reinterpret_cast<const unsigned char*>(p.x.p0);
        };
        // Two register sets of raw pixels: the set filled in iteration q is transformed in iteration q+2.
        // four pixels of one channel: 16 bytes of fp32 or 8 of fp16 (one 64-bit scalar: as a vector of two dwords behind the inline-assembly
        // load, hipcc read dword 0 for all four pixels)
        using RAW = std::conditional_t<X16, unsigned long long, f32x4>;
        struct RawSet {
            RAW raw[NU][8];    // per unit: 8 channels x 4 pixels
            bool ok[NU];       // row inside the image
            bool edge;         // wave-uniform: the tile has padding rows at all (interior tiles skip the masking)
        };
        RawSet set0, set1;
        // The pixel loads are inline assembly, like the weight DMA: hipcc's own vmcnt bookkeeping cannot see the DMA, so a
        // compiler-placed wait for these registers would drain the whole queue.  Every wait is explicit instead (use_set).
        // (wave-uniform 64-bit base in SGPRs + one 32-bit lane offset: no per-load 64-bit vector address arithmetic)
        auto sbase = [&](const void* q) __attribute__((always_inline)) {
            const unsigned long long v = (unsigned long long)q;
            return (const unsigned char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)v));
        };
        auto gload = [&](f32x4& d, const unsigned char* base, unsigned voff) __attribute__((always_inline)) {
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(base) : "memory");
        };
        auto gload_px = [&](RAW& d, const unsigned char* base, unsigned voff) __attribute__((always_inline)) {
#ifdef F2_NO_PXLOAD  // fault bisection (wrong results): no pixel load is issued
            asm volatile("" : "=v"(d) : "v"(voff), "s"(base));
            return;
#endif
#ifdef F2_NT_XLOAD  // experiment (round 5): the input pixels as streaming loads
            if constexpr (X16) asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(d) : "v"(voff), "s"(base) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(d) : "v"(voff), "s"(base) : "memory");
            return;
#endif
            if constexpr (X16) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(base) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(base) : "memory");
        };
        // the cursor's chunk, units 0 .. NUR - 1: 8 loads each; advances the cursor
        auto load_next = [&](RawSet& r, auto NUR) __attribute__((always_inline)) {
            const int ci0 = l_c * CK;
            const unsigned char* xq = sbase(ci0 >= c0 ? l_x1 + (long)(ci0 - c0) * HW * XE : l_x0 + (long)ci0 * HW * XE);
#pragma unroll
            for (int u = 0; u < decltype(NUR)::value; ++u) {
#pragma unroll
                for (int i = 0; i < 8; ++i) gload_px(r.raw[u][i], xq + (size_t)i * HW * XE, l_voff[u]);
                r.ok[u] = l_ok[u];
            }
            r.edge = l_edge;
            if (l_c + 1 < nchunks)
                ++l_c;
            else if (l_item + 1 < nIt) {
                l_c = 0;
                set_load_item(++l_item);
            }  // (past the end: the last chunk again, unused)
        };
        // wait until at most `newer` younger VMEM operations are in flight, then hand the set's registers to the compiler
        auto use_set = [&](RawSet& r, auto NUR, auto NEWER) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(NEWER)::value) : "memory");
#pragma unroll
            for (int u = 0; u < decltype(NUR)::value; ++u)
                asm volatile("" : "+v"(r.raw[u][0]), "+v"(r.raw[u][1]), "+v"(r.raw[u][2]), "+v"(r.raw[u][3]), "+v"(r.raw[u][4]), "+v"(r.raw[u][5]), "+v"(r.raw[u][6]), "+v"(r.raw[u][7]));
        };

        // ---- the folded GroupNorm affine (a, d) of the tile's sample: Cin pairs in an LDS table, refreshed per tile ----
        // Four more 16-byte global loads per thread and chunk were a quarter of the stagers' memory instructions (their third
        // segment is bound by the CU's memory-instruction issue rate).  The table of tile T+1 is fetched into registers while
        // T's second-to-last chunk is transformed and written to LDS behind the transform of T's last chunk -- the block's
        // barrier #3 publishes it before the first chunk of T+1 is transformed.
        f32x4 ad4[NU][4];  // (a, d) of this thread's 8 channels of the chunk being transformed (per unit)
        f32x4 tab_v;       // table prefetch: pairs 2 t4, 2 t4 + 1 of the next tile's sample
        int t_item = 0;    // tile whose table is in LDS
        int x_c = 0;       // chunk (within its tile) that is transformed next
        const f32x4* adtab = reinterpret_cast<const f32x4*>(smem + ADTAB0);
        const bool tab_lane = PRO != PRO_NONE && t4 * 2 < p.Cin;
        // GroupNorm folded into this launch (ConvParams::gn_partial): the table of the block's ONE sample is computed in the prologue, from
        // the producers' statistics slots, and never refreshed
        const bool fold = PRO != PRO_NONE && p.gn_partial != nullptr;
        auto table_fetch = [&](int it) __attribute__((always_inline)) {  // one VMEM operation (asm: counted by hand like the others)
            if (PRO == PRO_NONE) return;
            int cot, b, th, tw;
            decode(it < nIt ? it : nIt - 1, cot, b, th, tw);
            const unsigned char* src = sbase(reinterpret_cast<const float*>(p.aff) + (size_t)b * p.Cin * 2);
            gload(tab_v, src, tab_lane ? (unsigned)t4 * 16u : 0u);
        };
        auto table_store = [&](auto NEWER) __attribute__((always_inline)) {
            if (PRO == PRO_NONE) return;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(NEWER)::value) : "memory");
            asm volatile("" : "+v"(tab_v));
            if (tab_lane) *reinterpret_cast<f32x4*>(smem + ADTAB0 + t4 * 16) = tab_v;
        };
        // ---- the tile's biases (EPI2): fetched while its chunk 1 is staged (three or more chunks before the multipliers want them), one float
        // per thread, into the slot of the tile's parity -- read last at the end of the tile two back
        float bias_v = 0.f;
        auto bias_fetch = [&]() __attribute__((always_inline)) {  // one VMEM operation, counted by hand like the others
            int cot, b, th, tw;
            decode(t_item < nIt ? t_item : nIt - 1, cot, b, th, tw);
            const unsigned char* src = sbase(p.bias + cot * COT);
            asm volatile("global_load_dword %0, %1, %2" : "=v"(bias_v) : "v"(t4 < COT ? (unsigned)t4 * 4u : 0u), "s"(src) : "memory");
        };
        auto bias_store = [&](auto NEWER) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(NEWER)::value) : "memory");
            asm volatile("" : "+v"(bias_v));
            if (t4 < COT) *reinterpret_cast<float*>(smem + BIAS0 + ((t_item & 1) * COT + t4) * 4) = bias_v;
        };
        auto table_read = [&](int u, bool ok, bool edge) __attribute__((always_inline)) {  // unit u's 8 channels of chunk x_c
            if (PRO == PRO_NONE) return;
            const int j0 = (x_c * CK + s_g[u] * 8) >> 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // zero padding of the ACTIVATED tensor: a = d = 0 gives silu(0) = 0
                const f32x4 v = adtab[j0 + j];
                ad4[u][j] = v;
                if (edge) ad4[u][j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };

        // ---- transform: affine, SiLU (same arithmetic as conv_bf16x3_pair_kernel), f16 split, pack ----
        unsigned xpk[2][NPLK][4];
        auto xf = [&](RawSet& r, int u, float& qv0, float& qv1, float& qm0, float& qm1, int k, int sl) __attribute__((always_inline)) {
            const int e = k >> 2, i2 = k & 3, eo = e & 1;
            constexpr bool silu = PRO == PRO_AFFINE_SILU;
            if (sl == 0) {
                if constexpr (X16) {  // (pixel e of the four: half e % 2 of dword e / 2)
                    qv0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(r.raw[u][2 * i2] >> (16 * e)));
                    qv1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(r.raw[u][2 * i2 + 1] >> (16 * e)));
                } else {
                    qv0 = r.raw[u][2 * i2][e];
                    qv1 = r.raw[u][2 * i2 + 1][e];
                }
                if (PRO != PRO_NONE) {
                    qv0 = qv0 * ad4[u][i2][0] + ad4[u][i2][1];
                    qv1 = qv1 * ad4[u][i2][2] + ad4[u][i2][3];
                }
#ifdef R2DM_ACCURATE_SILU  // accuracy ablation (scripts/error_budget.py): libm exp + IEEE division instead of v_exp / v_rcp
            } else if (sl == 1) {
                if (silu) { qv0 = qv0 / (1.0f + expf(-qv0)); qv1 = qv1 / (1.0f + expf(-qv1)); }
            } else if (sl >= 2 && sl <= 5) {
#else
            } else if (sl == 1) {
                if (silu) { qm0 = qv0 * -1.4426950408889634f; qm1 = qv1 * -1.4426950408889634f; }
            } else if (sl == 2) {
                if (silu) { qm0 = __builtin_amdgcn_exp2f(qm0); qm1 = __builtin_amdgcn_exp2f(qm1); }
            } else if (sl == 3) {
                if (silu) { qm0 = 1.0f + qm0; qm1 = 1.0f + qm1; }
            } else if (sl == 4) {
                if (silu) { qm0 = __builtin_amdgcn_rcpf(qm0); qm1 = __builtin_amdgcn_rcpf(qm1); }
            } else if (sl == 5) {
                if (silu) { qv0 *= qm0; qv1 *= qm1; }
#endif
            } else if (sl == 6) {
                if (PRO == PRO_NONE && r.edge) {
                    qv0 = r.ok[u] ? qv0 : 0.f;
                    qv1 = r.ok[u] ? qv1 : 0.f;
                }
            } else if (sl == 7) {
#ifdef F2_NO_XF  // timing ablation (wrong results)
                xpk[eo][0][i2] = xpk[eo][NPLK - 1][i2] = __float_as_uint(qv0);
#else
                if constexpr (NPLK == 2) {
                    if constexpr (LSCALED) split_f16x2(qv0, qv1, xpk[eo][0][i2], xpk[eo][1][i2]);
                    else split_f16x2_true(qv0, qv1, xpk[eo][0][i2], xpk[eo][1][i2]);
                } else {
                    using f32x2 = __attribute__((ext_vector_type(2))) float;
                    xpk[eo][0][i2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{qv0, qv1}, f16x2));  // v_cvt_pk_f16_f32 (RNE)
                }
#endif
            }
        };
        // pixels 2*half, 2*half+1 of the quads of units 0 .. NUR - 1 into x buffer `buf`
        auto transform_half = [&](RawSet& r, auto NUR, int half, unsigned char* buf, bool from_table = true) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < decltype(NUR)::value; ++u) {
                float v0[2][4], v1[2][4], m0[2][4], m1[2][4];
                if (half == 0 && from_table) table_read(u, r.ok[u], r.edge);
                // both pixels stage by stage: eight independent dependency chains (16 values) per stage -- the stager shares its
                // SIMD with a multiplier and cannot afford to wait for its own results (exp2 / rcp are quarter rate)
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {
#ifdef F2_NO_XF
                    if (sl != 0 && sl != 7) continue;
#endif
#pragma unroll
                    for (int eo = 0; eo < 2; ++eo)
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) xf(r, u, v0[eo][i2], v1[eo][i2], m0[eo][i2], m1[eo][i2], 4 * (2 * half + eo) + i2, sl);
                }
#pragma unroll
                for (int eo = 0; eo < 2; ++eo)
#pragma unroll
                    for (int pl = 0; pl < NPLK; ++pl)
                        *reinterpret_cast<u32x4*>(buf + dsto[u][2 * half + eo] + pl * (XPL * 16)) =
                            u32x4{xpk[eo][pl][0], xpk[eo][pl][1], xpk[eo][pl][2], xpk[eo][pl][3]};
            }
        };

        // N1 = staging units of this wave in the chunks of odd index (register set 1), N0 = in those of even index (set 0): 1 and 1 but
        // for the eight-row tile (2 and 1 in waves 4, 5; 1 and 2 in waves 6, 7).  The hand-counted waits below depend on them.
        auto run = [&](auto N1C, auto N0C) __attribute__((always_inline)) {
            constexpr int NL1 = 8 * decltype(N1C)::value, NL0 = 8 * decltype(N0C)::value;
            // ---- prologue: ring stages 0..RING-2 requested, chunk 0 in x buffer 0, the pixels of chunks 1 and 2 requested ----
            // All 256 blocks start at once and the first pixels are a bandwidth burst (2 chunks x 32 KiB per CU = 16 MiB: 3-4 us of
            // HBM; in-kernel timeline, profiles/r03_launch_overhead.txt), with the multipliers waiting at P.  So: what the first
            // transform needs goes FIRST (the pixels of chunk 0, then this thread's (a, d) of chunk 0 -- read straight from global
            // memory: the other waves' table writes are only published by P), everything else queues behind it (ring stages and table:
            // L2 hits; the pixels of chunk 1), and the wait before the transform counts those as younger operations.
            stamp(30);
            set_load_item(0);
            set_dma_item(0);
            int fold_bits = 0;  // this thread's share of the range bound gn_finalize would have recorded
            if (PRO != PRO_NONE && fold) {
                // ---- the GroupNorm of this launch's input, folded in (round 5).  gn_finalize_kernel's reduction, per group g: thread t adds slots
                // t, t + 256, ... in ascending order, wave_sum_f64, the four waves as (w0 + w1) + (w2 + w3) -- the same slots, order and
                // arithmetic (gn_math.h), hence the same bits.  All 8 groups at once: <= 4 slots per thread and group (host-checked), two per round in
                // flight; the pixels of chunk 0 are requested behind the first round and land while it is reduced.
                using d2 = __attribute__((ext_vector_type(2))) double;
                int cot0, b0, th0, tw0;
                decode(0, cot0, b0, th0, tw0);
                const int cpg = p.gn_cpg, Cn = p.Cin;
                // the norm's per-channel parameters (w, sh) of the channels this thread will need, requested FIRST (they do not depend on the
                // statistics: fetched behind them they were a second serial round trip): two table channels, eight per staging unit of chunk 0
                float wq[2] = {1.f, 1.f}, hq[2] = {0.f, 0.f};       // table channels 2 t4, 2 t4 + 1
                f32x4 wu[decltype(N0C)::value][2], hu[decltype(N0C)::value][2];
                {
                    const bool ada = p.gn_ada != nullptr;
                    const unsigned char* wb = sbase(ada ? p.gn_ada + b0 * p.gn_ada_stride : p.gn_gamma);
                    const unsigned char* hb = sbase(ada ? p.gn_ada + b0 * p.gn_ada_stride + Cn : p.gn_beta);
                    const bool have_w = ada || p.gn_gamma != nullptr, have_h = ada || p.gn_beta != nullptr;
                    using f32x2 = __attribute__((ext_vector_type(2))) float;
                    f32x2 w2 = {1.f, 1.f}, h2 = {0.f, 0.f};
                    if (have_w) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(w2) : "v"(tab_lane ? (unsigned)t4 * 8u : 0u), "s"(wb) : "memory");
                    if (have_h) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(h2) : "v"(tab_lane ? (unsigned)t4 * 8u : 0u), "s"(hb) : "memory");
#pragma unroll
                    for (int u = 0; u < decltype(N0C)::value; ++u)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            wu[u][i] = f32x4{1.f, 1.f, 1.f, 1.f};
                            hu[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (have_w) gload(wu[u][i], wb, (unsigned)(s_g[u] * 8 + 4 * i) * 4u);
                            if (have_h) gload(hu[u][i], hb, (unsigned)(s_g[u] * 8 + 4 * i) * 4u);
                        }
                    // (the waits for the statistics below cover these: they are older)
                    asm volatile("" : "+v"(w2), "+v"(h2));
                    wq[0] = w2[0]; wq[1] = w2[1]; hq[0] = h2[0]; hq[1] = h2[1];
                }
                const bool is_ada = p.gn_ada != nullptr;  // AdaGN: w = 1 + scale (ops.py:190-200); added below, after the loads have landed
                auto wfix = [&](float w) __attribute__((always_inline)) { return is_ada ? 1.0f + w : w; };
                const unsigned char* pb = sbase(p.gn_partial + (size_t)b0 * 8 * p.gn_stride * 2);
                const int npg = (p.gn_slots + 255) >> 8;
                double gs[8], gq[8], ge[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) gs[g] = gq[g] = ge[g] = 0.0;
                for (int r0 = 0; r0 < npg; r0 += 2) {
                    d2 pv[8][2];
                    bool live[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int sl = t4 + 256 * (r0 + i);
                        live[i] = sl < p.gn_slots;
#pragma unroll
                        for (int g = 0; g < 8; ++g)
                            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(pv[g][i]) : "v"((unsigned)(g * p.gn_stride + (live[i] ? sl : 0)) * 16u), "s"(pb) : "memory");
                    }
                    if (r0 == 0) {
                        load_next(set0, N0C);
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL0) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            asm volatile("" : "+v"(pv[g][i]));
                            const d2 v = live[i] ? pv[g][i] : d2{0.0, 0.0};
                            gs[g] += v[0];
                            gq[g] += v[1];
                            ge[g] = v[1] > ge[g] ? v[1] : ge[g];
                        }
                }
#pragma unroll
                for (int u = 0; u < decltype(N0C)::value; ++u)
                    asm volatile("" : "+v"(wu[u][0]), "+v"(wu[u][1]), "+v"(hu[u][0]), "+v"(hu[u][1]));
                // all eight groups' wave totals with two reduce-scatters (lane L: group (L >> 3) & 7), the maxima one by one
                double* scr = reinterpret_cast<double*>(smem + RES0);  // [group][wave][sum, squares, max |x| bound] (the waiting slots are idle until the first tile's end)
                {
                    const double ta = wave_sum8_scatter(gs, lane), tq = wave_sum8_scatter(gq, lane);
                    if ((lane & 7) == 0) {
                        const int g = (lane >> 3) & 7;
                        scr[(g * 4 + wave) * 3 + 0] = ta;
                        scr[(g * 4 + wave) * 3 + 1] = tq;
                    }
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const float gm = wave_max_f32((float)sqrt(ge[g]) * 1.000001f);
                        if (lane == 0) scr[(g * 4 + wave) * 3 + 2] = (double)gm;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();  // P0 (the multipliers pass it too): the waves' totals are published
                asm volatile("" ::: "memory");
                const double n_el = (double)cpg * (double)HW;
                // one group per lane (0 .. 7 of every wave: two fp64 divisions and a square root, once), handed to the lanes that need it through
                // a private LDS row (LDS operations of a wave are in order: no barrier)
                float* mrow = reinterpret_cast<float*>(scr + 104) + wave * 32;  // [group][mean, rstd, gmax, well conditioned]
                if (lane < 8) {
                    const double* r = scr + lane * 12;
                    const GnMoments mo = gn_moments((r[0] + r[3]) + (r[6] + r[9]), (r[1] + r[4]) + (r[7] + r[10]), n_el, p.gn_eps);
                    const float gmax = fmaxf(fmaxf((float)r[2], (float)r[5]), fmaxf((float)r[8], (float)r[11]));
                    *reinterpret_cast<f32x4*>(mrow + lane * 4) = f32x4{mo.mean, mo.rstd, gmax, mo.well_conditioned ? 1.f : 0.f};
                }
                const int csh = __builtin_ctz((unsigned)cpg);  // (cpg is 8, 16, 32 or 64: host-checked)
                struct GroupOf { GnMoments mo; float gmax; };
                auto group_of = [&](int c) __attribute__((always_inline)) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(mrow + (c >> csh) * 4);
                    GroupOf o;
                    o.mo.mean = v[0]; o.mo.rstd = v[1]; o.mo.well_conditioned = v[3] != 0.f;
                    o.gmax = v[2];
                    return o;
                };
                if (tab_lane) {  // this thread's two table entries (published by P)
                    const GroupOf o = group_of(2 * t4);
                    wq[0] = wfix(wq[0]);
                    wq[1] = wfix(wq[1]);
                    const float2 e0 = gn_affine(o.mo, wq[0], hq[0]), e1 = gn_affine(o.mo, wq[1], hq[1]);
                    *reinterpret_cast<f32x4*>(smem + ADTAB0 + t4 * 16) = f32x4{e0.x, e0.y, e1.x, e1.y};
                    const int i0 = __float_as_int(gn_bound(o.mo, e0, wq[0], hq[0], o.gmax, n_el)), i1 = __float_as_int(gn_bound(o.mo, e1, wq[1], hq[1], o.gmax, n_el));
                    fold_bits = i0 > i1 ? i0 : i1;
                }
#pragma unroll
                for (int u = 0; u < decltype(N0C)::value; ++u) {  // (a, d) of this thread's channels of chunk 0, straight into registers
                    const GroupOf o = group_of(s_g[u] * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float2 e0 = gn_affine(o.mo, wfix(wu[u][j >> 1][2 * (j & 1)]), hu[u][j >> 1][2 * (j & 1)]);
                        const float2 e1 = gn_affine(o.mo, wfix(wu[u][j >> 1][2 * (j & 1) + 1]), hu[u][j >> 1][2 * (j & 1) + 1]);
                        ad4[u][j] = f32x4{e0.x, e0.y, e1.x, e1.y};
                    }
                }
                if (p.gn_range) {  // the block's bound: one value per wave now, one conditional atomic per block behind P
                    fold_bits = wave_max_i32(fold_bits);
                    if (lane == 0) reinterpret_cast<int*>(scr + 96)[wave] = fold_bits;  // (scr[96 .. 98): behind the [8][4][3] totals, in front of the moment rows at scr + 104)
                }
            } else
                load_next(set0, N0C);
            if (PRO != PRO_NONE && !fold) {
                int cot0, b0, th0, tw0;
                decode(0, cot0, b0, th0, tw0);
                const unsigned char* a0 = sbase(reinterpret_cast<const float*>(p.aff) + (size_t)b0 * p.Cin * 2);
#pragma unroll
                for (int u = 0; u < decltype(N0C)::value; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) gload(ad4[u][j], a0 + 16 * j, (unsigned)s_g[u] * 64u);
            }
            if (!fold) table_fetch(0);
            stamp(31);
            dma_stage(0, ic<0>{});
            dma_stage(0, ic<1>{});
            if constexpr (RING == 4) dma_stage(0, ic<2>{});
            load_next(set1, N1C);  // (nchunks >= 4: the cursor stays inside the first tile)
            constexpr int RING_NEWER = (RING - 1) * PPW;
            use_set(set0, N0C, ic<RING_NEWER + NL1>{});  // (ring stages and chunk 1 are younger)
            stamp(32);
            if (PRO != PRO_NONE) {
#pragma unroll
                for (int u = 0; u < decltype(N0C)::value; ++u) {
                    asm volatile("" : "+v"(ad4[u][0]), "+v"(ad4[u][1]), "+v"(ad4[u][2]), "+v"(ad4[u][3]));
#pragma unroll
                    for (int j = 0; j < 4; ++j) ad4[u][j] = set0.ok[u] ? ad4[u][j] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            transform_half(set0, N0C, 0, smem, false);
            transform_half(set0, N0C, 1, smem, false);
            x_c = 1;
            stamp(33);
            if (!fold) table_store(ic<RING_NEWER + NL1>{});  // (all stagers write their part of the first tile's table: published by P)
            load_next(set0, N0C);                 // chunk 2 (the cursor saturates: harmless for a one-tile, two-chunk block)
            // P needs the ring stages and this wave's x writes; the pixels of chunk 1 may still be on their way -- iteration 0 waits for
            // them itself (the multipliers get to their first taps that much earlier)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NL1 + NL0) : "memory");
            stamp(34);
            __builtin_amdgcn_s_barrier();  // P
            stamp(35);
            asm volatile("" ::: "memory");
            if (PRO != PRO_NONE && fold && p.gn_range && wave == 0 && lane == 0) {  // (the four waves' bounds were published by P)
                const int* wb = reinterpret_cast<const int*>(reinterpret_cast<const double*>(smem + RES0) + 96);
                int bits = wb[0] > wb[1] ? wb[0] : wb[1];
                bits = wb[2] > bits ? wb[2] : bits;
                bits = wb[3] > bits ? wb[3] : bits;
                if (bits > __atomic_load_n(p.gn_range + 1, __ATOMIC_RELAXED)) atomicMax(p.gn_range + 1, bits);
            }

            // ---- the staging wave's share of a tile end (EPI3): quarters NQ - NSQ .. NQ - 1 of its multiplying partner (same `wave`), which that wave has
            // turned into LDS in front of barrier E1.  Everything here is ordinary compiler-scheduled code: its own waits assume that only its own
            // loads are in flight -- the hand-issued pixel loads and DMA pieces still in the queue are OLDER and retire first (waits get stricter,
            // never wrong); the queue is drained before E2, so the hand-counted waits of the next iteration see what they always saw.
            int se_item = 0;  // tiles this wave has helped to finish
            auto tile_end_share = [&]() __attribute__((always_inline)) {
                using gcf = const float __attribute__((address_space(1)))*;
                using gcf4 = const f32x4 __attribute__((address_space(1)))*;
                using gf4 = f32x4 __attribute__((address_space(1)))*;
                constexpr int SEGW = TW / 32, Q0 = NQ - NSQ;
                int cot, b, th, tw;
                decode(se_item, cot, b, th, tw);
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const float sc_blk = p.scale ? *(gcf)p.scale : 1.0f, wsc = p.wscale ? *(gcf)p.wscale : 1.0f;
                const gf4 yu = (gf4)(p.y + (long)b * p.y_bs + (long)(cot * COT) * HW);
                const gcf4 ru = (gcf4)(p.res + (long)b * p.res_bs + (long)(cot * COT) * HW);  // (only dereferenced if p.res)
                auto qoff = [&](int qd) __attribute__((always_inline)) {  // (f32x4 units) this lane's four pixels of channel ln >> 3 of quarter qd's first block
                    const int n = qd % NR, sg = wave * NR + n;
                    return ((ln >> 3) * HW + (th * TH + sg / SEGW) * W + tw * TW + (sg % SEGW) * 32 + (ln & 7) * 4) >> 2;
                };
                // (all NSQ quarters' residual is requested up front: VMEM operations retire in order, stores included -- a load requested behind a quarter's
                // stores is only usable once those stores are acknowledged; first version, two buffers refilled in turn: 3 k cycles per quarter)
                f32x4 rv[NSQ > 0 ? NSQ : 1][4] = {};
                auto req = [&](int qd, f32x4 (&r)[4]) __attribute__((always_inline)) {
                    if (!p.res) return;
                    const int m = qd / NR, off = qoff(qd);
#pragma unroll
                    for (int k8 = 0; k8 < 4; ++k8) r[k8] = (ru + (long)(m * 32 + k8 * 8) * (HW >> 2))[off];
                };
#pragma unroll
                for (int k = 0; k < NSQ; ++k) req(Q0 + k, rv[k]);
                const float* btab = reinterpret_cast<const float*>(smem + BIAS0) + (se_item & 1) * COT + (ln >> 3);
                float amax = 0.f;
                stamp(60);
                __builtin_amdgcn_s_barrier();  // E1: the partner's quarters are in LDS
                asm volatile("" ::: "memory");
                stamp(61);
                float ps[2][4], pq[2][4];
#pragma unroll
                for (int k = 0; k < NSQ; ++k) {
                    const int qd = Q0 + k, m = qd / NR, off = qoff(qd);
                    const float* slot = reinterpret_cast<const float*>(smem + (k < 2 ? RES0 + (wave * 2 + k) * 4096 : XBYTES + 4096 + (wave * XDUMP + (k - 2)) * 4096));
#pragma unroll
                    for (int k8 = 0; k8 < 4; ++k8) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(slot + k8 * 256 + (ln >> 3) * 32 + (ln & 7) * 4);
                        f32x4 v = t * wsc + btab[m * 32 + k8 * 8];
                        v = rv[k][k8] + v;  // (without a residual the buffers stay zero)
                        v *= sc_blk;
                        (yu + (long)(m * 32 + k8 * 8) * (HW >> 2))[off] = v;
#ifdef F2_RANGE_ELEMENTWISE
                        if (p.range) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
#endif
                        ps[k & 1][k8] = (v[0] + v[1]) + (v[2] + v[3]);
                        pq[k & 1][k8] = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
#ifndef F2_RANGE_ELEMENTWISE
                        amax = fmaxf(amax, pq[k & 1][k8]);
#endif
                    }
                    if ((k & 1) && p.stat) {  // a pair of quarters = one image row of one 32-channel half = one statistics slot: the multipliers' half_stats
#ifndef F2_STATS_F64
                        float st_s[4], st_q[4];
#pragma unroll
                        for (int k8 = 0; k8 < 4; ++k8) {
                            st_s[k8] = ps[0][k8] + ps[1][k8];
                            st_q[k8] = pq[0][k8] + pq[1][k8];
                        }
#else
                        double st_s[4], st_q[4];
#pragma unroll
                        for (int k8 = 0; k8 < 4; ++k8) {
                            st_s[k8] = (double)ps[0][k8] + (double)ps[1][k8];
                            st_q[k8] = (double)pq[0][k8] + (double)pq[1][k8];
                        }
#endif
                        const int mm = qd >> 1;
                        if constexpr (NR == 4) epi_stat_write_bfly8(p, st_s, st_q, b, 2 * th + (wave >> 1), tw, nTw, cot * COT + (mm >> 1) * 32, 2 * (wave & 1) + (mm & 1), ln);
                        else epi_stat_write_bfly8(p, st_s, st_q, b, th, tw, nTw, cot * COT + mm * 32, wave, ln);
                    }
                }
                if (p.range) {
#ifndef F2_RANGE_ELEMENTWISE
                    const float a = sqrtf(wave_max_f32(amax)) * 1.000001f;
#else
                    const float a = wave_max_f32(amax);
#endif
                    const int bits = __float_as_int(a);
                    if (ln == 0 && bits > __atomic_load_n(p.range + 1, __ATOMIC_RELAXED)) atomicMax(p.range + 1, bits);
                }
                ++se_item;
                stamp(62);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the stores are out: the next iteration's counted waits start from an empty queue
                stamp(63);
                __builtin_amdgcn_s_barrier();  // E2: x buffer 1 (patches, waiting quarters) is the stagers' again
                asm volatile("" ::: "memory");
            };

            // ---- chunk q of the multipliers <-> this iteration stages chunk q+1 (three segments around the block's barriers) ----
            // Weight stage sigma is first read behind barrier B'_{sigma-1}; its ring slot is free again behind B'_{sigma} and takes
            // stage sigma+RING, due RING-1 barriers later.  One stage is requested per segment, right behind the barrier that frees
            // its slot:  D0(q) = stage 3q+RING-1, D1(q) = 3q+RING, D2(q) = 3q+RING+1.  A stage requested at the start of segment j
            // must have landed at the end of segment j+RING-2 (NEWER1..3 younger operations may still fly).
            // The pixels of chunk q+3 are requested in the third segment, into the register set emptied in the first two; they are
            // transformed a whole iteration later (the other set holds chunk q+2 meanwhile).  VMEM queue of one iteration:
            //   D0 [PPW] | #1 | D1 [PPW] | #2 | D2 [PPW], raw(q+3) [NLc] | #3          (NLc: this iteration's set, NLp: the other one's)
            auto stage_iter = [&](int q, RawSet& cur, auto NCUR, auto NPREV) __attribute__((always_inline)) {
                constexpr int NLc = 8 * decltype(NCUR)::value, NLp = 8 * decltype(NPREV)::value;
                constexpr int NEWER1 = RING == 4 ? 2 * PPW + NLp : NLp + PPW, NEWER2 = RING == 4 ? 2 * PPW + NLp : PPW,
                              NEWER3 = RING == 4 ? 2 * PPW + NLc : PPW + NLc;
                unsigned char* nbuf = smem + ((q + 1) & 1) * XBYTES;  // x buffer of chunk q+1 (read by nobody during chunk q)
                const bool tile_ends = x_c == 0;  // this iteration stages the NEXT tile's chunk 0: the multipliers are in their tile's last chunk
                const bool fetch = x_c == nchunks - 2, last_of_tile = x_c == nchunks - 1;
                const bool bias_now = EPI2 && x_c == 1;  // (nchunks >= 4: never the iteration of the table fetch or store)
                if (fetch && !fold) table_fetch(t_item + 1);  // (one more operation in the queue: the counted waits below only get stricter)
                if (bias_now) bias_fetch();          // (likewise)
                dma_stage(3 * q, ic<RING - 1>{});    // D0
                if (q == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLp + PPW) : "memory");  // (only chunk 2 and D0 are younger than chunk 1)
                use_set(cur, NCUR, ic<4 * PPW + NLp>{});  // raw(q+1): requested two iterations ago (the previous iteration's queue and D0 are younger)
                transform_half(cur, NCUR, 0, nbuf);       // (past the end: the last chunk again, into the buffer nobody reads)
                stamp(10);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEWER1) : "memory");  // stage 3q+1 landed
                stamp(11);
                __builtin_amdgcn_s_barrier();        // #1 = B'_{3q}
                stamp(12);
                asm volatile("" ::: "memory");
                dma_stage(3 * q, ic<RING>{});        // D1
                transform_half(cur, NCUR, 1, nbuf);
                stamp(13);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEWER2) : "memory");  // stage 3q+2 landed
                stamp(14);
                __builtin_amdgcn_s_barrier();        // #2 = B'_{3q+1}
                stamp(15);
                asm volatile("" ::: "memory");
                if (last_of_tile) {                  // the next transform belongs to the next tile: its table (fetched at the start
                    if (!fold) table_store(ic<5 * PPW + NLp>{});  // of the previous iteration) goes to LDS
                    ++t_item;
                }
                if (bias_now) bias_store(ic<2 * PPW>{});  // (fetched at this iteration's start: D0 and D1 are younger; published by #3)
                x_c = last_of_tile ? 0 : x_c + 1;
                dma_stage(3 * q, ic<RING + 1>{});    // D2
                load_next(cur, NCUR);                // pixels of chunk q+3
                stamp(16);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NEWER3) : "memory");  // stage 3q+3 landed; this wave's x writes done
                stamp(17);
                __builtin_amdgcn_s_barrier();        // #3 = B'_{3q+2}: publishes x(q+1)
                stamp(18);
                asm volatile("" ::: "memory");
                if constexpr (EPI3) {
                    if (tile_ends) tile_end_share();
                }
            };
            for (int q = 0; q < Q; q += 2) {  // (Q is even: chunks per tile are)
                stage_iter(q, set1, N1C, N0C);
                stage_iter(q + 1, set0, N0C, N1C);
            }
        };
        if constexpr (NU == 1) {
            run(ic<1>{}, ic<1>{});
        } else {
            if (wave < 2) run(ic<2>{}, ic<1>{});
            else run(ic<1>{}, ic<2>{});
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // nothing may outlive the block
        return;
    }

    // ============================================= multiplier waves =============================================
    // fragment addresses in x buffer 0 / 1.  The eight-row tile keeps ONE address per buffer (its four segments are constant offsets from
    // the first: they go into the instruction's offset field -- eight address registers were what pushed that tile into spilling)
    constexpr int NXB = NR == 4 ? 1 : NR;
    unsigned xb0[NXB], xb1[NXB];
#pragma unroll
    for (int n = 0; n < NXB; ++n) {
        const int s = wave * NR + n;
        xb0[n] = lds0 + (unsigned)(((hi * XR + (s >> 1)) * XS + (s & 1) * 32 + l31) * 16);
        xb1[n] = xb0[n] + XBYTES;
    }
    const unsigned lds_w0 = lds0 + WB0 + (unsigned)((hi * COT + l31) * 16);

    f32x16 acc[MR][NR], acl[ACC2 ? MR : 1][ACC2 ? NR : 1];  // xh wh | xh wl + xl wh (scaled by 2^11); wide tile: all three products in acc
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[m][n][r] = 0.f;
                if constexpr (ACC2) acl[m][n][r] = 0.f;
            }

    // Fragments of one tap: the l plane is used by the first two products only and is single-buffered -- the next tap's is
    // read as soon as its last product has been issued; the h plane is needed up to the last product and is double-buffered.
    u32x4 fa0[2][MR], fb0[2][NR], fa1[MR], fb1[NR];
    constexpr int NOPS = MR + NR;  // operands of a tap per plane: A (weights) 0 .. MR - 1, B (pixels) MR .. MR + NR - 1
    // read of plane PL, operand WW (A m0, A m1, B n0, B n1) of tap (ky, tx); plane 0 goes to buffer NB
    auto frag_rd = [&](const unsigned (&xb)[NXB], unsigned wb, auto KY, auto TX, auto PL, auto WW, auto NB) __attribute__((always_inline)) {
        constexpr int ky = decltype(KY)::value, tx = decltype(TX)::value, pl = decltype(PL)::value, w = decltype(WW)::value;
        constexpr int nb = decltype(NB)::value;
        if constexpr (w < MR) {
            u32x4& d = pl == 0 ? fa0[nb][w] : fa1[w];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(wb), "i"(pl * (3 * NG * COT * 16) + tx * (NG * COT * 16) + w * 512));
        } else {
            u32x4& d = pl == 0 ? fb0[nb][w - MR] : fb1[w - MR];
            constexpr int n = w - MR, seg = NXB == 1 ? ((n >> 1) * XS + (n & 1) * 32) * 16 : 0;  // (segment n of the wave: row n / 2, columns 32 (n % 2) ...)
            static_assert(XPL * 16 + 2 * (XS * 16) + 2 * 16 + 3 * XS * 16 < 65536, "ds_read offset field");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(xb[NXB == 1 ? 0 : n]), "i"(pl * (XPL * 16) + ky * (XS * 16) + tx * 16 + seg));
        }
    };
    // all fragments of a tap (0, 0) at once, in the order the taps read them: plane 0 (into h buffer NB), then plane 1
    auto frag_all = [&](const unsigned (&xb)[NXB], unsigned wb, auto NB) __attribute__((always_inline)) {
        auto f = [&](auto PL, auto WW) __attribute__((always_inline)) { frag_rd(xb, wb, ic<0>{}, ic<0>{}, PL, WW, NB); };
        f(ic<0>{}, ic<0>{}); f(ic<0>{}, ic<1>{}); f(ic<0>{}, ic<2>{});
        if constexpr (NOPS >= 4) f(ic<0>{}, ic<3>{});
        if constexpr (NOPS == 6) { f(ic<0>{}, ic<4>{}); f(ic<0>{}, ic<5>{}); }
        if constexpr (NPLK == 2) {
            f(ic<1>{}, ic<0>{}); f(ic<1>{}, ic<1>{}); f(ic<1>{}, ic<2>{});
            if constexpr (NOPS >= 4) f(ic<1>{}, ic<3>{});
            if constexpr (NOPS == 6) { f(ic<1>{}, ic<4>{}); f(ic<1>{}, ic<5>{}); }
        }
    };
    auto frag_first = [&](unsigned wb) __attribute__((always_inline)) {
        auto f = [&](auto PL, auto WW) __attribute__((always_inline)) { frag_rd(xb0, wb, ic<0>{}, ic<0>{}, PL, WW, ic<0>{}); };
        f(ic<0>{}, ic<0>{}); f(ic<0>{}, ic<1>{}); f(ic<0>{}, ic<2>{});
        if constexpr (NOPS >= 4) f(ic<0>{}, ic<3>{});
        if constexpr (NOPS == 6) { f(ic<0>{}, ic<4>{}); f(ic<0>{}, ic<5>{}); }
        if constexpr (NPLK == 2) {
            f(ic<1>{}, ic<0>{}); f(ic<1>{}, ic<1>{}); f(ic<1>{}, ic<2>{});
            if constexpr (NOPS >= 4) f(ic<1>{}, ic<3>{});
            if constexpr (NOPS == 6) { f(ic<1>{}, ic<4>{}); f(ic<1>{}, ic<5>{}); }
        }
    };

    int e_c = 0;  // chunk within the current tile
    // one tap = 3 MR NR units (12 / 24): one MFMA + at most one fragment read of the next tap.  T = tap within the chunk, PAR = parity of
    // the chunk (x buffer PAR; tap 8 reads the next chunk's tap 0 from the other x buffer, which the stagers published at
    // this tap's barrier -- also across a tile boundary).  Products (64-channel tile; the 128-channel tile sends all three to acc):
    //   xh wl -> acl | xl wh -> acl | xh wh -> acc        reads behind unit: 0 .. MR+NR-1 plane h, MR NR + (0 .. MR-1) wl' (A), 2 MR NR + (0, 1) xl' (B)
    // LDS returns in order, so the waits count younger reads: at the tap's start everything but xl' (2 reads) is needed,
    // before the second product those too (the MR + NR reads of this tap are younger).  A barrier tap waits for everything first:
    // its barrier retires a ring slot and, at ky = 2, the x buffer of the previous chunk.
    auto tap = [&](int q, auto TT, auto PAR) __attribute__((always_inline)) {
        constexpr int t = decltype(TT)::value, par = decltype(PAR)::value;
        constexpr int ky = t / 3, tx = t % 3, cur = (par * 9 + t) & 1;
        constexpr int kyn = t < 8 ? (t + 1) / 3 : 0, txn = t < 8 ? (t + 1) % 3 : 0;  // next tap
        const int sigma = 3 * q + ky;
        const int c = e_c;  // chunk within the tile (a running counter: q % nchunks would be a division per chunk)
        if (tx == 2) {
            // B'_sigma: every multiplier has its fragments of tap (sigma, 2) in registers (ring slot sigma % 8 retires), the
            // stagers' pieces of stage sigma+1 have landed and -- at ky == 2 -- the next chunk's x buffer is complete
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp(1 + ky);
            __builtin_amdgcn_s_barrier();
            stamp(4 + ky);
            asm volatile("" ::: "memory");
        } else if (NPLK == 2) {
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR) : "memory");  // (everything but xl': NR reads)
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one plane: all four fragments of this tap)
        }
        // ring slot of the next tap's stage (RING == 3: stage 3 q' + ky' sits in slot ky')
        const unsigned wbn = RING == 4 ? lds_w0 + (unsigned)(((t < 8 ? sigma + (kyn != ky ? 1 : 0) : sigma + 1) & 3) * WSTAGE) : lds_w0 + (unsigned)(kyn * WSTAGE);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int qq = NPLK == 2 ? i / (MR * NR) : 2, m = (i / NR) % MR, n = i % NR;  // (one plane: the xh wh product only)
            if (i == MR * NR) {
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NOPS < MR * NR ? NOPS : MR * NR) : "memory");  // (the reads this tap has issued so far are younger)
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x16& ac = !ACC2 || qq == 2 ? acc[m][n] : acl[ACC2 ? m : 0][ACC2 ? n : 0];
            const u32x4& ua = qq == 0 ? fa1[m] : fa0[cur][m];  // A = weights (rows = output channels), B = pixels
            const u32x4& ub = qq == 1 ? fb1[n] : fb0[cur][n];
            // (unit order: xh(B) wl(A) | xl(B) wh(A) | xh wh -- the A/B roles of "x" and "w" are spelled out in the reads above)
            const f16x8 fra = __builtin_bit_cast(f16x8, ua), frb = __builtin_bit_cast(f16x8, ub);
#ifdef F2_NO_MFMA  // timing ablation (wrong results): how fast are the stagers without a multiplier on their SIMD?
            if (i == 0) asm volatile("" :: "v"(fra), "v"(frb), "v"(ac));
            else
#endif
            if (t == 0 && c == 0 && (ACC2 ? (qq == 0 || qq == 2) : i < MR * NR)) {  // first product into each accumulator of a tile: start from zero
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(fra, frb, zero, 0, 0, 0);
            } else {
                ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(fra, frb, ac, 0, 0, 0);
            }
            // the next tap's fragments (at the end of a tile these are dropped -- the epilogue wants the registers -- and read
            // again after it)
            auto fr = [&](auto PL, auto WW) __attribute__((always_inline)) {
                if (t < 8)
                    frag_rd(par ? xb1 : xb0, wbn, ic<kyn>{}, ic<txn>{}, PL, WW, ic<cur ^ 1>{});
                else
                    frag_rd(par ? xb0 : xb1, wbn, ic<0>{}, ic<0>{}, PL, WW, ic<cur ^ 1>{});
            };
            if (i == 0) fr(ic<0>{}, ic<0>{});
            if (i == 1) fr(ic<0>{}, ic<1>{});
            if (i == 2) fr(ic<0>{}, ic<2>{});
            if constexpr (NPLK == 1 && UNITS == 2) {  // (32-channel tile, one plane: two MFMAs per tap carry three reads)
                if (i == 1) fr(ic<0>{}, ic<2>{});
            }
            if constexpr (NOPS >= 4) {
                if (i == 3) fr(ic<0>{}, ic<3>{});
            }
            if constexpr (NPLK == 2 && MR == 1) {  // (units: 0, 1 xh wl | 2, 3 xl wh | 4, 5 xh wh)
                if (i == 2) fr(ic<1>{}, ic<0>{});  // wl' (its last product was unit 1)
                if (i == 4) fr(ic<1>{}, ic<1>{});  // xl' (its last product was unit 3)
                if (i == 5) fr(ic<1>{}, ic<2>{});
            }
            if constexpr (NOPS == 6) {
                if (i == 4) fr(ic<0>{}, ic<4>{});
                if (i == 5) fr(ic<0>{}, ic<5>{});
            }
            if constexpr (NPLK == 2 && MR == 2 && NR == 4) {
                if (i == 8) fr(ic<1>{}, ic<0>{});  // wl' (its last product was unit 7)
                if (i == 9) fr(ic<1>{}, ic<1>{});
                if (i == 16) fr(ic<1>{}, ic<2>{});  // xl' (its last product was unit 15)
                if (i == 17) fr(ic<1>{}, ic<3>{});
                if (i == 18) fr(ic<1>{}, ic<4>{});
                if (i == 19) fr(ic<1>{}, ic<5>{});
            }
            if constexpr (NPLK == 2 && MR == 2 && NR == 2) {
                if (i == 4) fr(ic<1>{}, ic<0>{});  // wl' (its last product was unit 3)
                if (i == 5) fr(ic<1>{}, ic<1>{});
                if (i == 8) fr(ic<1>{}, ic<2>{});  // xl' (its last product was unit 7)
                if (i == 9) fr(ic<1>{}, ic<3>{});
            }
            if constexpr (NPLK == 2 && MR == 4) {
                if (i == 8) fr(ic<1>{}, ic<0>{});  // wl' (its last product was unit 7)
                if (i == 9) fr(ic<1>{}, ic<1>{});
                if (i == 10) fr(ic<1>{}, ic<2>{});
                if (i == 11) fr(ic<1>{}, ic<3>{});
                if (i == 16) fr(ic<1>{}, ic<4>{});  // xl' (its last product was unit 15)
                if (i == 17) fr(ic<1>{}, ic<5>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int e_item = 0, e_cot, e_b, e_th, e_tw;
    decode(0, e_cot, e_b, e_th, e_tw);

    // ---- epilogue: one quarter of the tile (32 channels x 32 pixels per wave) at the tile's end, the other three at the next
    // tile's first three chunk boundaries.  The multipliers are not what bounds a chunk (the stagers are, by ~0.7 k cycles), so
    // a quarter (~1.2 k cycles) behind a chunk costs the chunk ~0.5 k -- against ~10 k cycles of epilogue + refill per tile with
    // the matrix pipe idle when all four quarters ran at the tile's end.  The finished tile waits in LDS, already "turned"
    // (conv_epilogue.h: 4 consecutive pixels of one channel per lane) in the 48 KiB that used to hold the prefetched residual;
    // the residual now comes straight from global memory, each quarter requested one chunk before it is added (16 registers).
    using gcf = const float __attribute__((address_space(1)))*;
    using gcf4 = const f32x4 __attribute__((address_space(1)))*;
    using gf4 = f32x4 __attribute__((address_space(1)))*;
    constexpr int SEGW = TW / 32;
    static_assert(ONEACC || ((NQ == 4 || NQ == 2) && RESQ == NQ - 1), "all quarters but the first wait in LDS");
    float bias_e[MR * 4];             // this lane's biases of the tile whose epilogue is in progress
    using RV = std::conditional_t<Y16, unsigned long long, f32x4>;  // four pixels of one channel of the residual, as stored (fp16: one 64-bit scalar)
    using gcrv = const RV __attribute__((address_space(1)))*;
    using grv = RV __attribute__((address_space(1)))*;
    auto rv_f32 = [&](const RV& r) __attribute__((always_inline)) {
        if constexpr (Y16) {
            return f32x4{(float)__builtin_bit_cast(_Float16, (unsigned short)r), (float)__builtin_bit_cast(_Float16, (unsigned short)(r >> 16)),
                         (float)__builtin_bit_cast(_Float16, (unsigned short)(r >> 32)), (float)__builtin_bit_cast(_Float16, (unsigned short)(r >> 48))};
        } else {
            return r;
        }
    };
    RV rv_e[4] = {};                  // residual of the quarter that is processed next
    float cs[4] = {}, cq[4] = {};     // fp32 statistics of a half's first quarter, waiting for its second
    float amax_e = 0.f;
    int pe_b = 0, pe_th = 0, pe_tw = 0, pe_cot = 0;
    bool pending = false;
    const float sc_blk = p.scale ? *(gcf)p.scale : 1.0f;
    const float wsc = p.wscale ? *(gcf)p.wscale : 1.0f;  // inverse of the packer's power-of-two weight scale (exact)
    float* const dump = reinterpret_cast<float*>(smem + RES0) + wave * (RESQ * 1024);
    float* const patch = EPI3 ? reinterpret_cast<float*>(smem + XBYTES) + wave * 256 : !ONEACC ? reinterpret_cast<float*>(smem + PATCH0) + wave * 256 : dump;
    auto fresh_lane = [&]() __attribute__((always_inline)) {  // (per-lane constants must not be hoisted across the MFMA stream)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        return ln;
    };
    auto res_request_to = [&](RV (&rv)[4], auto QD, int b, int th, int tw, int cot, int ln) __attribute__((always_inline)) {
        if (!p.res) return;
        constexpr int qd = decltype(QD)::value, m = qd / NR, n = qd % NR;
        const int s = wave * NR + n;
        const int off = ((ln >> 3) * HW + (th * TH + s / SEGW) * W + tw * TW + (s % SEGW) * 32 + (ln & 7) * 4) >> 2;
        const gcrv ru = (gcrv)(reinterpret_cast<const unsigned char*>(p.res) + ((long)b * p.res_bs + (long)(cot * COT) * HW) * YE);
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
#ifdef F2_NT_RES  // experiment (round 5): the residual is read once
            rv[k8] = __builtin_nontemporal_load(ru + (long)(m * 32 + k8 * 8) * (HW >> 2) + off);
#else
            rv[k8] = (ru + (long)(m * 32 + k8 * 8) * (HW >> 2))[off];
#endif
        }
    };
    auto res_request = [&](auto QD, int b, int th, int tw, int cot, int ln) __attribute__((always_inline)) { res_request_to(rv_e, QD, b, th, tw, cot, ln); };
    // quarter QD's residual -> the wave's waiting slot QD (0 | 1) by LDS-DMA: four pieces of 1 KiB (one 8-channel block each)
    auto res_dma = [&](auto QD, int b, int th, int tw, int cot, int ln) __attribute__((always_inline)) {
        constexpr int qd = decltype(QD)::value, m = qd / NR, n = qd % NR;
        static_assert(qd < RESQ, "waiting slots");
        const int s = wave * NR + n;
        const unsigned voff = (unsigned)((ln >> 3) * HW + (th * TH + s / SEGW) * W + tw * TW + (s % SEGW) * 32 + (ln & 7) * 4) * 4u;
        const unsigned dst = lds0 + RES0 + (unsigned)((wave * RESQ + qd) * 4096);
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            const unsigned long long sv = (unsigned long long)(p.res + b * p.res_bs + (long)(cot * COT + m * 32 + k8 * 8) * HW);
            const unsigned char* src = (const unsigned char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sv >> 32)) << 32) |
                                                              (unsigned)__builtin_amdgcn_readfirstlane((int)sv));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(src), "s"(dst + (unsigned)k8 * 1024u) : "memory", "m0");
        }
    };
    // the turn: this wave's accumulator quarter (m, n) -> dst[8-channel block][8 channels][32 pixels]
    auto turn_write = [&](auto QD, float* dst, int ln) __attribute__((always_inline)) {
        constexpr int qd = decltype(QD)::value, m = qd / NR, n = qd % NR;
        const int l31e = ln & 31, hie = ln >> 5;
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8)
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[k8 * 256 + (j + 4 * hie) * 32 + l31e] = acc[m][n][4 * k8 + j];
    };
    // bias, residual, scale, store, statistics of one turned quarter (t[k8]: 4 consecutive pixels of channel 8 k8 + lane / 8)
    auto quarter = [&](auto QD, const f32x4 (&t)[4], const RV (&rv)[4], int b, int th, int tw, int cot, int ln, float (&ps)[4], float (&pq)[4]) __attribute__((always_inline)) {
        constexpr int qd = decltype(QD)::value, m = qd / NR, n = qd % NR;
        const int s = wave * NR + n;
        const int off = ((ln >> 3) * HW + (th * TH + s / SEGW) * W + tw * TW + (s % SEGW) * 32 + (ln & 7) * 4) >> 2;
        const grv yu = (grv)(reinterpret_cast<unsigned char*>(p.y) + ((long)b * p.y_bs + (long)(cot * COT) * HW) * YE);
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            f32x4 v = t[k8] * wsc + bias_e[m * 4 + k8];
            v = rv_f32(rv[k8]) + v;  // (without a residual the buffers stay zero -- res_request_to returns before it writes them: no select per element)
            v *= sc_blk;             // (1.0f without p.scale: exact)
#ifdef F2_NO_STORE  // fault bisection (wrong results): nothing is stored
            if (ln >= 0) { ps[k8] = v[0]; pq[k8] = v[1]; continue; }
#endif
            if constexpr (Y16) {     // stored as fp16 (RNE); the statistics and the range maximum are those of the stored values
                using f32x2 = __attribute__((ext_vector_type(2))) float;
                const RV pk = (RV)__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, f16x2)) |
                              ((RV)__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, f16x2)) << 32);
                (yu + (long)(m * 32 + k8 * 8) * (HW >> 2))[off] = pk;
                v = rv_f32(pk);
            } else {
#ifdef F2_NT_STORE  // experiment (round 5): the output as a streaming store -- it is not read again before the launch ends
                __builtin_nontemporal_store(v, yu + (long)(m * 32 + k8 * 8) * (HW >> 2) + off);
#else
                (yu + (long)(m * 32 + k8 * 8) * (HW >> 2))[off] = v;
#endif
            }
#ifdef F2_RANGE_ELEMENTWISE
            if (p.range) amax_e = fmaxf(fmaxf(amax_e, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
#endif
            ps[k8] = (v[0] + v[1]) + (v[2] + v[3]);
            pq[k8] = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
#ifndef F2_RANGE_ELEMENTWISE
            // Round 6: the range maximum rides on the statistics -- max over the sums of squares of four pixels, one instruction, no branch; its root
            // (range_flush) bounds max|y| from above by at most a factor 2.  Element by element it was seven instructions behind a wave-uniform
            // branch per store (the ISA: two taken branches per 16-byte store in the launches that do not track a range at all): -0.6 % on the step
            // (profiles/r06_tile_end_diet.txt).  An infinite or overflowing output gives an infinite sum: still a trip.
            amax_e = fmaxf(amax_e, pq[k8]);
#endif
        }
    };
    auto half_stats = [&](auto M, const float (&s0)[4], const float (&q0)[4], const float (&s1)[4], const float (&q1)[4], int b, int th,
                          int tw, int cot, int ln) __attribute__((always_inline)) {
        if (!p.stat) return;
#ifdef F2_NO_STORE
        if (ln >= 0) return;
#endif
#ifndef F2_STATS_F64
        float st_s[4], st_q[4];
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            st_s[k8] = s0[k8] + s1[k8];
            st_q[k8] = q0[k8] + q1[k8];
        }
#else
        double st_s[4], st_q[4];
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {  // (four pixels in fp32, fp64 beyond: conv_epilogue.h)
            st_s[k8] = (double)s0[k8] + (double)s1[k8];
            st_q[k8] = (double)q0[k8] + (double)q1[k8];
        }
#endif
        // M = index of a pair of quarters (2 M, 2 M + 1): one image row of one 32-channel half -- a statistics slot.  Four-row tiles: the
        // half is M, the slot (tile, wave).  The eight-row tile's wave owns rows 2 wave, 2 wave + 1: half M / 2, and the slot is the one the
        // four-row tiling gives that row -- tile row 2 th + wave / 2, "wave" 2 (wave % 2) + M % 2: same slots, same sums, same order.
        constexpr int mm = decltype(M)::value;
        if constexpr (NR == 4) epi_stat_write_bfly8(p, st_s, st_q, b, 2 * th + (wave >> 1), tw, nTw, cot * COT + (mm >> 1) * 32, 2 * (wave & 1) + (mm & 1), ln);
        else epi_stat_write_bfly8(p, st_s, st_q, b, th, tw, nTw, cot * COT + mm * 32, wave, ln);
    };
    auto range_flush = [&](int ln) __attribute__((always_inline)) {
        if (!p.range) return;
#ifndef F2_RANGE_ELEMENTWISE
        const float a = sqrtf(wave_max_f32(amax_e)) * 1.000001f;
#else
        const float a = wave_max_f32(amax_e);
#endif
        const int bits = __float_as_int(a);  // positive floats order like their bit patterns
        if (ln == 0 && bits > __atomic_load_n(p.range + 1, __ATOMIC_RELAXED)) atomicMax(p.range + 1, bits);
        amax_e = 0.f;
    };
    // deferred quarter S (1..3) of the pending tile; its residual is in rv (between chunks: rv_e, and the next quarter's is
    // requested here; at the block's end all three are requested up front)
    auto slice_rv = [&](auto SS, const RV (&rv)[4], auto NEXT) __attribute__((always_inline)) {
        constexpr int S = decltype(SS)::value;
        const int ln = fresh_lane();
        f32x4 t[4];
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) t[k8] = *reinterpret_cast<const f32x4*>(dump + (S - 1) * 1024 + k8 * 256 + (ln >> 3) * 32 + (ln & 7) * 4);
        float ps[4], pq[4];
        quarter(SS, t, rv, pe_b, pe_th, pe_tw, pe_cot, ln, ps, pq);
        if constexpr (S < NQ - 1 && decltype(NEXT)::value) res_request(ic<S + 1>{}, pe_b, pe_th, pe_tw, pe_cot, ln);
        if (S == 1) half_stats(ic<0>{}, cs, cq, ps, pq, pe_b, pe_th, pe_tw, pe_cot, ln);
        if (S == 2) {
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) { cs[k8] = ps[k8]; cq[k8] = pq[k8]; }
        }
        if (S == 3) half_stats(ic<1>{}, cs, cq, ps, pq, pe_b, pe_th, pe_tw, pe_cot, ln);
        if (S == NQ - 1) {  // the tile's last quarter
            range_flush(ln);
            pending = false;
        }
    };
    auto slice = [&](auto SS) __attribute__((always_inline)) { slice_rv(SS, rv_e, ic<1>{}); };

    auto chunk = [&](int q, auto PAR) __attribute__((always_inline)) {
        if constexpr (decltype(PAR)::value == 1) {
            if (e_c == nchunks - 1) {  // the tile's last chunk: its biases and the first quarter's residual are requested now
                const int ln = fresh_lane();
                if constexpr (!ONEACC) {  // (the one-accumulator tiles have no registers to hold their biases through a chunk: requested at the tile's end)
#pragma unroll
                    for (int m = 0; m < MR; ++m)
#pragma unroll
                        for (int k8 = 0; k8 < 4; ++k8) bias_e[m * 4 + k8] = ((gcf)p.bias)[e_cot * COT + m * 32 + k8 * 8 + (ln >> 3)];
                }
                // (the eight-row tile cannot hold 16 residual registers through a chunk -- as loads here they were spilled one by one behind
                // vmcnt(0): its first quarter's residual is requested at the tile's end, in front of the biases that quarter waits for anyway)
                if constexpr (EPI2) {
                    // the residual of quarters 0 and 1 by LDS-DMA into the wave's two waiting slots (4 KiB each: block k8 at 1 KiB k8, lane L's
                    // four pixels at 16 L -- the turned layout, read back with one ds_read_b128 per block); no register is written
                    if constexpr (!Y16) {
                        if (p.res) {
                            res_dma(ic<0>{}, e_b, e_th, e_tw, e_cot, ln);
                            res_dma(ic<1>{}, e_b, e_th, e_tw, e_cot, ln);
                        }
                    }
                } else if constexpr (NR != 4) res_request(ic<0>{}, e_b, e_th, e_tw, e_cot, ln);
            }
        }
        tap(q, ic<0>{}, PAR); tap(q, ic<1>{}, PAR); tap(q, ic<2>{}, PAR);
        tap(q, ic<3>{}, PAR); tap(q, ic<4>{}, PAR); tap(q, ic<5>{}, PAR);
        tap(q, ic<6>{}, PAR); tap(q, ic<7>{}, PAR); tap(q, ic<8>{}, PAR);
        ++e_c;
        if constexpr (EPI2) {
            // ---- one-accumulator tiles, round 5: eight quarters (m, n) per wave, numbered qd = m NR + n, in pairs (2 M, 2 M + 1) that share a
            // statistics slot, ALL finished at the tile's end.  Biases: LDS table (stagers).  Residual: quarters 0 and 1 from the wave's two
            // waiting slots (LDS-DMA, requested a chunk ago), quarters 2 .. 7 through four register buffers -- 2, 3 requested here (they have
            // quarters 0 and 1, ~2 k cycles, to arrive), each buffer refilled for the quarter four further on as soon as it is free.
            if constexpr (decltype(PAR)::value == 1) {
                if (e_c == nchunks) {
                    e_c = 0;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // no fragment read may land in a register the epilogue reuses
                    stamp(7);
                    const int ln = fresh_lane();
                    const int l31e = ln & 31, hie = ln >> 5;
                    RV rvA[4] = {}, rv1[4] = {}, rv2[4] = {}, rv3[4] = {};
                    // the DMA of this tile's last chunk has had the chunk to land; hipcc does not see it: an explicit wait, in front of every
                    // memory operation of the tile's end (nothing else of this wave is in flight: the wait is for the DMA alone)
                    if constexpr (!Y16) {
                        if (p.res) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else {  // (an fp16 residual has no 8-byte LDS-DMA form: the first two quarters' residual is requested here, like the others)
                        res_request_to(rvA, ic<0>{}, e_b, e_th, e_tw, e_cot, ln);
                        res_request_to(rv1, ic<1>{}, e_b, e_th, e_tw, e_cot, ln);
                    }
                    res_request_to(rv2, ic<2>{}, e_b, e_th, e_tw, e_cot, ln);
                    // (eight-row tile: the fourth buffer is requested behind quarter 0, whose accumulator registers it takes -- requested here it
                    // was spilled behind vmcnt(0))
                    if constexpr (NR != 4) res_request_to(rv3, ic<3>{}, e_b, e_th, e_tw, e_cot, ln);
                    const float* btab = reinterpret_cast<const float*>(smem + BIAS0) + (e_item & 1) * COT + (ln >> 3);
                    auto bias_load = [&](auto M) __attribute__((always_inline)) {
                        constexpr int m = decltype(M)::value;
#pragma unroll
                        for (int k8 = 0; k8 < 4; ++k8) bias_e[m * 4 + k8] = btab[m * 32 + k8 * 8];
                    };
                    bias_load(ic<0>{});
                    if constexpr (!Y16) {
                        if (p.res) {
#pragma unroll
                            for (int k8 = 0; k8 < 4; ++k8) {
                                rvA[k8] = *reinterpret_cast<const f32x4*>(dump + k8 * 256 + ln * 4);
                                rv1[k8] = *reinterpret_cast<const f32x4*>(dump + 1024 + k8 * 256 + ln * 4);
                            }
                        }
                    }
                    auto do_q = [&](auto QD, RV (&rv)[4], float (&ps)[4], float (&pq)[4]) __attribute__((always_inline)) {
                        constexpr int qd = decltype(QD)::value, m = qd / NR, n = qd % NR;
                        f32x4 t[4];
#pragma unroll
                        for (int k8 = 0; k8 < 4; ++k8) {  // (the patch is the wave's first waiting slot -- EPI3: its KiB of x buffer 1 --: its residual is in registers by now, LDS operations of a wave are in order)
#pragma unroll
                            for (int j = 0; j < 4; ++j) patch[(j + 4 * hie) * 32 + l31e] = acc[m][n][4 * k8 + j];
                            t[k8] = *reinterpret_cast<const f32x4*>(patch + (ln >> 3) * 32 + (ln & 7) * 4);
                        }
                        quarter(QD, t, rv, e_b, e_th, e_tw, e_cot, ln, ps, pq);
                    };
                    float ps0[4], pq0[4], ps1[4], pq1[4];
                    if constexpr (EPI3) {
                        // the staging partner's quarters, turned, into the two waiting slots (read out above) and x buffer 1; E1 hands them over
                        auto dump_to = [&](auto K) __attribute__((always_inline)) {
                            constexpr int k = decltype(K)::value;
                            float* dst = k < 2 ? dump + k * 1024 : reinterpret_cast<float*>(smem + XBYTES + 4096 + (wave * XDUMP + (k - 2)) * 4096);
                            turn_write(ic<NQ - NSQ + k>{}, dst, ln);
                        };
                        dump_to(ic<0>{});
                        dump_to(ic<1>{});
                        if constexpr (NSQ == 4) {
                            dump_to(ic<2>{});
                            dump_to(ic<3>{});
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        stamp(20);
                        __builtin_amdgcn_s_barrier();  // E1
                        asm volatile("" ::: "memory");
                        do_q(ic<0>{}, rvA, ps0, pq0);
                        stamp(21);
                        if constexpr (NR == 4) res_request_to(rv3, ic<3>{}, e_b, e_th, e_tw, e_cot, ln);
                        if constexpr (NQ - NSQ > 4) res_request_to(rvA, ic<4>{}, e_b, e_th, e_tw, e_cot, ln);
                        do_q(ic<1>{}, rv1, ps1, pq1);
                        stamp(22);
                        if constexpr (NQ - NSQ > 4) res_request_to(rv1, ic<5>{}, e_b, e_th, e_tw, e_cot, ln);
                        half_stats(ic<0>{}, ps0, pq0, ps1, pq1, e_b, e_th, e_tw, e_cot, ln);
                        stamp(23);
                        if constexpr (MR == 4) bias_load(ic<1>{});
                        do_q(ic<2>{}, rv2, ps0, pq0);
                        stamp(24);
                        do_q(ic<3>{}, rv3, ps1, pq1);
                        half_stats(ic<1>{}, ps0, pq0, ps1, pq1, e_b, e_th, e_tw, e_cot, ln);
                        if constexpr (NQ - NSQ > 4) {
                            bias_load(ic<2>{});
                            do_q(ic<4>{}, rvA, ps0, pq0);
                            do_q(ic<5>{}, rv1, ps1, pq1);
                            half_stats(ic<2>{}, ps0, pq0, ps1, pq1, e_b, e_th, e_tw, e_cot, ln);
                        }
                        range_flush(ln);
                        stamp(8);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();  // E2: the staging partner is done with the waiting slots, this wave with its patch in x buffer 1
                        asm volatile("" ::: "memory");
                    } else {
                    stamp(20);
                    do_q(ic<0>{}, rvA, ps0, pq0);
                    stamp(21);
                    if constexpr (NR == 4) res_request_to(rv3, ic<3>{}, e_b, e_th, e_tw, e_cot, ln);
                    res_request_to(rvA, ic<4>{}, e_b, e_th, e_tw, e_cot, ln);
                    do_q(ic<1>{}, rv1, ps1, pq1);
                    stamp(22);
                    res_request_to(rv1, ic<5>{}, e_b, e_th, e_tw, e_cot, ln);
                    half_stats(ic<0>{}, ps0, pq0, ps1, pq1, e_b, e_th, e_tw, e_cot, ln);
                    stamp(23);
                    bias_load(ic<1>{});
                    do_q(ic<2>{}, rv2, ps0, pq0);
                    stamp(24);
                    res_request_to(rv2, ic<6>{}, e_b, e_th, e_tw, e_cot, ln);
                    do_q(ic<3>{}, rv3, ps1, pq1);
                    res_request_to(rv3, ic<7>{}, e_b, e_th, e_tw, e_cot, ln);
                    half_stats(ic<1>{}, ps0, pq0, ps1, pq1, e_b, e_th, e_tw, e_cot, ln);
                    if constexpr (MR == 4) bias_load(ic<2>{});
                    do_q(ic<4>{}, rvA, ps0, pq0);
                    do_q(ic<5>{}, rv1, ps1, pq1);
                    half_stats(ic<2>{}, ps0, pq0, ps1, pq1, e_b, e_th, e_tw, e_cot, ln);
                    if constexpr (MR == 4) bias_load(ic<3>{});
                    do_q(ic<6>{}, rv2, ps0, pq0);
                    do_q(ic<7>{}, rv3, ps1, pq1);
                    half_stats(ic<3>{}, ps0, pq0, ps1, pq1, e_b, e_th, e_tw, e_cot, ln);
                    range_flush(ln);
                    stamp(8);
                    }
                    // (the accumulators restart from C = 0 in the next tile's first products: end the old values' lives)
                    if constexpr (MR == 4) {
                        asm volatile("" : "=v"(acc[0][0]), "=v"(acc[0][1]), "=v"(acc[1][0]), "=v"(acc[1][1]));
                        asm volatile("" : "=v"(acc[2][0]), "=v"(acc[2][1]), "=v"(acc[3][0]), "=v"(acc[3][1]));
                    } else {
                        asm volatile("" : "=v"(acc[0][0]), "=v"(acc[0][1]), "=v"(acc[0][2]), "=v"(acc[0][3]));
                        asm volatile("" : "=v"(acc[1][0]), "=v"(acc[1][1]), "=v"(acc[1][2]), "=v"(acc[1][3]));
                    }
                    if (++e_item < nIt) decode(e_item, e_cot, e_b, e_th, e_tw);
                    frag_first(lds_w0);  // (stage 3 (q + 1): ring slot 0)
                }
            }
        } else if constexpr (ONEACC) {
            // ---- 128-channel tile and eight-row tile (round 4's tile end, -DF2_EPI_V1): ONE accumulator per MFMA tile, eight quarters (m, n) per wave, numbered qd = m NR + n, in pairs
            // (2 M, 2 M + 1) that share a statistics slot (128 channels: row block M, both 32-pixel segments of the wave's row; eight rows: row
            // block M / 2, the wave's row M % 2).  Six are finished at the tile's end, the
            // last row block's two (6, 7) wait, turned, in 32 KiB of LDS and are finished behind the next tile's first two chunks (the
            // multipliers bound a chunk of this tile -- 24 MFMAs per tap -- so a deferred quarter costs what it takes; at the tile's end all
            // four stagers and the matrix pipe wait).  Biases are requested at the tile's end, one row block ahead (the loop has no
            // registers for sixteen of them: as 16 loads in the last chunk they were spilled one by one behind vmcnt(0) -- 4.5 k cycles);
            // the residual of quarters 1 .. 3 (HBM: ~2 k cycles) is requested up front and a quarter's buffer takes the quarter four
            // further on as soon as it is free -- with one quarter of lead every quarter waited for its residual (17.7 k cycles per tile end).
            // the waiting row block (quarters 6, 7): both behind the next tile's FIRST chunk -- quarter 6's residual was requested at the
            // tile's end, quarter 7's is requested here and has quarter 6's ~1.5 k cycles to arrive; nothing but that residual and four
            // biases stays alive through a chunk (a statistics carry between two slices was spilled and re-read behind vmcnt(0))
            auto slice_w = [&](const RV (&rv6)[4], RV (&rv7)[4], auto REQ) __attribute__((always_inline)) {
                const int ln = fresh_lane();
                if constexpr (decltype(REQ)::value) res_request_to(rv7, ic<7>{}, pe_b, pe_th, pe_tw, pe_cot, ln);
                float ps0[4], pq0[4], ps1[4], pq1[4];
                {
                    f32x4 t[4];
#pragma unroll
                    for (int k8 = 0; k8 < 4; ++k8) t[k8] = *reinterpret_cast<const f32x4*>(dump + k8 * 256 + (ln >> 3) * 32 + (ln & 7) * 4);
                    quarter(ic<6>{}, t, rv6, pe_b, pe_th, pe_tw, pe_cot, ln, ps0, pq0);
                }
                {
                    f32x4 t[4];
#pragma unroll
                    for (int k8 = 0; k8 < 4; ++k8) t[k8] = *reinterpret_cast<const f32x4*>(dump + 1024 + k8 * 256 + (ln >> 3) * 32 + (ln & 7) * 4);
                    quarter(ic<7>{}, t, rv7, pe_b, pe_th, pe_tw, pe_cot, ln, ps1, pq1);
                }
                half_stats(ic<3>{}, ps0, pq0, ps1, pq1, pe_b, pe_th, pe_tw, pe_cot, ln);
                range_flush(ln);
                pending = false;
            };
            // The eight-row tile finishes all eight quarters at the tile's end: with a residual held for a waiting row block through the next
            // tile's first chunk it spilled twelve of those registers, each behind vmcnt(0); the multipliers bound a chunk of these tiles, so
            // a deferred quarter costs what it takes wherever it runs.
            constexpr bool DEFER_ROWBLOCK = NR != 4;
            if constexpr (decltype(PAR)::value == 0) {
                if (DEFER_ROWBLOCK && pending && e_c == 1) {
                    // the next chunk's first fragments (72 registers, prefetched by tap 8) are dropped for the slice and read again behind
                    // it: with them alive the slice spilled fragments around itself
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if constexpr (MR == 4) {
                        asm volatile("" : "=v"(fa0[1][0]), "=v"(fa0[1][1]), "=v"(fa0[1][2]), "=v"(fa0[1][3]), "=v"(fb0[1][0]), "=v"(fb0[1][1]));
                        asm volatile("" : "=v"(fa1[0]), "=v"(fa1[1]), "=v"(fa1[2]), "=v"(fa1[3]), "=v"(fb1[0]), "=v"(fb1[1]));
                    } else {
                        asm volatile("" : "=v"(fa0[1][0]), "=v"(fa0[1][1]), "=v"(fb0[1][0]), "=v"(fb0[1][1]), "=v"(fb0[1][2]), "=v"(fb0[1][3]));
                        asm volatile("" : "=v"(fa1[0]), "=v"(fa1[1]), "=v"(fb1[0]), "=v"(fb1[1]), "=v"(fb1[2]), "=v"(fb1[3]));
                    }
                    RV rv7[4] = {};
                    slice_w(rv_e, rv7, ic<1>{});
                    frag_all(xb1, lds_w0, ic<1>{});  // (chunk q + 1: x buffer 1, stage 3 (q + 1) in ring slot 0, h buffer 1)
                }
            } else {
                if (e_c == nchunks) {
                    e_c = 0;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // no fragment read may land in a register the epilogue reuses
                    stamp(7);
                    const int ln = fresh_lane();
                    const int l31e = ln & 31, hie = ln >> 5;
                    const bool last_tile = !DEFER_ROWBLOCK || e_item + 1 == nIt;
                    auto bias_request = [&](auto M) __attribute__((always_inline)) {
                        constexpr int m = decltype(M)::value;
#pragma unroll
                        for (int k8 = 0; k8 < 4; ++k8) bias_e[m * 4 + k8] = ((gcf)p.bias)[e_cot * COT + m * 32 + k8 * 8 + (ln >> 3)];
                    };
                    // (eight-row tile: nothing of the epilogue lives through the MFMA loop -- its first residual buffer is local to the tile's end;
                    // rv_e, which must keep its zeros for launches without a residual, costs the other tiles 16 registers all along)
                    RV rv0l[4] = {};
                    auto& rvA = [&]() -> RV (&)[4] { if constexpr (NR == 4) return rv0l; else return rv_e; }();
                    if constexpr (NR == 4) res_request_to(rvA, ic<0>{}, e_b, e_th, e_tw, e_cot, ln);
                    bias_request(ic<0>{});
                    bias_request(ic<1>{});  // (the eight-row tile has two row blocks: all of its biases)
                    RV rv1[4] = {}, rv2[4] = {}, rv3[4] = {};
                    res_request_to(rv1, ic<1>{}, e_b, e_th, e_tw, e_cot, ln);
                    res_request_to(rv2, ic<2>{}, e_b, e_th, e_tw, e_cot, ln);
                    // (eight-row tile: the fourth buffer is requested behind quarter 0, whose accumulator registers it takes -- requested here it
                    // was spilled behind vmcnt(0))
                    if constexpr (NR != 4) res_request_to(rv3, ic<3>{}, e_b, e_th, e_tw, e_cot, ln);
                    auto do_q = [&](auto QD, RV (&rv)[4], float (&ps)[4], float (&pq)[4]) __attribute__((always_inline)) {
                        constexpr int qd = decltype(QD)::value, m = qd / NR, n = qd % NR;
                        f32x4 t[4];
#pragma unroll
                        for (int k8 = 0; k8 < 4; ++k8) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) patch[(j + 4 * hie) * 32 + l31e] = acc[m][n][4 * k8 + j];
                            t[k8] = *reinterpret_cast<const f32x4*>(patch + (ln >> 3) * 32 + (ln & 7) * 4);
                        }
                        quarter(QD, t, rv, e_b, e_th, e_tw, e_cot, ln, ps, pq);
                    };
                    float ps0[4], pq0[4], ps1[4], pq1[4];
                    stamp(20);
                    do_q(ic<0>{}, rvA, ps0, pq0);
                    stamp(21);
                    if constexpr (NR == 4) res_request_to(rv3, ic<3>{}, e_b, e_th, e_tw, e_cot, ln);
                    res_request_to(rvA, ic<4>{}, e_b, e_th, e_tw, e_cot, ln);
                    do_q(ic<1>{}, rv1, ps1, pq1);
                    stamp(22);
                    res_request_to(rv1, ic<5>{}, e_b, e_th, e_tw, e_cot, ln);
                    half_stats(ic<0>{}, ps0, pq0, ps1, pq1, e_b, e_th, e_tw, e_cot, ln);
                    stamp(23);
                    if constexpr (MR == 4) bias_request(ic<2>{});
                    do_q(ic<2>{}, rv2, ps0, pq0);
                    stamp(24);
                    if (last_tile) res_request_to(rv2, ic<6>{}, e_b, e_th, e_tw, e_cot, ln);  // (nothing left to hide the waiting quarters behind)
                    do_q(ic<3>{}, rv3, ps1, pq1);
                    if (last_tile) res_request_to(rv3, ic<7>{}, e_b, e_th, e_tw, e_cot, ln);
                    half_stats(ic<1>{}, ps0, pq0, ps1, pq1, e_b, e_th, e_tw, e_cot, ln);
                    if constexpr (MR == 4) bias_request(ic<3>{});
                    do_q(ic<4>{}, rvA, ps0, pq0);
                    do_q(ic<5>{}, rv1, ps1, pq1);
                    half_stats(ic<2>{}, ps0, pq0, ps1, pq1, e_b, e_th, e_tw, e_cot, ln);
                    // the last row block waits in LDS (the patch -- the first KiB of slot 0 -- is free now: LDS operations of a wave are in order)
                    turn_write(ic<6>{}, dump, ln);
                    turn_write(ic<7>{}, dump + 1024, ln);
                    pe_b = e_b; pe_th = e_th; pe_tw = e_tw; pe_cot = e_cot;
                    pending = true;
                    stamp(8);
                    if (last_tile) {
                        slice_w(rv2, rv3, ic<0>{});  // (clears `pending`)
                        stamp(43);
                    } else {
                        res_request(ic<6>{}, e_b, e_th, e_tw, e_cot, ln);
                    }
                    // (the accumulators restart from C = 0 in the next tile's first products: end the old values' lives)
                    if constexpr (MR == 4) {
                        asm volatile("" : "=v"(acc[0][0]), "=v"(acc[0][1]), "=v"(acc[1][0]), "=v"(acc[1][1]));
                        asm volatile("" : "=v"(acc[2][0]), "=v"(acc[2][1]), "=v"(acc[3][0]), "=v"(acc[3][1]));
                    } else {
                        asm volatile("" : "=v"(acc[0][0]), "=v"(acc[0][1]), "=v"(acc[0][2]), "=v"(acc[0][3]));
                        asm volatile("" : "=v"(acc[1][0]), "=v"(acc[1][1]), "=v"(acc[1][2]), "=v"(acc[1][3]));
                    }
                    if (++e_item < nIt) decode(e_item, e_cot, e_b, e_th, e_tw);
                    frag_first(lds_w0);  // (stage 3 (q + 1): ring slot 0)
                }
            }
        } else {
        if constexpr (decltype(PAR)::value == 0) {
            if (pending && e_c == 1) slice(ic<1>{});
            if constexpr (NQ == 4) {
                if (pending && e_c == 3) slice(ic<3>{});
            }
        } else {
            if constexpr (NQ == 4) {
                if (pending && e_c == 2) slice(ic<2>{});
            }
            if (e_c == nchunks) {  // tile finished
                e_c = 0;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // no fragment read may land in a register the epilogue reuses
                stamp(7);
                const int ln = fresh_lane();
                // the two accumulators are combined (acc + 2^-11 acl)
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int n = 0; n < NR; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if constexpr (NPLK == 2) acc[m][n][r] = __builtin_fmaf(acl[m][n][r], LINV, acc[m][n][r]);
                // (the block's last tile has nothing left to hide its deferred quarters behind: their residuals are requested now, into
                // the registers the second accumulator has just left, and the quarters follow right below)
                const bool last_tile = e_item + 1 == nIt;
                RV rv2[4] = {}, rv3[4] = {};
                if constexpr (NQ == 4) {
                    if (last_tile) {
                        res_request_to(rv2, ic<2>{}, e_b, e_th, e_tw, e_cot, ln);
                        res_request_to(rv3, ic<3>{}, e_b, e_th, e_tw, e_cot, ln);
                    }
                }
                // quarter 0 right away (through the 1 KiB patch, block by block), quarters 1..3 into the LDS area
                {
                    f32x4 t[4];
                    const int l31e = ln & 31, hie = ln >> 5;
#pragma unroll
                    for (int k8 = 0; k8 < 4; ++k8) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) patch[(j + 4 * hie) * 32 + l31e] = acc[0][0][4 * k8 + j];
                        t[k8] = *reinterpret_cast<const f32x4*>(patch + (ln >> 3) * 32 + (ln & 7) * 4);
                    }
                    turn_write(ic<1>{}, dump, ln);
                    if constexpr (NQ == 4) {
                        turn_write(ic<2>{}, dump + 1024, ln);
                        turn_write(ic<3>{}, dump + 2048, ln);
                    }
                    quarter(ic<0>{}, t, rv_e, e_b, e_th, e_tw, e_cot, ln, cs, cq);
                    res_request(ic<1>{}, e_b, e_th, e_tw, e_cot, ln);
                }
                pe_b = e_b; pe_th = e_th; pe_tw = e_tw; pe_cot = e_cot;
                pending = true;
                stamp(8);
                if (last_tile) {
                    slice_rv(ic<1>{}, rv_e, ic<0>{});
                    stamp(41);
                    if constexpr (NQ == 4) {
                        slice_rv(ic<2>{}, rv2, ic<0>{});
                        stamp(42);
                        slice_rv(ic<3>{}, rv3, ic<0>{});  // (clears `pending`)
                        stamp(43);
                    }
                }
                // both accumulators restart from C = 0 in the next tile's first products; the compiler cannot see that the
                // "accumulate" branch is never taken there and would keep all 128 registers alive: an empty definition ends
                // the old values' lives (no instruction)
                if constexpr (MR == 2) {
                    asm volatile("" : "=v"(acc[0][0]), "=v"(acc[0][1]), "=v"(acc[1][0]), "=v"(acc[1][1]));
                    if constexpr (NPLK == 2) asm volatile("" : "=v"(acl[0][0]), "=v"(acl[0][1]), "=v"(acl[1][0]), "=v"(acl[1][1]));
                } else {
                    asm volatile("" : "=v"(acc[0][0]), "=v"(acc[0][1]));
                    if constexpr (ACC2) asm volatile("" : "=v"(acl[0][0]), "=v"(acl[0][1]));  // (one plane: acl is a [1][1] dummy -- indexing [0][1] there was out of bounds)
                }
                if (++e_item < nIt) decode(e_item, e_cot, e_b, e_th, e_tw);
                // first fragments of the next tile (its chunk 0 sits in x buffer 0, stage 3(q+1) in the ring: published at
                // this chunk's last barrier); also after the last tile -- harmless, keeps the registers plainly defined
                frag_first(lds_w0 + (unsigned)(((3 * (q + 1)) & (RING - 1)) * WSTAGE));
            }
        }
        }
    };

    if (PRO != PRO_NONE && p.gn_partial != nullptr) __builtin_amdgcn_s_barrier();  // P0 of the folded GroupNorm (stagers only work there)
    stamp(36);
    __builtin_amdgcn_s_barrier();  // P: ring stages 0..RING-2 and chunk 0 staged
    stamp(37);
    asm volatile("" ::: "memory");
    frag_first(lds_w0);
    for (int q = 0; q < Q; q += 2) {  // (Q is even: Cin % 32 == 0)
        chunk(q, ic<0>{});
        chunk(q + 1, ic<1>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // prefetched fragments must not outlive the block
    stamp_real(1);
}

// ---- weight packing: OIHW fp32 -> [co tile][chunk][kernel row][plane h / l][tap in row][group][co 64 | 128][8 ch] f16 ----
// cot = output channels per tile (64 | 128); lscaled: the l plane is 2^11 (w - h) (two-accumulator kernel) or w - h at its true scale
// (the 128-channel tile's single accumulator; the layer's power-of-two scale keeps it out of the fp16 subnormals for every weight
// within 2^-12 of the largest)
// range[0] is raised to 1 if a weight does not fit the fp16 range (|w| >= 65504); the packed value saturates.
// wscale (may be nullptr: unscaled): [0] = float bits of max|w| (launch_weight_absmax), [1] <- the inverse of the power-of-two
// scale s applied to every weight of the layer (f16x2_weight_scale: max|w| s in [2^9, 2^10); the kernel's epilogue multiplies
// the matrix product by it -- exact)
__global__ void pack_conv_f16x2_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int Cout, int Cin,
                                       long total, int* __restrict__ range, float* __restrict__ wscale, int cot_size, int lscaled) {
    using namespace f2;
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);  // MODE.FP16_OVFL
    float inv = 1.0f;
    const float ws = wscale ? f16x2_weight_scale(reinterpret_cast<const int*>(wscale)[0], &inv) : 1.0f;
    if (wscale && blockIdx.x == 0 && threadIdx.x == 0) wscale[1] = inv;
    const int nchunks = Cin / CK;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int ch = r % 8;
        r /= 8;
        const int col = r % cot_size;
        r /= cot_size;
        const int g = r % NG;
        r /= NG;
        const int tx = r % 3;
        r /= 3;
        const int pl = r % NPL;
        r /= NPL;
        const int ky = r % 3;
        r /= 3;
        const int c = r % nchunks;
        const int cot = r / nchunks;
        const int co = cot * cot_size + col, ci = c * CK + g * 8 + ch;
        const float v = w[((long)co * Cin + ci) * 9 + ky * 3 + tx] * ws;
        if (!(fabsf(v) < 65504.f) && range) atomicOr(range, 1);  // (scaled: only a non-finite weight gets here)
        unsigned ph, pq;
        if (lscaled) split_f16x2(v, v, ph, pq);
        else split_f16x2_true(v, v, ph, pq);
        dst[i] = (unsigned short)((pl == 0 ? ph : pq) & 0xffffu);
    }
}

// ---- launchers -----------------------------------------------------------------------------------------------------
static int f2_cu_count() {
    static const int v = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return v;
}

// 3x3, whole px_rows x 64 pixel tiles (the wide epilogue has no pixel predication), 64- (or 128-) channel output tiles, 64 <= Cin <= 512 in
// multiples of 64 (an even number of 16-channel chunks, at least four; the (a, d) table), a concat seam on a chunk boundary.
// Tiles (co_tile x px_rows): 64 x 4 and 32 x 4 (two accumulators), 128 x 4 and 64 x 8 (one).
bool conv_f16x2_supported(int Cin, int Cout, int taps, int H, int W, int co_tile, int px_rows) {
    const bool tile_ok = (co_tile == 64 && (px_rows == 4 || px_rows == 8)) || ((co_tile == 128 || co_tile == 32) && px_rows == 4);
    return taps == 9 && tile_ok && Cout % co_tile == 0 && Cin % (4 * f2::CK) == 0 && Cin * 8 <= f2::ADTAB_BYTES &&
           H % px_rows == 0 && W % f2::TW == 0 && H * (long)W * 16 < (1L << 31);
}

long conv_f16x2_packed_floats(int Cin, int Cout) { return (long)Cout * Cin * 9 * f2::NPL / 2; }

int conv_f16x2_pick_co_tile(int Cin, int Cout, int H, int W, long pixels_times_batch, int* px_rows) {
    const bool wide_ok = conv_f16x2_supported(Cin, Cout, 9, H, W, 128, 4), tall_ok = conv_f16x2_supported(Cin, Cout, 9, H, W, 64, 8);
    int rows_dummy;
    int& rows = px_rows ? *px_rows : rows_dummy;
    rows = 4;
    const bool narrow_ok = conv_f16x2_supported(Cin, Cout, 9, H, W, 32, 4);
    if (const char* e = getenv("R2DM_F2_CO_TILE")) {  // "32" | "64" | "128" | "64x8" (read per call: per-kernel tests switch it)
        if (atoi(e) == 128 && wide_ok) return 128;
        if (atoi(e) == 32 && narrow_ok) return 32;
        if (strstr(e, "x8") && tall_ok && px_rows) rows = 8;
        return 64;
    }
    // The one-accumulator tiles take three truncating accumulator updates per tap where the 64 x 4 tile takes one.  Measured per layer
    // against fp64 (tests/test_hip_kernels.py::test_conv3x3_both_operand_splits): their rms error is 0.4-0.6x a plain fp32 fmaf chain's (the
    // fp32-MFMA kernel with one accumulator) at every depth, 0.66-0.70x the library's fp32 kernel at Cin <= 128, 1.2x at Cin = 256 and 1.7x at
    // Cin = 512, where that kernel accumulates on two levels.  Used up to Cin = 256 (R2DM_F2_WIDE_MAX_CIN: experiments), where the launch
    // still has a tile per CU at the planned batch: 128 x 4 for layers with >= 128 output channels, else 64 x 8 (R2DM_F2_TALL=0: never).
    static const int max_cin = getenv("R2DM_F2_WIDE_MAX_CIN") ? atoi(getenv("R2DM_F2_WIDE_MAX_CIN")) : 256;
    static const bool tall_on = !getenv("R2DM_F2_TALL") || atoi(getenv("R2DM_F2_TALL")) != 0;
    const long px_tiles = pixels_times_batch / (4 * f2::TW);
    if (wide_ok && Cin <= max_cin && px_tiles * (Cout / 128) >= f2_cu_count()) return 128;
    // (the eight-row tile pays where a block has enough chunks to amortise its longer tile end: measured per launch at batch 8, profiles/r05_tall_tile.txt --
    // level-1 64 -> 64 -5 %, 128 -> 64 -6 %, 256 -> 64 @ 32 x 512 -4 %; 64 -> 64 @ 32 x 512, two four-chunk tiles per block, +3 %: left on the 64 x 4 tile)
    // fewer 64-channel tiles than CUs (u_block4 at batch 8: 128): 32-channel tiles, so that every CU has one (R2DM_F2_NARROW=0: never)
    static const bool narrow_on = !getenv("R2DM_F2_NARROW") || atoi(getenv("R2DM_F2_NARROW")) != 0;
    if (narrow_ok && narrow_on && px_tiles * (Cout / 64) < f2_cu_count() && px_tiles * (Cout / 32) >= px_tiles * (Cout / 64) * 2) return 32;
    if (tall_ok && tall_on && px_rows && Cin <= max_cin && (px_tiles / 2) * (Cout / 64) * (long)(Cin / f2::CK) >= 16L * f2_cu_count()) rows = 8;
    return 64;
}

// max_bits <- float bits of max|w[0 .. n)| (non-negative floats order like their bit patterns; NaN sorts above infinity)
__global__ void weight_absmax_kernel(const float* __restrict__ w, long n, int* __restrict__ max_bits) {
    float m = 0.f;
    bool nan = false;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = fabsf(w[i]);
        nan |= v != v;
        m = fmaxf(m, v);
    }
    m = wave_max_f32(m);
    if (__any(nan)) m = __int_as_float(0x7fc00000);
    if ((threadIdx.x & 63) == 0) atomicMax(max_bits, __float_as_int(m));
}

hipError_t launch_weight_absmax(const float* w, long n, int* max_bits, hipStream_t s) {
    hipError_t e = hipMemsetAsync(max_bits, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    weight_absmax_kernel<<<blocks, 256, 0, s>>>(w, n, max_bits);
    return hipGetLastError();
}

hipError_t launch_pack_conv_f16x2(const float* w, float* dst, int Cout, int Cin, int* range_flag, hipStream_t s, float* wscale, int co_tile, int px_rows) {
    if ((co_tile != 32 && co_tile != 64 && co_tile != 128) || Cout % co_tile || (px_rows != 4 && !(px_rows == 8 && co_tile == 64))) return hipErrorInvalidValue;
    const long total = (long)Cout * Cin * 9 * f2::NPL;
    if (wscale) {
        hipError_t e = launch_weight_absmax(w, (long)Cout * Cin * 9, reinterpret_cast<int*>(wscale), s);
        if (e != hipSuccess) return e;
    }
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    pack_conv_f16x2_kernel<<<blocks, 256, 0, s>>>(w, reinterpret_cast<unsigned short*>(dst), Cout, Cin, total, range_flag, wscale, co_tile, co_tile <= 64 && px_rows == 4);  // (the one-accumulator tiles: l planes at their true scale)
    return hipGetLastError();
}

template <int PRO, int NPLK, int MRK, int NRK, int IOM = 0>
static hipError_t launch_f2(const ConvParams& p, long tiles, hipStream_t s) {
    auto kern = conv_f16x2_kernel<PRO, NPLK, MRK, NRK, IOM>;
    using GEO = f2::Geo<MRK, NRK>;
    // The whole LDS of the CU, whatever the tile needs (the 32-channel tile: 98 KiB): a persistent block must not share its CU with a
    // workgroup of ANOTHER process.  Next to a neighbour process whose workgroups hold LDS, the 32-channel tile -- the only tile that left
    // room for them -- ended in a GPU memory fault (scripts/jobs/j314.sh .. j316.sh: not without the neighbour, not with R2DM_F2_NARROW=0);
    // the same signature as round 2's "wrong results next to a second process" of a 17 KiB-LDS kernel (LABNOTES.md section 6): whatever the
    // platform does there, a block that owns the CU's LDS is not exposed to it.  R2DM_F2_LDS_EXACT=1: experiments.
    static const bool lds_exact = getenv("R2DM_F2_LDS_EXACT") != nullptr;
    const int LDS_TOTAL = GEO::LDS_TOTAL >= 156 * 1024 || lds_exact ? GEO::LDS_TOTAL : 156 * 1024;
    constexpr int COT = GEO::COT;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        if (e != hipSuccess) return e;
#ifndef F2_SHARE_CU
        // "a block owns its CU" is what fences the open co-residency defect (root cause NOT identified) -- so it
        // is checked, once per instantiation, instead of assumed: with this launch's LDS request and the kernel's register allocation exactly ONE
        // block must fit a CU.  A later drop in register or LDS use (or -DF2_SHARE_CU / R2DM_F2_LDS_EXACT=1 builds, which skip this on purpose)
        // would otherwise re-expose the fault silently.
        if (!lds_exact) {
            int nb = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 512, (size_t)LDS_TOTAL);
            if (e != hipSuccess) return e;
            if (nb != 1) {
                fprintf(stderr, "r2dm: conv_f16x2_kernel<%d,%d,%d,%d,%d>: %d blocks would fit one CU (must be exactly 1: a block owns its CU)\n", PRO, NPLK, MRK, NRK, IOM, nb);
                return hipErrorLaunchFailure;
            }
        }
#endif
        attr_set = true;
    }
    static const bool one_tile_blocks = getenv("R2DM_F2_NONPERSISTENT") != nullptr;  // experiments: one tile per block
    const int n_cu = f2_cu_count();
    const unsigned grid = (unsigned)(tiles < n_cu || one_tile_blocks ? tiles : n_cu);
    const int nCoT = p.Cout / COT, nTw = p.W / f2::TW, nTh = p.H / GEO::TH;
    const int dmax = nCoT > nTw ? (nCoT > nTh ? nCoT : nTh) : (nTw > nTh ? nTw : nTh);
    if (tiles * dmax >= (1ll << 32)) return hipErrorInvalidValue;  // (F2Div is exact below that)
    const F2Div dv{f2_magic(nCoT), f2_magic(nTw), f2_magic(nTh)};
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_TOTAL, s, p, (int)tiles, dv);
    return hipGetLastError();
}

template <int MRK, int NRK>
static hipError_t launch_f2_tile(const ConvParams& p, hipStream_t s) {
    using GEO = f2::Geo<MRK, NRK>;
    const long tiles = (long)(p.Cout / GEO::COT) * (p.W / f2::TW) * (p.H / GEO::TH) * p.B;
    if constexpr (MRK == 2 && NRK == 2) {  // pre-split input (presplit.hip): 64 x 4 tiles (and 32 x 4 below)
        if (p.prologue == PRO_PRESPLIT) return p.pieces == 1 ? launch_f2<PRO_PRESPLIT, 1, 2, 2>(p, tiles, s) : launch_f2<PRO_PRESPLIT, 2, 2, 2>(p, tiles, s);
    }
    if constexpr (MRK == 1) {
        if (p.prologue == PRO_PRESPLIT) return p.pieces == 1 ? hipErrorInvalidValue : launch_f2<PRO_PRESPLIT, 2, 1, 2>(p, tiles, s);
    }
    {
    if (p.pieces == 1) {  // one fp16 product per MAC (the reduced-precision bulk mode)
        const int iom = (p.x16 ? 1 : 0) | (p.y16 ? 2 : 0);  // fp16 storage of the input / of the output and residual
        if (iom) {  // (the residual blocks' convolutions -- GroupNorm + SiLU input -- and the plain-input down- / up-sampling convolutions)
            if (p.prologue == PRO_AFFINE_SILU) {
                if (iom == 1) return launch_f2<PRO_AFFINE_SILU, 1, MRK, NRK, 1>(p, tiles, s);
                if (iom == 2) return launch_f2<PRO_AFFINE_SILU, 1, MRK, NRK, 2>(p, tiles, s);
                return launch_f2<PRO_AFFINE_SILU, 1, MRK, NRK, 3>(p, tiles, s);
            }
            if (p.prologue == PRO_NONE && iom == 3) return launch_f2<PRO_NONE, 1, MRK, NRK, 3>(p, tiles, s);
            return hipErrorInvalidValue;
        }
        switch (p.prologue) {
            case PRO_NONE: return launch_f2<PRO_NONE, 1, MRK, NRK>(p, tiles, s);
            case PRO_AFFINE: return launch_f2<PRO_AFFINE, 1, MRK, NRK>(p, tiles, s);
            case PRO_AFFINE_SILU: return launch_f2<PRO_AFFINE_SILU, 1, MRK, NRK>(p, tiles, s);
        }
    }
    switch (p.prologue) {
        case PRO_NONE: return launch_f2<PRO_NONE, 2, MRK, NRK>(p, tiles, s);
        case PRO_AFFINE: return launch_f2<PRO_AFFINE, 2, MRK, NRK>(p, tiles, s);
        case PRO_AFFINE_SILU: return launch_f2<PRO_AFFINE_SILU, 2, MRK, NRK>(p, tiles, s);
    }
    return hipErrorInvalidValue;
    }
}

// GroupNorm folded into this launch?  8 groups of 8 .. 64 channels over exactly the input channels, at most 4 x 256 statistics slots per
// group to read (the fold keeps them in registers: beyond that the separate gn_finalize launch is the faster way -- 128 x 2048 level 1), and
// every block's tiles inside ONE sample (the table is computed once per block) -- true at batch 8 / 2 / 1 for every layer of the network,
// checked here for whatever else arrives.
bool conv_f16x2_fold_supported(const ConvParams& p, int groups, int slots) {
    const char* fe = getenv("R2DM_GN_FOLD");  // (R2DM_GN_FOLD=0: experiments, A/B, the bit-identity test; read per call)
    const bool fold_on = !fe || atoi(fe) != 0;
    if (!fold_on || groups != 8 || p.Cin % 8 || slots < 1 || slots > 4 * 256) return false;
    const int cpg = p.Cin / 8;
    if (cpg < 8 || cpg > 64 || (cpg & (cpg - 1))) return false;
    if (!conv_f16x2_supported(p.Cin, p.Cout, p.taps, p.H, p.W, p.co_tile, p.px_rows) || p.prologue == PRO_NONE || p.prologue == PRO_PRESPLIT) return false;
    const long tps = (long)(p.Cout / p.co_tile) * (p.W / f2::TW) * (p.H / p.px_rows), tiles = tps * p.B;
    const int n_cu = f2_cu_count();
    if (tiles <= n_cu) return true;  // one tile per block
    if (getenv("R2DM_F2_NONPERSISTENT")) return true;
    const int G = n_cu;
    for (int blk = 0; blk < G; ++blk) {
        const long nIt = (tiles - blk + G - 1) / G;
        long l0 = xcd_remap(blk, (int)tiles), l1 = xcd_remap((int)(blk + (nIt - 1) * G), (int)tiles);
        if (p.reverse) { l0 = tiles - 1 - l0; l1 = tiles - 1 - l1; }
        if (l0 / tps != l1 / tps) return false;  // (a block's tile indices are monotonic: the first and the last bound the rest)
    }
    return true;
}

hipError_t launch_conv_f16x2(const ConvParams& p_in, hipStream_t s) {
    ConvParams p = p_in;
    static const int stagger = getenv("R2DM_F2_STAGGER") ? atoi(getenv("R2DM_F2_STAGGER")) : 0;
    p.stagger = stagger;
    if (!conv_f16x2_supported(p.Cin, p.Cout, p.taps, p.H, p.W, p.co_tile, p.px_rows)) return hipErrorInvalidValue;
    if (p.x.p1 && p.x.c0 % f2::CK) return hipErrorInvalidValue;  // a chunk must not straddle the concat seam
    if (p.prologue != PRO_NONE && p.prologue != PRO_PRESPLIT && p.aff == nullptr && p.gn_partial == nullptr) return hipErrorInvalidValue;
    if ((p.x16 || p.y16) && p.pieces != 1) return hipErrorInvalidValue;  // (fp16 storage exists in the one-plane mode only)
    if (p.gn_partial && (p.aff != nullptr || p.gn_cpg * 8 != p.Cin || !conv_f16x2_fold_supported(p, 8, p.gn_slots))) return hipErrorInvalidValue;
    if (p.prologue == PRO_PRESPLIT && ((p.co_tile != 64 && p.co_tile != 32) || p.px_rows != 4 || p.x.p1 != nullptr)) return hipErrorInvalidValue;
#ifndef F2_PROF
    if (p.prof != nullptr) return hipErrorInvalidValue;
#endif
    return p.co_tile == 128 ? launch_f2_tile<4, 2>(p, s) : p.co_tile == 32 ? launch_f2_tile<1, 2>(p, s) : p.px_rows == 8 ? launch_f2_tile<2, 4>(p, s) : launch_f2_tile<2, 2>(p, s);
}

}  // namespace r2dm
