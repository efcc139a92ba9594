// Weight-gradient kernels for gfx950.
//
// dW[oc][c][tap] = sum_{n,y,x} G[n][oc][y][x] * X[n][c][y*S+ky-P][x*S+kx-P]
// GEMM view per tap: D[oc][c] = sum_k A[oc][k] * B[k][c], k = output pixel.  Both operands are stored
// pixel-major ([pixel][16 channels]) so the MFMA fragments (8 consecutive k per lane) are gathered with
// the LDS transpose read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group).
// One workgroup (4 waves) = one "part" (32 oc x up to 64 cin x all taps of one conv) x one pixel split.
// wave w: cin tile = w & 1, tap group = w >> 1.  Partial sums -> workspace; wgrad_reduce sums the splits
// in a fixed order (deterministic) and scatters into the reference layout [cout][cin][kh][kw].
#include "common.h"
#include <type_traits>

namespace {

template <int KH, int STRIDE>
struct WCfg {
    static constexpr int NTAPS = KH * KH;
    static constexpr int TAPS_PER_PART = NTAPS <= 16 ? NTAPS : 10;  // 5x5: three parts of 10 / 10 / 5 taps
    static constexpr int TPG = (TAPS_PER_PART + 1) / 2;             // taps per wave
    static constexpr int PH = STRIDE == 2 ? 2 : ((KH == 3 || KH == 1) ? 8 : 4), PW = 16;  // output-pixel tile (PH k-steps of 16 pixels)
    static constexpr int IH = (PH - 1) * STRIDE + KH;
    static constexpr int IW = (PW - 1) * STRIDE + KH;
    static constexpr int GPLANE = PH * PW * 32 + 128;  // bytes; stride = 128 (mod 256): the two planes of a
    static constexpr int IPLANE_RAW = IH * IW * 32;    // 32-lane half hit disjoint bank halves
    static constexpr int IPLANE = IPLANE_RAW + ((IPLANE_RAW % 256) == 128 ? 0 : ((128 - (IPLANE_RAW % 256) + 256) % 256));
    static constexpr int G_BYTES = 2 * GPLANE;
    static constexpr int I_BYTES = 4 * IPLANE;
    static constexpr int LDS_BYTES = G_BYTES + I_BYTES;
};

#ifdef DASR_TRACE
__device__ unsigned long long* g_wtrace = nullptr;  // [grid][16] s_memtime stamps of wave 0 (slot 15/14: s_memrealtime entry/exit)
#define WTRACE(k)                                                                                        \
    do {                                                                                                 \
        if (g_wtrace && threadIdx.x == 0) g_wtrace[(size_t)blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define WTRACE(k) do {} while (0)
#endif

template <bool F16>
__device__ __forceinline__ float frag_f32(bf16x8 a, int j) {
    if constexpr (F16) return (float)__builtin_bit_cast(f16x8, a)[j];
    else return (float)a[j];
}

__device__ __forceinline__ bf16x8 frag_tr(const char* base, int off0, int off1) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((DASR_LDS bf16x4*)(base + off0));
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((DASR_LDS bf16x4*)(base + off1));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Staging helpers: one "piece" = 16 B of LDS payload (8 bf16 channels of one pixel).  For f32 sources a piece is
// assembled from two 16 B global loads (8 floats) and rounded to bf16 on the way.
template <bool F32>
struct StageReg {
    u32x4 a, b;  // b only used when F32
};

// branch-free staging loads: an out-of-range buffer offset reads as zero (halo / partial tiles / missing planes)
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
}

template <bool F32>
__device__ __forceinline__ void stage_load(StageReg<F32>& r, __amdgpu_buffer_rsrc_t rs, unsigned eoff, bool ok) {
    if constexpr (F32) {
        const unsigned o = ok ? eoff * 4u : OOB;
        r.a = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
        r.b = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 16, 0);
    } else {
        r.a = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? eoff * 2u : OOB, 0, 0);
    }
}

// F16: the f32 values are multiplied by `scale` (a power of two) and rounded to f16 instead of bf16
template <bool F32, bool F16 = false>
__device__ __forceinline__ void stage_store(const StageReg<F32>& r, char* dst, float scale = 1.f) {
    if constexpr (F32 && F16) {
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = (f16_t)(__uint_as_float(r.a[j]) * scale);
            o[4 + j] = (f16_t)(__uint_as_float(r.b[j]) * scale);
        }
        *(f16x8*)dst = o;
    } else if constexpr (F32) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = (bf16_t)__uint_as_float(r.a[j]);
            o[4 + j] = (bf16_t)__uint_as_float(r.b[j]);
        }
        *(bf16x8*)dst = o;
    } else {
        *(u32x4*)dst = r.a;
    }
}

// (4x4 kernels: 8 accumulator tiles per wave + the staging registers of an f32 source need ~290 registers: one workgroup per CU instead of
// scratch spills inside the MFMA loop -- VERDICT r02 item 10)
template <int KH, int STRIDE, bool USE_TR, bool F32, bool F16 = false>
__global__ __launch_bounds__(256, KH == 4 ? 1 : 2) void wgrad_kernel(const dasr_wgrad_part* __restrict__ parts, int nparts, int nsplit,
                                                       float* __restrict__ ws) {
    static_assert(!F16 || F32, "f16 staging converts f32 tensors");
    using C = WCfg<KH, STRIDE>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* gl = smem;
    char* il = smem + C::G_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int part_id = blockIdx.x / nsplit, split = blockIdx.x - part_id * nsplit;
    const dasr_wgrad_part P = parts[part_id];
    const int ct = wave & 1, tg = wave >> 1;
    const bool active = ct < P.n_ctiles;
    const int tiles_x = (P.Wout + C::PW - 1) / C::PW, tiles_y = (P.Hout + C::PH - 1) / C::PH;
    const int ntiles = tiles_x * tiles_y * P.N;
    const int HL = P.ups ? 2 * P.Hin : P.Hin, WL = P.ups ? 2 * P.Win : P.Win;
    constexpr int ESZ = F32 ? 4 : 2;
    constexpr int GPIX = C::PH * C::PW, IPIX = C::IH * C::IW;
    constexpr int GPIECES = 2 * GPIX * 2;          // 2 planes x pixels x 2 halves (8 channels each)
    constexpr int GR = (GPIECES + 255) / 256;
    constexpr int IPIECES_MAX = 4 * IPIX * 2;
    constexpr int IR = (IPIECES_MAX + 255) / 256;
    const int ipieces = 2 * P.n_ctiles * IPIX * 2;
    const float gsc = (F16 && P.g_scale != 0.f) ? P.g_scale : 1.f;

    f32x16 acc[C::TPG];
#pragma unroll
    for (int t = 0; t < C::TPG; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    float bsum = 0.f;
    const int gg = lane >> 4, li = lane & 15;
    const int fplane = gg & 1, khalf = gg >> 1;

    StageReg<F32> greg[GR], ireg[IR];

    auto prefetch = [&](int tile) {
        int t2 = tile;
        const int tx = t2 % tiles_x;
        t2 /= tiles_x;
        const int ty = t2 % tiles_y;
        const int n = t2 / tiles_y;
        const int oy0 = ty * C::PH, ox0 = tx * C::PW;
        const int iy0 = oy0 * STRIDE - P.pad, ix0 = ox0 * STRIDE - P.pad;
        const __amdgpu_buffer_rsrc_t gb = make_rsrc((const char*)P.g.p + (size_t)n * P.g.n_stride * ESZ);
        const __amdgpu_buffer_rsrc_t ib = make_rsrc((const char*)P.in.p + (size_t)n * P.in.n_stride * ESZ);
#pragma unroll
        for (int r = 0; r < GR; ++r) {
            const int q = tid + r * 256;
            const int half = q & 1, pix = (q >> 1) % GPIX, pl = (q >> 1) / GPIX;
            const int py = pix / C::PW, px = pix - py * C::PW;
            const int oy = oy0 + py, ox = ox0 + px;
            const bool ok = q < GPIECES && oy < P.Hout && ox < P.Wout && pl < P.g_planes;
            stage_load<F32>(greg[r], gb, (unsigned)(pl * (int)P.g.cb_stride + (oy * P.Wout + ox) * 16 + half * 8), ok);
        }
#pragma unroll
        for (int r = 0; r < IR; ++r) {
            const int q = tid + r * 256;
            const int half = q & 1, pix = (q >> 1) % IPIX, pl = (q >> 1) / IPIX;
            const int iy = pix / C::IW, ix = pix - iy * C::IW;
            const int gy = iy0 + iy, gx = ix0 + ix;
            const bool ok = q < ipieces && gy >= 0 && gy < HL && gx >= 0 && gx < WL && pl < P.in_planes;
            const int sy = P.ups ? gy >> 1 : gy, sx = P.ups ? gx >> 1 : gx;
            stage_load<F32>(ireg[r], ib, (unsigned)(pl * (int)P.in.cb_stride + (sy * P.Win + sx) * 16 + half * 8), ok);
        }
    };
    char* const dummy = smem + C::LDS_BYTES;  // 16 B slot that swallows the tail threads' stores (no divergent branch)
    // f32 gradients: the bias gradient (sum over pixels, heavy cancellation) is accumulated from the UNROUNDED values
    // while staging; thread t always stages the same 8 channels (plane, half) in round r.
    float bacc[GR][8];
#pragma unroll
    for (int r = 0; r < GR; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) bacc[r][j] = 0.f;
    auto commit = [&]() {
#pragma unroll
        for (int r = 0; r < GR; ++r) {
            const int q = tid + r * 256;
            const int half = q & 1, pix = (q >> 1) % GPIX, pl = (q >> 1) / GPIX;
            stage_store<F32, F16>(greg[r], q < GPIECES ? gl + pl * C::GPLANE + pix * 32 + half * 16 : dummy, gsc);
            if constexpr (F32) {   // (scaled like the staged values: the caller's reduce scale undoes g_scale for weights and bias alike)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bacc[r][j] += __uint_as_float(greg[r].a[j]) * gsc;
                    bacc[r][4 + j] += __uint_as_float(greg[r].b[j]) * gsc;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < IR; ++r) {
            const int q = tid + r * 256;
            const int half = q & 1, pix = (q >> 1) % IPIX, pl = (q >> 1) / IPIX;
            stage_store<F32, F16>(ireg[r], q < ipieces ? il + pl * C::IPLANE + pix * 32 + half * 16 : dummy);
        }
    };

    if (split < ntiles) prefetch(split);
    for (int tile = split; tile < ntiles; tile += nsplit) {
        __syncthreads();  // previous tile's LDS reads are done
        commit();
        __syncthreads();
        if (tile + nsplit < ntiles) prefetch(tile + nsplit);  // global loads fly under the MFMAs below
        if (!active) continue;
#pragma unroll 2
        for (int r = 0; r < C::PH; ++r) {
            bf16x8 a;
            if constexpr (USE_TR) {
                const int o0 = fplane * C::GPLANE + (r * C::PW + 8 * khalf + (li >> 2)) * 32 + (li & 3) * 8;
                a = frag_tr(gl, o0, o0 + 4 * 32);
            } else {
                const char* b0 = gl + fplane * C::GPLANE + (r * C::PW + 8 * khalf) * 32 + li * 2;
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = *(const bf16_t*)(b0 + j * 32);
            }
            if constexpr (!F32) {
                if (P.want_bias && tg == 0 && ct == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum += (float)a[j];
                }
            }
#pragma unroll
            for (int t = 0; t < C::TPG; ++t) {
                const int tl = tg * C::TPG + t;
                const int tap = P.tap0 + tl;
                if (tl < C::TAPS_PER_PART && tap < C::NTAPS) {  // uniform per wave
                    const int ky = tap / KH, kx = tap - ky * KH;
                    bf16x8 b;
                    const int prow = (r * STRIDE + ky) * C::IW + kx;
                    if constexpr (USE_TR) {
                        const int o0 = (ct * 2 + fplane) * C::IPLANE + (prow + (8 * khalf + (li >> 2)) * STRIDE) * 32 + (li & 3) * 8;
                        b = frag_tr(il, o0, o0 + 4 * STRIDE * 32);
                    } else {
                        const char* b0 = il + (ct * 2 + fplane) * C::IPLANE + (prow + 8 * khalf * STRIDE) * 32 + li * 2;
#pragma unroll
                        for (int j = 0; j < 8; ++j) b[j] = *(const bf16_t*)(b0 + j * STRIDE * 32);
                    }
                    acc[t] = mfma16<F16>(a, b, acc[t]);
                }
            }
        }
    }

    // ---- write partials: ws[part][split][tap][oc 32][cin 64] ----
    if (active) {
        float* w = ws + P.ws_off + (size_t)split * C::TAPS_PER_PART * 32 * 64;
        const int cin = ct * 32 + (lane & 31), h = lane >> 5;
#pragma unroll
        for (int t = 0; t < C::TPG; ++t) {
            const int tl = tg * C::TPG + t;
            if (tl < C::TAPS_PER_PART && P.tap0 + tl < C::NTAPS) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int oc = (j & 3) + 8 * (j >> 2) + 4 * h;
                    w[((size_t)tl * 32 + oc) * 64 + cin] = acc[t][j];
                }
            }
        }
        if constexpr (!F32) {
            if (P.want_bias && tg == 0 && ct == 0) {
                const float tot = bsum + __shfl_xor(bsum, 32, 64);
                if (lane < 32) ws[P.ws_bias_off + (size_t)split * 32 + lane] = tot;
            }
        }
    }
    if constexpr (F32) {
        if (P.want_bias) {  // uniform: deterministic fixed-order reduction of the per-thread fp32 partials through LDS
            __syncthreads();
            float* red = (float*)smem;  // [256][GR*8]
#pragma unroll
            for (int r = 0; r < GR; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[tid * (GR * 8) + r * 8 + j] = (tid + r * 256 < GPIECES) ? bacc[r][j] : 0.f;
            __syncthreads();
            if (tid < 32) {
                const int pl = tid >> 4, half = (tid >> 3) & 1, j = tid & 7;
                float tot = 0.f;
                for (int t = 0; t < 256; ++t) {
#pragma unroll
                    for (int r = 0; r < GR; ++r) {
                        const int q = t + r * 256;
                        if (q < GPIECES && (q & 1) == half && (q >> 1) / GPIX == pl) tot += red[t * (GR * 8) + r * 8 + j];
                    }
                }
                ws[P.ws_bias_off + (size_t)split * 32 + tid] = tot;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad v3 (3x3, stride 1): one workgroup = 6 waves = one 64-channel INPUT block x up to three 32-oc tiles of the
// gradient slab x all 9 taps.  wave w: oc tile = w >> 1, cin tile = w & 1, 9 accumulators.  Compared with the
// 4-wave kernel above this halves the bytes staged per MFMA (the input block is shared by three oc tiles instead of
// one): the dense-block wgrad is bound by L2/HBM traffic, not by MFMA issue.
// workspace per part: [split][tap][ot 3][oc 32][cin 64]; bias: [split][96]
// ---------------------------------------------------------------------------------------------------------------
// workgroup -> (part, pixel split).  nsplit_flags: bits 0-15 nsplit, bits 16-23 ppu, bit 24 staggered prefetch (wgrad3_kernel).
// ppu = 0: block = part * nsplit + split; with nsplit a multiple of 8 every part's workgroup of one pixel split runs on XCD split % 8
// (workgroup b -> XCD b % 8) and the G / X tiles the parts share are fetched into ONE L2.
// ppu > 0 (grouped launches of many RDBs with few splits, nsplit not a multiple of 8): the parts come in units of ppu consecutive parts that
// read the same slab pair (one RDB); unit u = (rdb, split) is placed on XCD u % 8 as a whole: block b -> xcd = b % 8, slot = b / 8,
// unit = (slot / ppu) * 8 + xcd, part = rdb * ppu + slot % ppu.  The host only sets ppu when (nparts / ppu) * nsplit is a multiple of 8.
__device__ __forceinline__ void w3_block_map(int nsplit_flags, int& part_id, int& split) {
    const int nsplit = nsplit_flags & 0xffff, ppu = (nsplit_flags >> 16) & 0xff;
    const int b = blockIdx.x;
    if (ppu == 0) {
        part_id = b / nsplit;
        split = b - part_id * nsplit;
    } else {
        const int xcd = b & 7, slot = b >> 3;
        const int su = slot / ppu, pin = slot - su * ppu;
        const int unit = su * 8 + xcd;
        const int rdb = unit / nsplit;
        split = unit - rdb * nsplit;
        part_id = rdb * ppu + pin;
    }
}

struct W3 {
    static constexpr int NTAPS = 9, PH = 8, PW = 16, IH = 10, IW = 18;
    static constexpr int GPIX = PH * PW, IPIX = IH * IW;
    static constexpr int GPLANE = GPIX * 32 + 128;
    static constexpr int IPLANE = IPIX * 32;  // 5760 = 128 (mod 256)
    static constexpr int G_BYTES = 6 * GPLANE, I_BYTES = 4 * IPLANE;
    static constexpr int LDS_BYTES = G_BYTES + I_BYTES;
    static constexpr int NT = 768;  // 12 waves: (oc tile, cin tile) pair = wave % 6, tap half = wave / 6 -> 3 waves per SIMD
    static constexpr int GPIECES = 6 * GPIX * 2, IPIECES = 4 * IPIX * 2;
    static constexpr int GR = (GPIECES + NT - 1) / NT, IR = (IPIECES + NT - 1) / NT;
};

// One workgroup = one part (64 input channels x up to three 32-oc tiles of one gradient tensor) x one pixel split.
// Wave w: pair = w % 6 -> (oc tile ot = pair / 2, cin tile ct = pair % 2); taps 0..4 (w < 6) or 5..8 (w >= 6) of that pair's
// 3x3 weight gradient, one 32x32 accumulator per tap.  Per 8x16-pixel tile: G and the X halo tile are staged into LDS by all
// 768 threads (register prefetch one tile ahead; per-thread piece geometry is computed once), then per pixel row r and tap:
// A = G^T fragment [oc][16 pixels], B = X fragment [16 pixels shifted by the tap][cin], both gathered with ds_read_b64_tr_b16.
// F16: the 16-bit tensors hold f16 (HR tail of the generator in f16 storage, gradients pre-scaled by a power of two): f16 MFMA
// ABL (instantiated != 0 only under -DDASR_BENCH = libdasr_hip_ablate.so; WRONG results, timing only): bit 0 no LDS commit / barriers after the first
// tile, 1 no fragment reads after the first, 2 no MFMA, 3 no global prefetch -- what does each component of a tile cost?
template <bool USE_TR, bool F32, bool F16 = false, int ABL = 0>
__global__ __launch_bounds__(768, 1) void wgrad3_kernel(const dasr_wgrad_part* __restrict__ parts, int nparts, int nsplit_flags,
                                                        float* __restrict__ ws) {
    static_assert(!F16 || !F32, "wgrad3 F16: 16-bit f16 tensors");
    using C = W3;
    const int nsplit = nsplit_flags & 0xffff;
    const bool g_stagger_flag = (nsplit_flags >> 24) & 1;  // A/B: staggered in-compute prefetch issue
    constexpr int abl = ABL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* gl = smem;
    char* il = smem + C::G_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int part_id, split;
    w3_block_map(nsplit_flags, part_id, split);
    const dasr_wgrad_part P = parts[part_id];
    const int pair = wave % 6, th = wave / 6;
    const int ot = pair >> 1, ct = pair & 1;
    const int n_ot = (P.g_planes + 1) >> 1;  // oc tiles present in this part (planes are 16 channels)
    const bool active = ct < P.n_ctiles && ot < n_ot;
    const int tiles_x = (P.Wout + C::PW - 1) / C::PW, tiles_y = (P.Hout + C::PH - 1) / C::PH;
    const int ntiles = tiles_x * tiles_y * P.N;
    const int HL = P.ups ? 2 * P.Hin : P.Hin, WL = P.ups ? 2 * P.Win : P.Win;
    constexpr int ESZ = F32 ? 4 : 2;
    const int gpieces = P.g_planes * C::GPIX * 2, ipieces = 2 * P.n_ctiles * C::IPIX * 2;
#ifdef DASR_TRACE
    if (g_wtrace && threadIdx.x == 0) g_wtrace[(size_t)blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memrealtime();
#endif
    WTRACE(0);

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    float bsum = 0.f;
    const int gg = lane >> 4, li = lane & 15;
    const int fplane = gg & 1, khalf = gg >> 1;
    StageReg<F32> greg[C::GR], ireg[C::IR];
    float bacc[F32 ? C::GR : 1][8];
    if constexpr (F32) {
#pragma unroll
        for (int r = 0; r < C::GR; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) bacc[r][j] = 0.f;
    }
    char* const dummy = smem + C::LDS_BYTES;

    // per-thread piece geometry (tile independent): pixel coordinates inside the tile, element offset relative to the tile origin,
    // LDS destination
    int gpy[C::GR], gpx[C::GR], geo[C::GR], ipy[C::IR], ipx[C::IR], ipl[C::IR], ihalf[C::IR];
    char* gdst[C::GR];
    char* idst[C::IR];
#pragma unroll
    for (int r = 0; r < C::GR; ++r) {
        const int q = tid + r * C::NT;
        const int half = q & 1, pix = (q >> 1) % C::GPIX, pl = (q >> 1) / C::GPIX;
        gpy[r] = q < gpieces ? pix / C::PW : 0x40000000;  // out-of-range piece: never valid
        gpx[r] = pix % C::PW;
        geo[r] = pl * (int)P.g.cb_stride + (gpy[r] * P.Wout + gpx[r]) * 16 + half * 8;
        gdst[r] = q < gpieces ? gl + pl * C::GPLANE + pix * 32 + half * 16 : dummy;
    }
#pragma unroll
    for (int r = 0; r < C::IR; ++r) {
        const int q = tid + r * C::NT;
        const int pix = (q >> 1) % C::IPIX, pl = (q >> 1) / C::IPIX;
        ihalf[r] = q & 1;
        ipl[r] = pl;
        const bool ok = (q < ipieces) & (pl < P.in_planes);
        ipy[r] = ok ? pix / C::IW : 0x40000000;
        ipx[r] = pix % C::IW;
        idst[r] = q < ipieces ? il + pl * C::IPLANE + pix * 32 + (q & 1) * 16 : dummy;
    }

    auto prefetch = [&](int tile) {
        int t2 = tile;
        const int tx = t2 % tiles_x;
        t2 /= tiles_x;
        const int ty = t2 % tiles_y;
        const int n = t2 / tiles_y;
        const int oy0 = ty * C::PH, ox0 = tx * C::PW;
        const int iy0 = oy0 - P.pad, ix0 = ox0 - P.pad;
        const __amdgpu_buffer_rsrc_t gb = make_rsrc((const char*)P.g.p + (size_t)n * P.g.n_stride * ESZ);
        const __amdgpu_buffer_rsrc_t ib = make_rsrc((const char*)P.in.p + (size_t)n * P.in.n_stride * ESZ);
        const int gorg = (oy0 * P.Wout + ox0) * 16;
#pragma unroll
        for (int r = 0; r < C::GR; ++r) {
            const bool ok = (oy0 + gpy[r] < P.Hout) & (ox0 + gpx[r] < P.Wout);
            stage_load<F32>(greg[r], gb, (unsigned)(gorg + geo[r]), ok);
        }
#pragma unroll
        for (int r = 0; r < C::IR; ++r) {
            const int gy = iy0 + ipy[r], gx = ix0 + ipx[r];
            const bool ok = (gy >= 0) & (gy < HL) & (gx >= 0) & (gx < WL);
            const int sy = P.ups ? gy >> 1 : gy, sx = P.ups ? gx >> 1 : gx;
            stage_load<F32>(ireg[r], ib, (unsigned)(ipl[r] * (int)P.in.cb_stride + (sy * P.Win + sx) * 16 + ihalf[r] * 8), ok);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int r = 0; r < C::GR; ++r) {
            stage_store<F32>(greg[r], gdst[r]);
            if constexpr (F32) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bacc[r][j] += __uint_as_float(greg[r].a[j]);
                    bacc[r][4 + j] += __uint_as_float(greg[r].b[j]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < C::IR; ++r) stage_store<F32>(ireg[r], idst[r]);
    };

    const int gbase = (ot * 2 + fplane) * C::GPLANE + (8 * khalf + (li >> 2)) * 32 + (li & 3) * 8;
    const int ibase = (ct * 2 + fplane) * C::IPLANE + (8 * khalf + (li >> 2)) * 32 + (li & 3) * 8;
    // one tile of MFMAs for the tap range [T0, T0 + NA): flat software pipeline, fragments of step i+1 requested before MFMA i
    // next_tile >= 0: this wave requests the next tile's staging loads at MFMA step 5 * (pair index), so the twelve waves' requests
    // are spread over the compute phase instead of hitting the texture path together before it
    auto compute = [&](auto t0c, auto nac, int next_tile) {
        constexpr int T0 = decltype(t0c)::value, NA = decltype(nac)::value;
        if constexpr (USE_TR) {
            bf16x8 a[2], b[2];
            a[0] = frag_tr(gl, gbase, gbase + 4 * 32);
            {
                constexpr int ky = T0 / 3, kx = T0 - ky * 3;
                const int o = ibase + (ky * C::IW + kx) * 32;
                b[0] = frag_tr(il, o, o + 4 * 32);
            }
            if constexpr (abl & 2) {
                a[1] = a[0];
                b[1] = b[0];
            }
            const int pf_step = next_tile >= 0 ? 5 * pair : -1;
#pragma unroll
            for (int i = 0; i < C::PH * NA; ++i) {
                const int r = i / NA, t = T0 + (i - r * NA);
                if (i % 5 == 0 && i <= 25) {
                    if (i == pf_step && !(abl & 8)) prefetch(next_tile);
                }
                if (i + 1 < C::PH * NA && !(abl & 2)) {
                    const int r1 = (i + 1) / NA, t1 = T0 + ((i + 1) - r1 * NA);
                    const int ky1 = t1 / 3, kx1 = t1 - ky1 * 3;
                    const int o1 = ibase + ((r1 + ky1) * C::IW + kx1) * 32;
                    b[(i + 1) & 1] = frag_tr(il, o1, o1 + 4 * 32);
                    if (t1 == T0) {
                        const int g1 = gbase + r1 * C::PW * 32;
                        a[r1 & 1] = frag_tr(gl, g1, g1 + 4 * 32);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(abl & 4)) acc[t - T0] = mfma16<F16>(a[r & 1], b[i & 1], acc[t - T0]);
                else asm volatile("" ::"v"(a[r & 1]), "v"(b[i & 1]));
                if constexpr (!F32 && T0 == 0) {
                    if (t == 2 && P.want_bias && ct == 0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) bsum += frag_f32<F16>(a[r & 1], j);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 2
            for (int r = 0; r < C::PH; ++r) {
                bf16x8 a;
                const char* b0 = gl + (ot * 2 + fplane) * C::GPLANE + (r * C::PW + 8 * khalf) * 32 + li * 2;
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = *(const bf16_t*)(b0 + j * 32);
                if constexpr (!F32 && T0 == 0) {
                    if (P.want_bias && ct == 0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) bsum += frag_f32<F16>(a, j);
                    }
                }
#pragma unroll
                for (int t = T0; t < T0 + NA; ++t) {
                    const int ky = t / 3, kx = t - ky * 3;
                    bf16x8 b;
                    const int prow = (r + ky) * C::IW + kx;
                    const char* b1 = il + (ct * 2 + fplane) * C::IPLANE + (prow + 8 * khalf) * 32 + li * 2;
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[j] = *(const bf16_t*)(b1 + j * 32);
                    acc[t - T0] = mfma16<F16>(a, b, acc[t - T0]);
                }
            }
        }
    };

    if (split < ntiles) prefetch(split);
    WTRACE(1);
    int it = 0;
    for (int tile = split; tile < ntiles; tile += nsplit, ++it) {
        if (it == 2) WTRACE(2);
        if (!(abl & 1) || it == 0) {
            __syncthreads();
            if (it == 2) WTRACE(3);
            commit();
            if (it == 2) WTRACE(4);
            __syncthreads();
        }
        if (it == 2) WTRACE(5);
        const int next_tile = tile + nsplit < ntiles ? tile + nsplit : -1;
        const bool stagger = USE_TR && active && g_stagger_flag;
        if (next_tile >= 0 && !stagger && !(abl & 8)) prefetch(next_tile);
        if (it == 2) WTRACE(6);
        if (!active) continue;
        if (th == 0) compute(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{}, stagger ? next_tile : -1);
        else compute(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{}, stagger ? next_tile : -1);
        if (it == 2) WTRACE(7);
    }

    WTRACE(8);
    if (active) {
        float* w = ws + P.ws_off + (size_t)split * 9 * 3 * 2048 + (size_t)ot * 2048;
        const int cin = ct * 32 + (lane & 31), h = lane >> 5;
        const int t0 = th * 5, na = th ? 4 : 5;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            if (t < na) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int oc = (j & 3) + 8 * (j >> 2) + 4 * h;
                    w[(size_t)(t0 + t) * 3 * 2048 + oc * 64 + cin] = acc[t][j];
                }
            }
        }
        if constexpr (!F32) {
            if (P.want_bias && ct == 0 && th == 0) {
                const float tot = bsum + __shfl_xor(bsum, 32, 64);
                if (lane < 32) ws[P.ws_bias_off + (size_t)split * 96 + ot * 32 + lane] = tot;
            }
        }
    }
    if constexpr (F32) {
        if (P.want_bias) {
            __syncthreads();
            float* red = (float*)smem;  // [768][GR*8] floats = 49 KB <= LDS_BYTES
#pragma unroll
            for (int r = 0; r < C::GR; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[tid * (C::GR * 8) + r * 8 + j] = (tid + r * C::NT < gpieces) ? bacc[r][j] : 0.f;
            __syncthreads();
            if (tid < 96) {
                const int pl = tid >> 4, half = (tid >> 3) & 1, j = tid & 7;
                float tot = 0.f;
                for (int t = 0; t < C::NT; ++t) {
#pragma unroll
                    for (int r = 0; r < C::GR; ++r) {
                        const int q = t + r * C::NT;
                        if (q < gpieces && (q & 1) == half && (q >> 1) / C::GPIX == pl) tot += red[t * (C::GR * 8) + r * 8 + j];
                    }
                }
                ws[P.ws_bias_off + (size_t)split * 96 + tid] = tot;
            }
        }
    }
    WTRACE(9);
#ifdef DASR_TRACE
    if (g_wtrace && threadIdx.x == 0) g_wtrace[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---------------------------------------------------------------------------------------------------
// wgrad3 with LDS-DMA staging (bf16 tensors, transpose reads): same decomposition as wgrad3_kernel, but the G tile and the X
// halo tile of the NEXT pixel tile are written straight into the other LDS buffer by `buffer_load_dwordx4 ... lds` while the
// current tile is multiplied: no staging registers, no ds_write, one barrier per tile.  48 one-KiB DMA instructions per tile
// (6 G planes x 4, 4 X planes x 6), four per wave; each lane's source address is tile origin + a per-lane constant.
// ---------------------------------------------------------------------------------------------------
// One LDS-DMA instruction (64 lanes x 16 bytes -> 1 KiB of LDS at `lds_addr`, lane l at +16 l) issued through inline assembly.
// Why not __builtin_amdgcn_raw_ptr_buffer_load_lds here: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 that
// follows a pending LDS-DMA in program order (the transpose-read intrinsic carries no memory operand, so the waitcnt pass must assume it reads
// what the DMA writes) -- with the DMA of the NEXT tile issued between the fragment reads of the current one, every k-step then waited for a full
// global round trip (first wgrad4 build: 3.4 ms per launch instead of 2.2).  Issued from assembly WITHOUT a "memory" clobber (with one, the waitcnt
// pass treats the statement itself as a pending vector-memory access and waits just the same) the compiler does not see the DMA at all; its
// completion is awaited by the explicit `s_waitcnt vmcnt(0)` + barrier that ends every tile, which volatile asm cannot cross.
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc) : "m0");
}

struct W3G {
    static constexpr int PH = 8, PW = 16, IH = 10, IW = 18, GPIX = PH * PW, IPIX = IH * IW;
    static constexpr int GPLANE = GPIX * 32 + 128;           // 4 DMA instructions per plane
    static constexpr int IPLANE = 6 * 1024 + 128;            // 6 DMA instructions per plane (180 pixels = 5.6 KiB), = 128 (mod 256)
    static constexpr int G_BYTES = 6 * GPLANE, I_BYTES = 4 * IPLANE;
    static constexpr int BUF_BYTES = G_BYTES + I_BYTES;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
    static constexpr int NT = 768;
};

// per-lane geometry of one of the wave's four DMA instructions (j = wave + 12 k), tile independent
struct W3GPiece {
    int dy, dx;   // pixel position inside the tile (G) / halo tile (X); dy huge = never valid
    int rel;      // element offset of the lane's 8 channels relative to the tile origin pixel
    int lds;      // LDS byte offset of the instruction inside a buffer (wave uniform)
};

__device__ __forceinline__ W3GPiece w3g_piece(const dasr_wgrad_part& P, int j, int lane) {
    using C = W3G;
    W3GPiece g;
    if (j < 24) {
        const int pl = j >> 2, sub = j & 3;
        const int q = sub * 64 + lane, pix = q >> 1, half = q & 1;
        g.dy = pl < P.g_planes ? (pix >> 4) : 0x40000000;
        g.dx = pix & 15;
        g.rel = pl * (int)P.g.cb_stride + half * 8;
        g.lds = pl * C::GPLANE + sub * 1024;
    } else {
        const int jj = j - 24, pl = jj / 6, sub = jj - pl * 6;
        const int q = sub * 64 + lane, pix = q >> 1, half = q & 1;
        const int iy = pix / C::IW;
        g.dy = ((pix < C::IPIX) & (pl < P.in_planes) & (pl < 2 * P.n_ctiles)) ? iy : 0x40000000;
        g.dx = pix - iy * C::IW;
        g.rel = pl * (int)P.in.cb_stride + half * 8;
        g.lds = C::G_BYTES + pl * C::IPLANE + sub * 1024;
    }
    return g;
}

struct W3GTile {
    __amdgpu_buffer_rsrc_t gb, ib;
    int oy0, ox0;
};

__device__ __forceinline__ W3GTile w3g_tile(const dasr_wgrad_part& P, int tile, int tiles_x, int tiles_y) {
    W3GTile T;
    int t2 = tile;
    const int tx = t2 % tiles_x;
    t2 /= tiles_x;
    const int ty = t2 % tiles_y;
    const int n = t2 / tiles_y;
    T.oy0 = ty * W3G::PH;
    T.ox0 = tx * W3G::PW;
    T.gb = make_rsrc((const bf16_t*)P.g.p + (size_t)n * P.g.n_stride);
    T.ib = make_rsrc((const bf16_t*)P.in.p + (size_t)n * P.in.n_stride);
    return T;
}

// the wave's k-th 1-KiB DMA instruction (j = wave + 12 k) of pixel tile T into `buf`
__device__ __forceinline__ void w3g_dma(const dasr_wgrad_part& P, const W3GTile& T, const W3GPiece& pc, int j, char* buf, int HL, int WL) {
    if (j < 24) {
        const int oy = T.oy0 + pc.dy, ox = T.ox0 + pc.dx;
        const bool ok = (oy < P.Hout) & (ox < P.Wout);
        const unsigned off = ok ? (unsigned)((pc.rel + (oy * P.Wout + ox) * 16) * 2) : OOB;
        lds_dma16(T.gb, (unsigned)(size_t)(DASR_LDS char*)buf + pc.lds, off);
    } else {
        const int gy = T.oy0 - P.pad + pc.dy, gx = T.ox0 - P.pad + pc.dx;
        const bool ok = (gy >= 0) & (gy < HL) & (gx >= 0) & (gx < WL);
        const int sy = P.ups ? gy >> 1 : gy, sx = P.ups ? gx >> 1 : gx;
        const unsigned off = ok ? (unsigned)((pc.rel + (sy * P.Win + sx) * 16) * 2) : OOB;
        lds_dma16(T.ib, (unsigned)(size_t)(DASR_LDS char*)buf + pc.lds, off);
    }
}

#ifdef DASR_BENCH   // measured alternative of round 3 (libdasr_hip_ablate.so only); the product's 3x3 weight-gradient kernel is wgrad3_ld_kernel
__global__ __launch_bounds__(768, 1) void wgrad3_glds_kernel(const dasr_wgrad_part* __restrict__ parts, int nparts, int nsplit_flags, float* __restrict__ ws) {
    using C = W3G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsplit = nsplit_flags & 0xffff;
    int part_id, split;
    w3_block_map(nsplit_flags, part_id, split);
    const dasr_wgrad_part P = parts[part_id];
    const int pair = wave % 6, th = wave / 6;
    const int ot = pair >> 1, ct = pair & 1;
    const int n_ot = (P.g_planes + 1) >> 1;
    const bool active = ct < P.n_ctiles && ot < n_ot;
    const int tiles_x = (P.Wout + C::PW - 1) / C::PW, tiles_y = (P.Hout + C::PH - 1) / C::PH;
    const int ntiles = tiles_x * tiles_y * P.N;
    const int HL = P.ups ? 2 * P.Hin : P.Hin, WL = P.ups ? 2 * P.Win : P.Win;
#ifdef DASR_TRACE
    if (g_wtrace && threadIdx.x == 0) g_wtrace[(size_t)blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memrealtime();
#endif
    WTRACE(0);
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    float bsum = 0.f;
    const int gg = lane >> 4, li = lane & 15;
    const int fplane = gg & 1, khalf = gg >> 1;
    const int gbase = (ot * 2 + fplane) * C::GPLANE + (8 * khalf + (li >> 2)) * 32 + (li & 3) * 8;
    const int ibase = C::G_BYTES + (ct * 2 + fplane) * C::IPLANE + (8 * khalf + (li >> 2)) * 32 + (li & 3) * 8;

    W3GPiece pc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        pc[k] = w3g_piece(P, wave + 12 * k, lane);
        pc[k].lds = __builtin_amdgcn_readfirstlane(pc[k].lds);
    }
    // taps of this wave: th = 0 -> 0..4, th = 1 -> 5..8 (the fifth slot repeats tap 8's address and its MFMA is skipped): one code
    // path for both halves (two specialised copies cost 60 VGPRs and spilled)
    int tb[5];
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        const int t = th ? (a < 4 ? 5 + a : 8) : a;
        tb[a] = ibase + ((t / 3) * C::IW + (t % 3)) * 32;
    }
    auto compute = [&](const char* buf, char* nbuf, const W3GTile& T, bool more) {
        bf16x8 a[2], b[2];
        a[0] = frag_tr(buf, gbase, gbase + 4 * 32);
        b[0] = frag_tr(buf, tb[0], tb[0] + 4 * 32);
#pragma unroll
        for (int i = 0; i < C::PH * 5; ++i) {
            const int r = i / 5, t = i - r * 5;
            if (i + 1 < C::PH * 5) {
                const int r1 = (i + 1) / 5, t1 = (i + 1) - r1 * 5;
                const int o1 = tb[t1] + r1 * C::IW * 32;
                b[(i + 1) & 1] = frag_tr(buf, o1, o1 + 4 * 32);
                if (t1 == 0) {
                    const int g1 = gbase + r1 * C::PW * 32;
                    a[r1 & 1] = frag_tr(buf, g1, g1 + 4 * 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t < 4 || th == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r & 1], b[i & 1], acc[t], 0, 0, 0);
            if (t == 2 && P.want_bias && ct == 0 && th == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += (float)a[r & 1][j];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (split < ntiles) {
        const W3GTile T0 = w3g_tile(P, split, tiles_x, tiles_y);
#pragma unroll
        for (int k = 0; k < 4; ++k) w3g_dma(P, T0, pc[k], wave + 12 * k, smem, HL, WL);
    }
    WTRACE(1);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    int it = 0;
    for (int tile = split; tile < ntiles; tile += nsplit, ++it) {
        char* buf = smem + (it & 1) * C::BUF_BYTES;
        char* nbuf = smem + ((it + 1) & 1) * C::BUF_BYTES;
        if (it == 2) WTRACE(2);
        const bool more = tile + nsplit < ntiles;
        const W3GTile T = w3g_tile(P, more ? tile + nsplit : tile, tiles_x, tiles_y);
        if (it == 2) WTRACE(6);
        if (more) {  // the next tile's pieces are requested up front (issuing them between the MFMAs stalled the stream: measured slower)
#pragma unroll
            for (int k = 0; k < 4; ++k) w3g_dma(P, T, pc[k], wave + 12 * k, nbuf, HL, WL);
        }
        if (active) compute(buf, nbuf, T, more);
        if (it == 2) WTRACE(7);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
    WTRACE(8);
    if (active) {
        float* w = ws + P.ws_off + (size_t)split * 9 * 3 * 2048 + (size_t)ot * 2048;
        const int cin = ct * 32 + (lane & 31), h = lane >> 5;
        const int t0 = th * 5, na = th ? 4 : 5;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            if (t < na) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int oc = (j & 3) + 8 * (j >> 2) + 4 * h;
                    w[(size_t)(t0 + t) * 3 * 2048 + oc * 64 + cin] = acc[t][j];
                }
            }
        }
        if (P.want_bias && ct == 0 && th == 0) {
            const float tot = bsum + __shfl_xor(bsum, 32, 64);
            if (lane < 32) ws[P.ws_bias_off + (size_t)split * 96 + ot * 32 + lane] = tot;
        }
    }
    WTRACE(9);
#ifdef DASR_TRACE
    if (g_wtrace && threadIdx.x == 0) g_wtrace[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memrealtime();
#endif
}

#endif  // DASR_BENCH

// ---------------------------------------------------------------------------------------------------------------
// wgrad3 with LOADER WAVES (round 3): the decomposition of wgrad3_glds_kernel (12 compute waves: oc tile x cin tile x tap half, 5 accumulators,
// transposed LDS reads, double-buffered LDS image filled by LDS-DMA) plus FOUR waves, one per SIMD, that do nothing but issue the 48 DMA
// instructions of the next pixel tile and wait for them.  The round-3 ablations (profiles/r03_wgrad_ablation.txt) showed that the weight gradient
// pays the SUM of its MFMA time and of its L2 -> LDS transfer time: a wave that issues a load into a busy memory pipe stalls, and its MFMAs with
// it.  Here the compute waves never touch vector memory inside the tile loop; the loaders stall instead, and a stalled loader costs no MFMA slot.
// 16 waves x 128 registers: the compute side fits because nothing is staged through registers.
// ---------------------------------------------------------------------------------------------------------------
struct W3L {
    static constexpr int NT = 1024, NLD = 4, NDMA = 12;   // 12 compute waves + 4 loaders; 48 DMA instructions per tile, 12 per loader
    static constexpr int NBUF = 3;                        // ring of three tile images (151 KB): the loaders run one tile ahead of the wait
    static constexpr int LDS_BYTES = NBUF * W3G::BUF_BYTES;
};

// The k-steps of one pixel tile for one compute wave, REGISTER-WINDOW form (round 6).  The 12-wave decomposition of round 3 read 6 fragments per
// k-step (5 X + 1 G) for 5 (th = 0) or 4 (th = 1) MFMAs: 576 KB of transposed LDS reads per tile and CU + 47 KB of LDS-DMA writes = 4 870 cycles of the
// LDS port (128 B / clk) against 3 840 cycles of MFMA -- the kernel sat at 71 % MFMA busy because the LDS port, not the matrix pipe, bounded it
// (profiles/r03_wgrad_ablation.txt: MFMA + fragment reads alone 1 686 us against 1 275 us MFMA-bound; profiles/r06f_pmc_mfma_busy.txt).
// Here a wave owns taps that share a COLUMN of the 3 x 3 stencil, so that the X fragment of halo row r + dy is the same registers for every (r, dy)
// with the same sum: th = 0: (dy 0..2, dx 0) + (dy 0..1, dx 1); th = 1: (dy 0..2, dx 2) + (2, 1).  Per k-step: ONE new fragment per column + the G
// fragment = 3 reads for 5 / 4 MFMAs (312 KB per tile and CU).  A fragment is reloaded IN PLACE right behind the last MFMA that reads it (dy = 0) and is
// next needed as dy = 2 (dy = 1 for the two-row column) of the following k-step: 28 fragment registers instead of 36.
struct W3Win {
    bf16x8 xa[3], xb[2], g[2];   // column A: halo rows r, r + 1, r + 2 (ring of three); column B: two rows (th 0) / one row, double-buffered (th 1); G row r, r + 1
};
template <int TH>
__device__ __forceinline__ bf16x8 w3_win_x(const char* buf, int ibase, int R, int dx) {
    const int o = ibase + (R * W3G::IW + dx) * 32;
    return frag_tr(buf, o, o + 4 * 32);
}
__device__ __forceinline__ bf16x8 w3_win_g(const char* buf, int gbase, int r) {
    const int o = gbase + r * W3G::PW * 32;
    return frag_tr(buf, o, o + 4 * 32);
}
// the fragments k-step 0 of a tile starts from (the first tile of a workgroup; later tiles get them in place during the previous tile's last k-step)
template <int TH>
__device__ __forceinline__ void w3_win_prologue(const char* buf, int gbase, int ibase, W3Win& w) {
    constexpr int DXA = TH ? 2 : 0;
    w.g[0] = w3_win_g(buf, gbase, 0);
    w.xa[0] = w3_win_x<TH>(buf, ibase, 0, DXA);
    w.xb[0] = w3_win_x<TH>(buf, ibase, TH ? 2 : 0, 1);
    w.xa[1] = w3_win_x<TH>(buf, ibase, 1, DXA);
    if constexpr (TH == 0) w.xb[1] = w3_win_x<TH>(buf, ibase, 1, 1);
    w.xa[2] = w3_win_x<TH>(buf, ibase, 2, DXA);
}
// k-steps 0 .. 6 of a tile
// ABL (DASR_BENCH builds, WRONG results): bit 1 = no fragment reads inside the k-steps (the MFMAs reuse the first tile's fragments)
template <bool F16, int TH, int ABL = 0>
__device__ __forceinline__ void w3_win_steps(const char* buf, int gbase, int ibase, W3Win& w, f32x16 (&acc)[5], float& bsum, bool want_bias) {
    constexpr int DXA = TH ? 2 : 0;
    constexpr bool RD = !(ABL & 2);
#pragma unroll
    for (int r = 0; r < W3G::PH - 1; ++r) {
        if constexpr (RD) w.g[(r + 1) & 1] = w3_win_g(buf, gbase, r + 1);
        if constexpr (TH == 1 && RD) w.xb[(r + 1) & 1] = w3_win_x<TH>(buf, ibase, r + 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = mfma16<F16>(w.g[r & 1], w.xa[r % 3], acc[0]);                       // (0, DXA)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RD) w.xa[r % 3] = w3_win_x<TH>(buf, ibase, r + 3, DXA);        // -> (2, DXA) of k-step r + 1
        __builtin_amdgcn_sched_barrier(0);
        acc[3] = mfma16<F16>(w.g[r & 1], w.xb[r & 1], acc[3]);                       // th 0: (0, 1); th 1: (2, 1)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TH == 0 && RD) {
            w.xb[r & 1] = w3_win_x<TH>(buf, ibase, r + 2, 1);                        // -> (1, 1) of k-step r + 1
            __builtin_amdgcn_sched_barrier(0);
        }
        acc[1] = mfma16<F16>(w.g[r & 1], w.xa[(r + 1) % 3], acc[1]);                 // (1, DXA)
        if constexpr (TH == 0) {
            acc[4] = mfma16<F16>(w.g[r & 1], w.xb[(r + 1) & 1], acc[4]);             // (1, 1)
            if (want_bias) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += frag_f32<F16>(w.g[r & 1], j);
            }
        }
        acc[2] = mfma16<F16>(w.g[r & 1], w.xa[(r + 2) % 3], acc[2]);                 // (2, DXA)
        __builtin_amdgcn_sched_barrier(0);
    }
}
// k-step 7 of a tile: nothing is read from the tile's own image any more; every fragment register is re-filled IN PLACE, right behind its last MFMA, with
// what k-step 0 of the NEXT tile starts from (`nbuf`: landed -- the barrier in front of this k-step is the loaders' "tile t + 1 is in LDS"; behind the last tile
// of the workgroup the reads fetch a stale image and nobody uses them).  So a tile starts without a burst of prologue reads behind its barrier.
template <bool F16, int TH, int ABL = 0>
__device__ __forceinline__ void w3_win_last(const char* nbuf, int gbase, int ibase, W3Win& w, f32x16 (&acc)[5], float& bsum, bool want_bias) {
    constexpr int DXA = TH ? 2 : 0, r = W3G::PH - 1;
    constexpr bool RD = !(ABL & 2);
    static_assert(r == 7, "ring indices below are those of k-step 7");
    if constexpr (RD) w.g[0] = w3_win_g(nbuf, gbase, 0);
    if constexpr (TH == 1 && RD) w.xb[0] = w3_win_x<TH>(nbuf, ibase, 2, 1);
    __builtin_amdgcn_sched_barrier(0);
    acc[2] = mfma16<F16>(w.g[1], w.xa[0], acc[2]);                                   // (2, DXA): halo row 9
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (RD) w.xa[0] = w3_win_x<TH>(nbuf, ibase, 0, DXA);
    __builtin_amdgcn_sched_barrier(0);
    acc[0] = mfma16<F16>(w.g[1], w.xa[1], acc[0]);                                   // (0, DXA): halo row 7
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (RD) w.xa[1] = w3_win_x<TH>(nbuf, ibase, 1, DXA);
    __builtin_amdgcn_sched_barrier(0);
    acc[1] = mfma16<F16>(w.g[1], w.xa[2], acc[1]);                                   // (1, DXA): halo row 8
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (RD) w.xa[2] = w3_win_x<TH>(nbuf, ibase, 2, DXA);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TH == 0) {
        acc[4] = mfma16<F16>(w.g[1], w.xb[0], acc[4]);                               // (1, 1): halo row 8
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RD) w.xb[0] = w3_win_x<TH>(nbuf, ibase, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (want_bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum += frag_f32<F16>(w.g[1], j);
        }
        acc[3] = mfma16<F16>(w.g[1], w.xb[1], acc[3]);                               // (0, 1): halo row 7
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RD) w.xb[1] = w3_win_x<TH>(nbuf, ibase, 1, 1);
    } else {
        acc[3] = mfma16<F16>(w.g[1], w.xb[1], acc[3]);                               // (2, 1): halo row 9
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ABL (DASR_BENCH builds, WRONG results, timing only): bit 0 = the loaders request nothing after the first two tiles, bit 1 = no fragment reads inside the k-steps
template <bool F16, bool WIN, int ABL = 0>
__device__ __forceinline__ void wgrad3_ld_body(const dasr_wgrad_part* __restrict__ parts, int nparts, int nsplit_flags, float* __restrict__ ws) {
    using C = W3G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsplit = nsplit_flags & 0xffff;
    int part_id, split;
    w3_block_map(nsplit_flags, part_id, split);
    const dasr_wgrad_part P = parts[part_id];
    const int tiles_x = (P.Wout + C::PW - 1) / C::PW, tiles_y = (P.Hout + C::PH - 1) / C::PH;
    const int ntiles = tiles_x * tiles_y * P.N;
    const unsigned lds0 = (unsigned)(size_t)(DASR_LDS char*)smem;
    if (wave >= 12) {
        // ---- loader wave lw: quarter lw (32 pixels) of each of the 6 G planes, and the 6 sixths of X plane lw
        const int lw = wave - 12;
        const int HL = P.ups ? 2 * P.Hin : P.Hin, WL = P.ups ? 2 * P.Win : P.Win;
        const int half8 = (lane & 1) * 8;
        int pos_g, pos_x[6];
        {
            const int pix = (lw * 64 + lane) >> 1;
            pos_g = (pix >> 4) | ((pix & 15) << 16);
        }
        const bool x_plane_ok = (lw < P.in_planes) & (lw < 2 * P.n_ctiles);
#pragma unroll
        for (int sub = 0; sub < 6; ++sub) {
            const int pix = (sub * 64 + lane) >> 1, iy = pix / C::IW;
            pos_x[sub] = (((pix < C::IPIX) & x_plane_ok) ? iy : 0x7fff) | ((pix - iy * C::IW) << 16);
        }
        const int rel_x = lw * (int)P.in.cb_stride + half8;
        auto fill = [&](int tile, unsigned buf) {
            const W3GTile T = w3g_tile(P, tile, tiles_x, tiles_y);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int oy = T.oy0 + (pos_g & 0xffff), ox = T.ox0 + (pos_g >> 16);
                const bool ok = (oy < P.Hout) & (ox < P.Wout) & (k < P.g_planes);
                const unsigned off = ok ? (unsigned)((k * (int)P.g.cb_stride + half8 + (oy * P.Wout + ox) * 16) * 2) : OOB;
                lds_dma16(T.gb, buf + k * C::GPLANE + lw * 1024, off);
            }
#pragma unroll
            for (int sub = 0; sub < 6; ++sub) {
                const int gy = T.oy0 - P.pad + (pos_x[sub] & 0xffff), gx = T.ox0 - P.pad + (pos_x[sub] >> 16);
                const bool ok = (gy >= 0) & (gy < HL) & (gx >= 0) & (gx < WL);
                const int sy = P.ups ? gy >> 1 : gy, sx = P.ups ? gx >> 1 : gx;
                const unsigned off = ok ? (unsigned)((rel_x + (sy * P.Win + sx) * 16) * 2) : OOB;
                lds_dma16(T.ib, buf + C::G_BYTES + lw * C::IPLANE + sub * 1024, off);
            }
        };
        if constexpr (ABL & 4) {
            // (experiment, correct results) the loaders stage through REGISTERS: buffer_load_dwordx4 -> 12 x 4 registers -> ds_write_b128, two register sets =
            // two tiles of lookahead (tile t + 1 is written to LDS during tile t, its loads were issued during tile t - 2).  Asks whether what the LDS-DMA
            // costs next to the MFMAs (-11 % without it) is its write path into LDS or the bytes themselves.
            u32x4 st[2][12];
            auto load = [&](int tile, u32x4 (&r)[12]) {
                const W3GTile T = w3g_tile(P, tile, tiles_x, tiles_y);
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int oy = T.oy0 + (pos_g & 0xffff), ox = T.ox0 + (pos_g >> 16);
                    const bool ok = (oy < P.Hout) & (ox < P.Wout) & (k < P.g_planes);
                    const unsigned off = ok ? (unsigned)((k * (int)P.g.cb_stride + half8 + (oy * P.Wout + ox) * 16) * 2) : OOB;
                    r[k] = __builtin_amdgcn_raw_buffer_load_b128(T.gb, off, 0, 0);
                }
#pragma unroll
                for (int sub = 0; sub < 6; ++sub) {
                    const int gy = T.oy0 - P.pad + (pos_x[sub] & 0xffff), gx = T.ox0 - P.pad + (pos_x[sub] >> 16);
                    const bool ok = (gy >= 0) & (gy < HL) & (gx >= 0) & (gx < WL);
                    const int sy = P.ups ? gy >> 1 : gy, sx = P.ups ? gx >> 1 : gx;
                    const unsigned off = ok ? (unsigned)((rel_x + (sy * P.Win + sx) * 16) * 2) : OOB;
                    r[6 + sub] = __builtin_amdgcn_raw_buffer_load_b128(T.ib, off, 0, 0);
                }
            };
            auto commit = [&](int slot, const u32x4 (&r)[12]) {
                char* b = smem + slot * C::BUF_BYTES + lane * 16;
#pragma unroll
                for (int k = 0; k < 6; ++k) *(u32x4*)(b + k * C::GPLANE + lw * 1024) = r[k];
#pragma unroll
                for (int sub = 0; sub < 6; ++sub) *(u32x4*)(b + C::G_BYTES + lw * C::IPLANE + sub * 1024) = r[6 + sub];
            };
            if (split < ntiles) {
                load(split, st[0]);
                commit(0, st[0]);
            }
            if (split + nsplit < ntiles) load(split + nsplit, st[1]);
            if (split + 2 * nsplit < ntiles) load(split + 2 * nsplit, st[0]);
            asm volatile("s_waitcnt lgkmcnt(0)");
            __builtin_amdgcn_s_barrier();
            int cur = 0, tile = split;
            auto iter = [&](auto par) {   // iteration of tile `tile`: tile + nsplit (registers of set `par`) goes to LDS, tile + 3 nsplit is requested into the same set
                constexpr int S = decltype(par)::value;
                const int nxt = cur == 2 ? 0 : cur + 1;
                if (tile + nsplit < ntiles) commit(nxt, st[S]);
                if (tile + 3 * nsplit < ntiles) load(tile + 3 * nsplit, st[S]);
                asm volatile("s_waitcnt lgkmcnt(0)");
                __builtin_amdgcn_s_barrier();
                cur = nxt;
                tile += nsplit;
            };
            while (tile < ntiles) {
                iter(std::integral_constant<int, 1>{});
                if (tile < ntiles) iter(std::integral_constant<int, 0>{});
            }
            return;
        }
        // ring of three: during tile t the loaders request tile t + 2 (into the slot tile t - 1 was read from) and wait only for tile t + 1,
        // requested a whole tile earlier: the barrier that ends a tile never waits for a load in flight
        if (split < ntiles) fill(split, lds0);
        if (split + nsplit < ntiles) {
            fill(split + nsplit, lds0 + C::BUF_BYTES);
            asm volatile("s_waitcnt vmcnt(12)");
        } else {
            asm volatile("s_waitcnt vmcnt(0)");
        }
        __syncthreads();
        int cur = 0;
        for (int tile = split; tile < ntiles; tile += nsplit) {
            const int slot2 = cur == 0 ? 2 : cur - 1;
            const bool more = tile + 2 * nsplit < ntiles && !(ABL & 1);
            if (more) {
                fill(tile + 2 * nsplit, lds0 + slot2 * C::BUF_BYTES);
                asm volatile("s_waitcnt vmcnt(12)");   // tile t + 1 has landed (this loader's share), tile t + 2 may fly ...
            } else {
                asm volatile("s_waitcnt vmcnt(0)");
            }
            __syncthreads();                           // ... everybody's; and the compute waves are done with the current slot
            cur = cur == 2 ? 0 : cur + 1;
        }
        return;
    }
    // ---- compute wave: pair = wave % 6 -> (oc tile, cin tile); taps 0..4 (wave < 6) or 5..8
    const int pair = wave % 6, th = wave / 6;
    const int ot = pair >> 1, ct = pair & 1;
    const int n_ot = (P.g_planes + 1) >> 1;
    const bool active = ct < P.n_ctiles && ot < n_ot;
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    float bsum = 0.f;
    const int gg = lane >> 4, li = lane & 15;
    const int fplane = gg & 1, khalf = gg >> 1;
    const int gbase = (ot * 2 + fplane) * C::GPLANE + (8 * khalf + (li >> 2)) * 32 + (li & 3) * 8;
    const int ibase = C::G_BYTES + (ct * 2 + fplane) * C::IPLANE + (8 * khalf + (li >> 2)) * 32 + (li & 3) * 8;
    // taps of this wave: th = 0 -> 0..4, th = 1 -> 5..8 (the fifth slot repeats tap 8's address and its MFMA is skipped): one code path
    int tb[5];
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        const int t = th ? (a < 4 ? 5 + a : 8) : a;
        tb[a] = ibase + ((t / 3) * C::IW + (t % 3)) * 32;
    }
    const bool want_bias = P.want_bias && ct == 0 && th == 0;
    __syncthreads();   // tile 0 is in LDS
    int cur = 0;
    if constexpr (WIN) {
        // ONE barrier per tile, in front of its LAST k-step: for the loaders it is "tile t + 1 has landed, and nobody requests anything from tile t's image
        // any more" (k-step 7 reads registers only; its in-place reloads go to tile t + 1's image).  One loop per tap set: the wave-uniform choice is made
        // once; every wave, active or not, meets the same barriers.
        auto run = [&](auto th_c) {
            constexpr int TH = decltype(th_c)::value;
            W3Win w;
            if (active) w3_win_prologue<TH>(smem, gbase, ibase, w);
            for (int tile = split; tile < ntiles; tile += nsplit) {
                const char* buf = smem + cur * C::BUF_BYTES;
                cur = cur == 2 ? 0 : cur + 1;
                if (active) w3_win_steps<F16, TH, ABL>(buf, gbase, ibase, w, acc, bsum, want_bias);
                __syncthreads();
                if (active) w3_win_last<F16, TH, ABL>(smem + cur * C::BUF_BYTES, gbase, ibase, w, acc, bsum, want_bias);
            }
        };
        if (th == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, 1>{});
    } else
    for (int tile = split; tile < ntiles; tile += nsplit) {
        const char* buf = smem + cur * C::BUF_BYTES;
        cur = cur == 2 ? 0 : cur + 1;
        if (active) {
            bf16x8 a[2], b[2];
            a[0] = frag_tr(buf, gbase, gbase + 4 * 32);
            b[0] = frag_tr(buf, tb[0], tb[0] + 4 * 32);
#pragma unroll
            for (int i = 0; i < C::PH * 5; ++i) {
                const int r = i / 5, t = i - r * 5;
                if (i + 1 < C::PH * 5) {
                    const int r1 = (i + 1) / 5, t1 = (i + 1) - r1 * 5;
                    const int o1 = tb[t1] + r1 * C::IW * 32;
                    b[(i + 1) & 1] = frag_tr(buf, o1, o1 + 4 * 32);
                    if (t1 == 0) {
                        const int g1 = gbase + r1 * C::PW * 32;
                        a[r1 & 1] = frag_tr(buf, g1, g1 + 4 * 32);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (t < 4 || th == 0) acc[t] = mfma16<F16>(a[r & 1], b[i & 1], acc[t]);
                if (t == 2 && want_bias) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum += frag_f32<F16>(a[r & 1], j);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    if (active) {
        float* w = ws + P.ws_off + (size_t)split * 9 * 3 * 2048 + (size_t)ot * 2048;
        const int cin = ct * 32 + (lane & 31), h = lane >> 5;
        const int t0 = th * 5, na = th ? 4 : 5;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            if (t < na) {
                const int tap = WIN ? (th ? (t == 0 ? 2 : t == 1 ? 5 : t == 2 ? 8 : 7) : (t == 0 ? 0 : t == 1 ? 3 : t == 2 ? 6 : t == 3 ? 1 : 4)) : t0 + t;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int oc = (j & 3) + 8 * (j >> 2) + 4 * h;
                    w[(size_t)tap * 3 * 2048 + oc * 64 + cin] = acc[t][j];
                }
            }
        }
        if (want_bias) {
            const float tot = bsum + __shfl_xor(bsum, 32, 64);
            if (lane < 32) ws[P.ws_bias_off + (size_t)split * 96 + ot * 32 + lane] = tot;
        }
    }
}

template <bool F16>
__global__ __launch_bounds__(1024, 1) void wgrad3_ld_kernel(const dasr_wgrad_part* __restrict__ parts, int nparts, int nsplit_flags, float* __restrict__ ws) {
    wgrad3_ld_body<F16, true>(parts, nparts, nsplit_flags, ws);
}
#ifdef DASR_BENCH   // the round-3 compute form (six fragment reads per k-step), kept for the A/B of profiles/r06_wgrad_window.txt
template <bool F16>
__global__ __launch_bounds__(1024, 1) void wgrad3_ld6_kernel(const dasr_wgrad_part* __restrict__ parts, int nparts, int nsplit_flags, float* __restrict__ ws) {
    wgrad3_ld_body<F16, false>(parts, nparts, nsplit_flags, ws);
}
template <int ABL>
__global__ __launch_bounds__(1024, 1) void wgrad3_ld_abl_kernel(const dasr_wgrad_part* __restrict__ parts, int nparts, int nsplit_flags, float* __restrict__ ws) {
    wgrad3_ld_body<false, true, ABL>(parts, nparts, nsplit_flags, ws);
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// wgrad v4 (round 3): the same parts (one 64-channel input block x up to three 32-oc tiles x 9 taps = 54 accumulator tiles) and the same
// workspace layout as wgrad3, re-blocked around what the round-3 ablation of wgrad3 measured (profiles/r03e_wgrad3_ablation.txt: MFMA + fragment
// reads alone 73 % of the launch, the register-staged global prefetch 21 %, the 12-wave LDS commit phase 4 %; 2.4 transposed LDS reads per MFMA):
//  * FOUR waves, one per SIMD, wave = (cin tile ct, role).  A wave owns all 9 taps of its own oc tile (role 0: tile 0, role 1: tile 2) and a share
//    of the taps of oc tile 1 (role 0: taps 0-3, role 1: taps 4-8): 13 / 14 accumulators (<= 256 AGPRs), 104 / 112 MFMAs per 8 x 16-pixel tile.
//    The code of the two roles is IDENTICAL: the own tile reads its X fragments from a sliding register window (three halo rows x three column
//    shifts, 3 new fragments per k-step); the shared tile's up to five X fragments are read at per-role LDS offsets held in scalar registers.
//    0.7 fragment reads per MFMA (wgrad3: 1.2), one fragment set ahead of the MFMAs.
//  * global -> LDS by LDS-DMA into the other buffer while the current tile is multiplied (no staging registers, no ds_write, ONE barrier per
//    tile); the twelve 1-KiB DMA instructions of a wave are issued one or two per k-step between the MFMAs.
// ---------------------------------------------------------------------------------------------------------------
struct W4 {
    static constexpr int PH = 8, PW = 16, IH = 10, IW = 18, GPIX = PH * PW, IPIX = IH * IW;
    static constexpr int GPLANE = GPIX * 32 + 128;           // 4 DMA instructions per plane
    static constexpr int IPLANE = 6 * 1024 + 128;            // 6 DMA instructions per plane (180 pixels = 5.6 KiB), = 128 (mod 256)
    static constexpr int G_BYTES = 6 * GPLANE, I_BYTES = 4 * IPLANE;
    static constexpr int BUF_BYTES = G_BYTES + I_BYTES;
    static constexpr int LDS_BYTES = 3 * BUF_BYTES;          // ring of three tiles: 151 KB of the CU's 160
    static constexpr int NT = 256, NDMA = 12;                // 48 DMA instructions per tile, 12 per wave
};

// ABL (instantiated != 0 only under -DDASR_BENCH, WRONG results): bit 0 no DMA after the prologue, bit 1 no fragment requests inside the k-steps,
// bit 2 no MFMA, bit 3 no barrier / DMA wait per tile
template <bool F16, int ABL = 0>
__global__ __launch_bounds__(256, 1) void wgrad4_kernel(const dasr_wgrad_part* __restrict__ parts, int nparts, int nsplit_flags, float* __restrict__ ws) {
    using C = W4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsplit = nsplit_flags & 0xffff;
    int part_id, split;
    w3_block_map(nsplit_flags, part_id, split);
    const dasr_wgrad_part P = parts[part_id];
    const int ct = wave & 1, role = wave >> 1;
    const int n_ot = (P.g_planes + 1) >> 1;
    const int otA = role ? 2 : 0;
    const bool act_ct = ct < P.n_ctiles;
    const bool actA = act_ct && otA < n_ot, actB = act_ct && 1 < n_ot;
    const int tiles_x = (P.Wout + C::PW - 1) / C::PW, tiles_y = (P.Hout + C::PH - 1) / C::PH;
    const int ntiles = tiles_x * tiles_y * P.N;
    const int HL = P.ups ? 2 * P.Hin : P.Hin, WL = P.ups ? 2 * P.Win : P.Win;
    const int gg = lane >> 4, li = lane & 15;
    const int fplane = gg & 1, khalf = gg >> 1;
    const int frag_off = (8 * khalf + (li >> 2)) * 32 + (li & 3) * 8;
    const int gbaseA = (otA * 2 + fplane) * C::GPLANE + frag_off, gbaseB = (2 + fplane) * C::GPLANE + frag_off;
    const int ibase = C::G_BYTES + (ct * 2 + fplane) * C::IPLANE + frag_off;
    const bool want_bias = P.want_bias && ct == 0;
    // this wave's twelve DMA instructions per tile (1 KiB each): k < 6: quarter `wave` (32 pixels) of G plane k; k >= 6: sixth k - 6 (32 pixels) of X
    // plane `wave`.  Per-lane geometry: ONE pixel position for all G pieces, six for the X pieces (packed dy | dx << 16, dy = 0x7fff: never valid);
    // the plane offsets are wave-uniform.
    const int half8 = (lane & 1) * 8;
    int pos_g, pos_x[6];
    {
        const int pix = (wave * 64 + lane) >> 1;
        pos_g = (pix >> 4) | ((pix & 15) << 16);
    }
    const bool x_plane_ok = (wave < P.in_planes) & (wave < 2 * P.n_ctiles);
#pragma unroll
    for (int sub = 0; sub < 6; ++sub) {
        const int pix = (sub * 64 + lane) >> 1, iy = pix / C::IW;
        pos_x[sub] = (((pix < C::IPIX) & x_plane_ok) ? iy : 0x7fff) | ((pix - iy * C::IW) << 16);
    }
    const int rel_x = wave * (int)P.in.cb_stride + half8;
    auto dma = [&](int k, const W3GTile& T, unsigned buf) {   // buf: LDS byte address of the target buffer; k is a compile-time constant at every call site
        if (k < 6) {
            const int oy = T.oy0 + (pos_g & 0xffff), ox = T.ox0 + (pos_g >> 16);
            const bool ok = (oy < P.Hout) & (ox < P.Wout) & (k < P.g_planes);
            const unsigned off = ok ? (unsigned)((k * (int)P.g.cb_stride + half8 + (oy * P.Wout + ox) * 16) * 2) : OOB;
            lds_dma16(T.gb, buf + k * C::GPLANE + wave * 1024, off);
        } else {
            const int sub = k - 6;
            const int gy = T.oy0 - P.pad + (pos_x[sub] & 0xffff), gx = T.ox0 - P.pad + (pos_x[sub] >> 16);
            const bool ok = (gy >= 0) & (gy < HL) & (gx >= 0) & (gx < WL);
            const int sy = P.ups ? gy >> 1 : gy, sx = P.ups ? gx >> 1 : gx;
            const unsigned off = ok ? (unsigned)((rel_x + (sy * P.Win + sx) * 16) * 2) : OOB;
            lds_dma16(T.ib, buf + C::G_BYTES + wave * C::IPLANE + sub * 1024, off);
        }
    };
    const unsigned lds0 = (unsigned)(size_t)(DASR_LDS char*)smem;   // LDS byte address of the dynamic shared memory
    // ring of THREE tile buffers: the pieces of tile t + 2 are requested in the first k-steps of tile t, so the wait that ends tile t (for the
    // pieces of tile t + 1, requested a whole tile earlier) never sees a load in flight
    if (split < ntiles) {
        const W3GTile T0 = w3g_tile(P, split, tiles_x, tiles_y);
#pragma unroll
        for (int k = 0; k < C::NDMA; ++k) dma(k, T0, lds0);
    }
    if (split + nsplit < ntiles) {
        const W3GTile T1 = w3g_tile(P, split + nsplit, tiles_x, tiles_y);
#pragma unroll
        for (int k = 0; k < C::NDMA; ++k) dma(k, T1, lds0 + C::BUF_BYTES);
        asm volatile("s_waitcnt vmcnt(12)");   // tile 0 has landed (this wave's pieces), tile 1 may still fly
    } else {
        asm volatile("s_waitcnt vmcnt(0)");
    }
    __syncthreads();

    // The two roles run two specialised copies of the whole tile loop + epilogue (a wave executes one): ONE wave per SIMD means nothing hides a
    // stall and every instruction beside an MFMA costs issue time (~5 fit into an MFMA's 32 cycles), so the stream is laid out by hand and kept
    // minimal: per k-step 13 / 14 MFMAs, 5 fragment requests (3 X fragments of halo row r + 2, the next row's two G fragments: 10 transposed
    // LDS reads), all addresses = one VGPR per buffer + immediate offsets.  Role 0: shared-tile taps 0-3 = window (row r: kx 0-2), (row r+1: kx 0);
    // role 1: taps 4-8 = (row r+1: kx 1, 2), (row r+2: kx 0-2).
    auto body = [&](auto role_c) {
        constexpr int ROLE = decltype(role_c)::value;
        constexpr int NB = ROLE ? 5 : 4, TB0 = ROLE ? 4 : 0;
        constexpr int NM = 9 + NB;
        f32x16 accA[9], accB[NB];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) accA[t][j] = 0.f;
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) accB[t][j] = 0.f;
        float bsumA = 0.f, bsumB = 0.f;
        int cur = 0;   // ring slot of the current tile
        for (int tile = split; tile < ntiles; tile += nsplit) {
            const char* gA = smem + cur * C::BUF_BYTES + gbaseA;   // per-lane base addresses inside the current buffer: everything else is an immediate
            const char* gB = smem + cur * C::BUF_BYTES + gbaseB;
            const char* xb = smem + cur * C::BUF_BYTES + ibase;
            const int slot2 = cur == 0 ? 2 : cur - 1;   // (cur + 2) % 3: the slot tile t - 1 was read from
            const unsigned nbuf = lds0 + slot2 * C::BUF_BYTES;
            const bool more = tile + 2 * nsplit < ntiles;   // a tile t + 2 exists: request it now
            const W3GTile T = w3g_tile(P, more ? tile + 2 * nsplit : tile, tiles_x, tiles_y);
            if (act_ct) {
                bf16x8 win[3][3], ga[2], gb[2];
#pragma unroll
                for (int row = 0; row < 2; ++row)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) win[row][kx] = frag_tr(xb, (row * C::IW + kx) * 32, (row * C::IW + kx) * 32 + 4 * 32);
                ga[0] = frag_tr(gA, 0, 4 * 32);
                gb[0] = frag_tr(gB, 0, 4 * 32);
                if constexpr (ABL & 2) {
                    ga[1] = ga[0];
                    gb[1] = gb[0];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) win[2][kx] = win[0][kx];
                }
#pragma unroll
                for (int r = 0; r < C::PH; ++r) {
#pragma unroll
                    for (int i = 0; i < NM; ++i) {
                        // MFMA i of the k-step.  Order: everything that needs only halo rows r, r + 1 (in registers) first, ky = 2 last.
                        // role 0: shared taps 0-3, own taps 0-5, own taps 6-8; role 1: shared taps 4, 5, own 0-5, own 6-8, shared 6-8
                        int own = -1, sh = -1;
                        if (ROLE == 0) {
                            if (i < 4) sh = i;
                            else own = i - 4;
                        } else {
                            if (i < 2) sh = 4 + i;
                            else if (i < 11) own = i - 2;
                            else sh = 6 + (i - 11);
                        }
                        if constexpr (!(ABL & 4)) {
                            if (own >= 0) accA[own] = mfma16<F16>(ga[r & 1], win[(r + own / 3) % 3][own % 3], accA[own]);
                            else accB[sh - TB0] = mfma16<F16>(gb[r & 1], win[(r + sh / 3) % 3][sh % 3], accB[sh - TB0]);
                        } else {
                            asm volatile("" ::"v"(ga[r & 1]), "v"(gb[r & 1]), "v"(win[(r + i / 3) % 3][i % 3]));
                        }
                        // at most one request behind each MFMA
                        if (i < 3) {            // halo row r + 2 (first used >= 6 MFMAs later)
                            if constexpr (!(ABL & 2)) win[(r + 2) % 3][i] = frag_tr(xb, ((r + 2) * C::IW + i) * 32, ((r + 2) * C::IW + i) * 32 + 4 * 32);
                        } else if (i == 3) {
                            if (r + 1 < C::PH && !(ABL & 2)) ga[(r + 1) & 1] = frag_tr(gA, (r + 1) * C::PW * 32, (r + 1) * C::PW * 32 + 4 * 32);
                        } else if (i == 4) {
                            if (r + 1 < C::PH && !(ABL & 2)) gb[(r + 1) & 1] = frag_tr(gB, (r + 1) * C::PW * 32, (r + 1) * C::PW * 32 + 4 * 32);
                        } else if (i < 9) {     // this wave's DMA pieces of tile t + 2: four per k-step in the first three
                            if (more && r < 3 && !(ABL & 1)) dma(4 * r + (i - 5), T, nbuf);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (want_bias) {   // sum over pixels of G from the 16-bit fragments: own tile; the shared tile by role 0 only
#pragma unroll
                        for (int j = 0; j < 8; ++j) bsumA += frag_f32<F16>(ga[r & 1], j);
                        if (ROLE == 0) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) bsumB += frag_f32<F16>(gb[r & 1], j);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else if (more && !(ABL & 1)) {
#pragma unroll
                for (int k = 0; k < C::NDMA; ++k) dma(k, T, nbuf);
            }
            // this wave's pieces of tile t + 1 have landed (the twelve of tile t + 2, if requested, may still fly) ...
            if constexpr (!(ABL & 8)) {
                if (more && !(ABL & 1)) asm volatile("s_waitcnt vmcnt(12)");
                else asm volatile("s_waitcnt vmcnt(0)");
                __syncthreads();                      // ... everybody's; and everybody is done reading the current buffer
            }
            cur = cur == 2 ? 0 : cur + 1;
        }
        // ---- partial sums of this pixel split: ws[part][split][tap 9][ot 3][oc 32][cin 64]; bias [split][96]
        float* w0 = ws + P.ws_off + (size_t)split * 9 * 3 * 2048;
        const int cin = ct * 32 + (lane & 31), h = lane >> 5;
        if (actA) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float* w = w0 + (size_t)t * 3 * 2048 + (size_t)otA * 2048;
#pragma unroll
                for (int j = 0; j < 16; ++j) w[((j & 3) + 8 * (j >> 2) + 4 * h) * 64 + cin] = accA[t][j];
            }
        }
        if (actB) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                float* w = w0 + (size_t)(TB0 + i) * 3 * 2048 + 2048;
#pragma unroll
                for (int j = 0; j < 16; ++j) w[((j & 3) + 8 * (j >> 2) + 4 * h) * 64 + cin] = accB[i][j];
            }
        }
        if (want_bias) {
            const float tA = bsumA + __shfl_xor(bsumA, 32, 64), tB = bsumB + __shfl_xor(bsumB, 32, 64);
            if (lane < 32) {
                if (actA) ws[P.ws_bias_off + (size_t)split * 96 + otA * 32 + lane] = tA;
                if (ROLE == 0 && actB) ws[P.ws_bias_off + (size_t)split * 96 + 32 + lane] = tB;
            }
        }
    };
    if (role == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
}

// deterministic split reduction.  One workgroup = one output channel x 16 input channels of one part; thread (split lane sl = t >> 4, cin =
// t & 15) sums splits sl, sl + 16, ... of every tap (two accumulators, fixed order), the 16 lane sums meet in LDS and are added in a fixed
// tree, and the reference layout [cout][cin][kh][kw] is written as ONE contiguous run of 16 * ntaps floats per workgroup (the first version
// stored 4 bytes per lane 36 bytes apart: 8x write amplification, ~1 TB/s, 8 % of the DSN iteration).  128 workgroups per part: enough
// parallelism both for the many-parts / 16-splits launches of the RRDB trunk and the 2-parts / 128-splits launches of the DSN generator.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const dasr_wgrad_reduce_part* __restrict__ parts, int nparts,
                                                           const float* __restrict__ ws, float* __restrict__ grad, float scale) {
    __shared__ float red[16 * 16 * 17];
    const dasr_wgrad_reduce_part P = parts[blockIdx.y];
    // gridDim.x == 32 (the launcher's `few_splits`): every part takes the few-splits path below, which needs ONE workgroup per output channel -- the three
    // of four workgroups that only returned cost more than the work (7680 workgroups per launch of the grouped trunk gradients: 55 us for 18 MB, dispatch-bound)
    const bool per_oc = gridDim.x == 32;
    const int oc = per_oc ? blockIdx.x : blockIdx.x >> 2, cg = per_oc ? 0 : blockIdx.x & 3, goc = P.oc0 + oc;
    if (goc >= P.cout) return;   // uniform per block
    const int per = P.ntaps * 32 * 64;
    const long long sstride = P.split_stride > 0 ? P.split_stride : per, tstride = P.tap_stride > 0 ? P.tap_stride : 2048;
    const long long bstride = P.bias_stride > 0 ? P.bias_stride : 32;
    const int NT = P.ntaps_total > 0 ? P.ntaps_total : P.ntaps;
    const int ntaps = min(P.ntaps, NT - P.tap0);                        // taps of this part that exist in the kernel window
    const int n_c = min(min(32 * P.n_ctiles, P.cin - P.c0), 64) - cg * 16;   // valid input channels of this 16-channel group (may be <= 0)
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    if (P.nsplit <= 4 && ntaps <= 16) {
        // few splits (the grouped dense-block launches of the RRDB trunk: 60 parts x 4 pixel splits): the 16 split lanes would be three quarters
        // idle and a workgroup would move 2.3 KB.  Instead ONE workgroup per output channel covers all four 16-channel groups: lane group
        // sl = 4 * (channel group) + split, nine independent loads per thread, one contiguous run of 64 * ntaps floats out.  Same summation
        // order as the general path ((s0 + s1) + (s2 + s3)): bit-identical results.
        if (cg != 0) return;   // (uniform per block)
        const int n_all = min(min(32 * P.n_ctiles, P.cin - P.c0), 64);
        const int cgi = sl >> 2, sp = sl & 3, c = cgi * 16 + cl;
        if (c < n_all && sp < P.nsplit) {
            const float* src = ws + P.ws_off + (size_t)sp * sstride + oc * 64 + c;
            float q[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) q[t] = t < ntaps ? src[(long long)t * tstride] : 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (t < ntaps) red[(sl * 16 + cl) * 17 + t] = q[t];
        } else if (c < n_all) {
            for (int t = 0; t < ntaps; ++t) red[(sl * 16 + cl) * 17 + t] = 0.f;
        }
        __syncthreads();
        float* dst = grad + P.dst_w_off + ((long long)goc * P.cin + P.c0) * NT + P.tap0;
        const int nvalid = n_all * ntaps;
        for (int o = threadIdx.x; o < nvalid; o += 256) {
            const int cc = o / ntaps, tap = o - cc * ntaps;
            const float* r = red + (((cc >> 4) * 4) * 16 + (cc & 15)) * 17 + tap;
            const float tot = (r[0] + r[16 * 17]) + (r[2 * 16 * 17] + r[3 * 16 * 17]);
            dst[ntaps == NT ? (long long)o : (long long)cc * NT + tap] = tot * scale;
        }
        if (P.dst_b_off >= 0 && threadIdx.x < 64) {
            float b = threadIdx.x < (P.bias_nsplit > 0 ? P.bias_nsplit : P.nsplit) ? ws[P.ws_bias_off + (size_t)threadIdx.x * bstride + oc] : 0.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) b += __shfl_xor(b, o, 64);
            if (threadIdx.x == 0) grad[P.dst_b_off + goc] = b * scale;
        }
        return;
    }
    if (n_c > 0) {
        const float* src = ws + P.ws_off + oc * 64 + cg * 16 + cl;
        for (int tap0 = 0; tap0 < ntaps; tap0 += 3) {      // three taps x eight splits = 24 independent loads in flight per thread: the
            float a[3] = {0.f, 0.f, 0.f};                    // few-parts / many-splits launches (DSN: 2 parts x 128 splits) are latency-bound
            if (cl < n_c) {
                for (int sp = sl; sp < P.nsplit; sp += 16 * 8) {
                    float q[3][8];
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            q[t][k] = (tap0 + t < ntaps && sp + 16 * k < P.nsplit) ? src[(size_t)(sp + 16 * k) * sstride + (long long)(tap0 + t) * tstride] : 0.f;
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int k = 0; k < 8; ++k) a[t] += q[t][k];   // ascending split order, whatever the unrolling
                }
            }
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (tap0 + t < ntaps) red[(sl * 16 + cl) * 17 + tap0 + t] = a[t];
        }
    }
    __syncthreads();
    if (n_c > 0) {
        float* dst = grad + P.dst_w_off + ((long long)goc * P.cin + P.c0 + cg * 16) * NT + P.tap0;
        const int nvalid = (n_c < 16 ? n_c : 16) * ntaps;
        for (int o = threadIdx.x; o < nvalid; o += 256) {
            const int c = o / ntaps, tap = o - c * ntaps;
            const float* r = red + c * 17 + tap;
            float q[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) q[k] = r[k * 16 * 17];
            const float tot = (((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]))) +
                              (((q[8] + q[9]) + (q[10] + q[11])) + ((q[12] + q[13]) + (q[14] + q[15])));
            // the whole kernel window: [cin][tap] is contiguous in the destination; a tap range of a larger window (5x5 in parts of 10 taps): runs
            dst[ntaps == NT ? (long long)o : (long long)c * NT + tap] = tot * scale;
        }
    }
    if (cg == 0 && P.dst_b_off >= 0 && threadIdx.x < 64) {   // bias of this output channel: lane l sums splits l, l + 64, ...; fixed xor tree
        float b = 0.f;
        const int nb_ = P.bias_nsplit > 0 ? P.bias_nsplit : P.nsplit;
        for (int sp = threadIdx.x; sp < nb_; sp += 64) b += ws[P.ws_bias_off + (size_t)sp * bstride + oc];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) b += __shfl_xor(b, o, 64);
        if (threadIdx.x == 0) grad[P.dst_b_off + goc] = b * scale;
    }
}

__global__ void probe_tr16_kernel(int* result) {
    __shared__ __attribute__((aligned(16))) short lds[4 * 64];
    const int l = threadIdx.x;
    // 4 groups, each a 4 x 16 row-major matrix M[row][col] = 1000*group + 16*row + col
    for (int i = l; i < 256; i += 64) lds[i] = (short)(1000 * (i >> 6) + (i & 63));
    __syncthreads();
    const int gg = l >> 4, li = l & 15;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((DASR_LDS s16x4*)((char*)lds + gg * 128 + (li >> 2) * 32 + (li & 3) * 8));
    int ok = 1;
    for (int j = 0; j < 4; ++j) ok &= (v[j] == (short)(1000 * gg + 16 * j + li));
    const unsigned long long m = __ballot(ok);
    if (l == 0) *result = (m == ~0ull) ? 1 : 0;
}

template <int KH, int STRIDE, bool USE_TR, bool F32, bool F16 = false>
int launch_wgrad(const dasr_wgrad_part* parts, int nparts, int nsplit, float* ws, hipStream_t s) {
    using C = WCfg<KH, STRIDE>;
    auto kfn = wgrad_kernel<KH, STRIDE, USE_TR, F32, F16>;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES + 16));
        attr_set = true;
    }
    DASR_LAUNCH_TAG(__PRETTY_FUNCTION__, kfn, dim3(nparts * nsplit), dim3(256), C::LDS_BYTES + 16, s, parts, nparts, nsplit, ws);
    return (int)hipGetLastError();
}

extern int g_wgrad3_stagger, g_wgrad3_abl;
template <bool USE_TR, bool F32, bool F16 = false, int ABL = 0>
int launch_wgrad3(const dasr_wgrad_part* parts, int nparts, int nsplit, float* ws, hipStream_t s) {
    auto kfn = wgrad3_kernel<USE_TR, F32, F16, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, W3::LDS_BYTES + 16));
        attr_set = true;
    }
    DASR_LAUNCH_TAG(__PRETTY_FUNCTION__, kfn, dim3(nparts * (nsplit & 0xffff)), dim3(W3::NT), W3::LDS_BYTES + 16, s, parts, nparts, nsplit | (g_wgrad3_stagger << 24), ws);
    return (int)hipGetLastError();
}

#ifdef DASR_BENCH
int launch_wgrad3_glds(const dasr_wgrad_part* parts, int nparts, int nsplit, float* ws, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)wgrad3_glds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W3G::LDS_BYTES));
        attr_set = true;
    }
    DASR_LAUNCH(wgrad3_glds_kernel, dim3(nparts * (nsplit & 0xffff)), dim3(W3G::NT), W3G::LDS_BYTES, s, parts, nparts, nsplit, ws);
    return (int)hipGetLastError();
}
#endif

template <bool F16, int ABL = 0>
int launch_wgrad4(const dasr_wgrad_part* parts, int nparts, int nsplit, float* ws, hipStream_t s) {
    auto kfn = wgrad4_kernel<F16, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, W4::LDS_BYTES));
        attr_set = true;
    }
    DASR_LAUNCH_TAG(__PRETTY_FUNCTION__, kfn, dim3(nparts * (nsplit & 0xffff)), dim3(W4::NT), W4::LDS_BYTES, s, parts, nparts, nsplit, ws);
    return (int)hipGetLastError();
}

template <bool F16>
int launch_wgrad3_ld(const dasr_wgrad_part* parts, int nparts, int nsplit, float* ws, hipStream_t s) {
    auto kfn = wgrad3_ld_kernel<F16>;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, W3L::LDS_BYTES));
        attr_set = true;
    }
    DASR_LAUNCH_TAG(__PRETTY_FUNCTION__, kfn, dim3(nparts * (nsplit & 0xffff)), dim3(W3L::NT), W3L::LDS_BYTES, s, parts, nparts, nsplit, ws);
    return (int)hipGetLastError();
}
#ifdef DASR_BENCH
template <int ABL>
int launch_wgrad3_ld_abl(const dasr_wgrad_part* parts, int nparts, int nsplit, float* ws, hipStream_t s) {
    auto kfn = wgrad3_ld_abl_kernel<ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, W3L::LDS_BYTES));
        attr_set = true;
    }
    DASR_LAUNCH_TAG(__PRETTY_FUNCTION__, kfn, dim3(nparts * (nsplit & 0xffff)), dim3(W3L::NT), W3L::LDS_BYTES, s, parts, nparts, nsplit, ws);
    return (int)hipGetLastError();
}
template <bool F16>
int launch_wgrad3_ld6(const dasr_wgrad_part* parts, int nparts, int nsplit, float* ws, hipStream_t s) {
    auto kfn = wgrad3_ld6_kernel<F16>;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, W3L::LDS_BYTES));
        attr_set = true;
    }
    DASR_LAUNCH_TAG(__PRETTY_FUNCTION__, kfn, dim3(nparts * (nsplit & 0xffff)), dim3(W3L::NT), W3L::LDS_BYTES, s, parts, nparts, nsplit, ws);
    return (int)hipGetLastError();
}
#endif

int g_use_tr = -1;  // -1 unknown, 0 gather fallback, 1 transpose reads
int g_wgrad3_ld = 1;  // 3x3 stride-1 weight gradients on 16-bit tensors: wgrad3_ld_kernel (12 compute + 4 loader waves); 0 = wgrad3_kernel
// 3x3 stride-1 weight gradients on 16-bit tensors: 0 = wgrad3_kernel (default), 1 = wgrad4_kernel (4 waves, LDS-DMA, register window).
// Measured in round 3 (profiles/r03_wgrad_ablation.txt): wgrad4 halves the LDS fragment reads and removes the commit phase, but with ONE wave per
// SIMD every LDS-DMA instruction that waits for the memory pipe also stops that SIMD's MFMAs: 0.79 PFLOP/s against wgrad3's 0.96 - 1.01 (three
// waves per SIMD cover each other).  Both kernels move 110 B of L2 -> LDS traffic per MFMA; the chip sustains ~9 TB/s of it, which costs 0.95 ms
// per grouped launch next to 1.3 - 1.6 ms of MFMA time: the weight gradient is bound by that SUM, not by either term.
int g_wgrad4 = 0;
int g_wgrad3_stagger = 1;
int g_wgrad3_abl = 0;   // DASR_BENCH builds: ablation bits of wgrad3_kernel (dasr_wgrad_set_mode bits 3-6); ignored by the product build
int g_wgrad3_ld_abl = 0;   // DASR_BENCH builds: ablation bits of wgrad3_ld_kernel (dasr_wgrad_set_mode bits 10-11)
int g_wgrad3_ld6 = 0;   // DASR_BENCH builds: wgrad3_ld6_kernel (the round-3 compute form of the loader-wave kernel) instead of wgrad3_ld_kernel (A/B)
int g_wgrad3_glds = 0;  // LDS-DMA wgrad3: faster alone (490 vs 470 TFLOP/s) but its 101 KB of LDS keeps the other sub-batch stream off the CU: -2.5 % on the step

}  // namespace

extern "C" int dasr_probe_tr16(void* stream) {
    int* d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(int)));
    DASR_LAUNCH(probe_tr16_kernel, dim3(1), dim3(64), 0, as_stream(stream), d);
    int h = -1;
    hipError_t e = hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, as_stream(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
    (void)hipFree(d);
    if (e != hipSuccess) return -(int)e;
    g_use_tr = h;
    return h;
}

#ifdef DASR_TRACE
extern "C" int dasr_debug_set_wtrace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wtrace), &buf, sizeof(buf)); }
#endif

extern "C" int dasr_wgrad_set_mode(int use_tr) {
#ifdef DASR_BENCH
    g_wgrad3_stagger = (use_tr & 4) ? 0 : 1;  // bit 2: all waves request the next tile before computing (A/B)
    g_wgrad3_abl = (use_tr >> 3) & 0xf;
    g_wgrad3_glds = (use_tr & 2) ? 1 : 0;  // bit 1: LDS-DMA wgrad3 instead of the register-staged one (A/B)
    g_wgrad4 = (use_tr & 128) ? 1 : 0;     // bit 7: wgrad4_kernel instead of wgrad3_kernel (A/B)
    g_wgrad3_ld = (use_tr & 256) ? 0 : 1;  // bit 8: the register-staged 12-wave wgrad3_kernel instead of the loader-wave kernel (A/B)
    g_wgrad3_ld_abl = (use_tr >> 10) & 7;  // bits 10-12 (4 = register-staged loaders, correct results); 1 / 2: wgrad3_ld_kernel without its DMA (after two tiles) / without its fragment reads (WRONG results)
    g_wgrad3_ld6 = (use_tr & 512) ? 1 : 0; // bit 9: the loader-wave kernel with six fragment reads per k-step (round 3) instead of the register window (A/B)
#else
    if (use_tr & ~1) return DASR_EINVAL;   // the kernel-selection bits exist in libdasr_hip_ablate.so only
#endif
    g_use_tr = use_tr & 1;
    return 0;
}

namespace {
template <int KH, int STRIDE>
int dispatch_wgrad(const dasr_wgrad_part* parts, int nparts, int nsplit, bool tr, int f32, float* ws, hipStream_t s) {
    if (f32 & 2) {   // f32 tensors rounded to f16 (pre-scaled gradient), f16 MFMA
        if (tr) return launch_wgrad<KH, STRIDE, true, true, true>(parts, nparts, nsplit, ws, s);
        return launch_wgrad<KH, STRIDE, false, true, true>(parts, nparts, nsplit, ws, s);
    }
    if (tr) return f32 ? launch_wgrad<KH, STRIDE, true, true>(parts, nparts, nsplit, ws, s) : launch_wgrad<KH, STRIDE, true, false>(parts, nparts, nsplit, ws, s);
    return f32 ? launch_wgrad<KH, STRIDE, false, true>(parts, nparts, nsplit, ws, s) : launch_wgrad<KH, STRIDE, false, false>(parts, nparts, nsplit, ws, s);
}
}  // namespace

extern "C" int dasr_wgrad(const dasr_wgrad_part* parts_dev, int32_t nparts, int32_t nsplit, int32_t kh, int32_t stride, int32_t f32,
                          float* ws, void* stream) {
    hipStream_t s = as_stream(stream);
    if (nparts <= 0 || nsplit <= 0) return DASR_EINVAL;
    if (kh != 33 && (nsplit >> 16)) return DASR_EINVAL;   // the unit placement bits exist for the 12-wave 3x3 kernel only
    if (kh == 33) {
        const int ns = nsplit & 0xffff, ppu = (nsplit >> 16) & 0xff;
        if (ns <= 0 || (nsplit >> 24) || (ppu && ((nparts % ppu) || (((long long)(nparts / ppu) * ns) & 7)))) return DASR_EINVAL;
    }
    if (g_use_tr < 0) return DASR_EINVAL;  // dasr_probe_tr16 must run once per process (outside graph capture)
    const bool tr = g_use_tr == 1;
    if (kh == 33) {  // 3x3 stride 1 on 16-bit tensors: one part = 64 input channels x up to three 32-oc tiles; wgrad3_ld_kernel (12 compute + 4 loader waves)
#ifdef DASR_BENCH   // libdasr_hip_ablate.so: the measured alternatives of rounds 2-3 (wgrad3_kernel, wgrad3_glds_kernel, wgrad4_kernel) and their ablations
        if (tr && g_wgrad4 && f32 == 0) switch (g_wgrad3_abl) {
            case 1: return launch_wgrad4<false, 1>(parts_dev, nparts, nsplit, ws, s);
            case 2: return launch_wgrad4<false, 2>(parts_dev, nparts, nsplit, ws, s);
            case 4: return launch_wgrad4<false, 4>(parts_dev, nparts, nsplit, ws, s);
            case 8: return launch_wgrad4<false, 8>(parts_dev, nparts, nsplit, ws, s);
            case 3: return launch_wgrad4<false, 3>(parts_dev, nparts, nsplit, ws, s);
            case 6: return launch_wgrad4<false, 6>(parts_dev, nparts, nsplit, ws, s);
            case 9: return launch_wgrad4<false, 9>(parts_dev, nparts, nsplit, ws, s);
            case 11: return launch_wgrad4<false, 11>(parts_dev, nparts, nsplit, ws, s);
            case 15: return launch_wgrad4<false, 15>(parts_dev, nparts, nsplit, ws, s);
            default: break;
        }
        if (tr && g_wgrad4 && f32 != 1) return f32 == 2 ? launch_wgrad4<true>(parts_dev, nparts, nsplit, ws, s) : launch_wgrad4<false>(parts_dev, nparts, nsplit, ws, s);
        if (tr && !f32 && g_wgrad3_glds) return launch_wgrad3_glds(parts_dev, nparts, nsplit, ws, s);
        if (tr && !f32) switch (g_wgrad3_abl) {
            case 1: return launch_wgrad3<true, false, false, 1>(parts_dev, nparts, nsplit, ws, s);
            case 2: return launch_wgrad3<true, false, false, 2>(parts_dev, nparts, nsplit, ws, s);
            case 4: return launch_wgrad3<true, false, false, 4>(parts_dev, nparts, nsplit, ws, s);
            case 8: return launch_wgrad3<true, false, false, 8>(parts_dev, nparts, nsplit, ws, s);
            case 6: return launch_wgrad3<true, false, false, 6>(parts_dev, nparts, nsplit, ws, s);
            case 7: return launch_wgrad3<true, false, false, 7>(parts_dev, nparts, nsplit, ws, s);
            case 9: return launch_wgrad3<true, false, false, 9>(parts_dev, nparts, nsplit, ws, s);
            case 14: return launch_wgrad3<true, false, false, 14>(parts_dev, nparts, nsplit, ws, s);
            case 15: return launch_wgrad3<true, false, false, 15>(parts_dev, nparts, nsplit, ws, s);
            default: break;
        }
        if (tr && !g_wgrad3_ld && f32 != 1) return f32 == 2 ? launch_wgrad3<true, false, true>(parts_dev, nparts, nsplit, ws, s) : launch_wgrad3<true, false>(parts_dev, nparts, nsplit, ws, s);
        if (tr && g_wgrad3_ld_abl && f32 == 0) switch (g_wgrad3_ld_abl) {
            case 1: return launch_wgrad3_ld_abl<1>(parts_dev, nparts, nsplit, ws, s);
            case 2: return launch_wgrad3_ld_abl<2>(parts_dev, nparts, nsplit, ws, s);
            case 4: return launch_wgrad3_ld_abl<4>(parts_dev, nparts, nsplit, ws, s);
            default: return launch_wgrad3_ld_abl<3>(parts_dev, nparts, nsplit, ws, s);
        }
        if (tr && g_wgrad3_ld6 && f32 != 1) return f32 == 2 ? launch_wgrad3_ld6<true>(parts_dev, nparts, nsplit, ws, s) : launch_wgrad3_ld6<false>(parts_dev, nparts, nsplit, ws, s);
#endif
        if (!tr || f32 == 1) return DASR_EINVAL;   // gfx950 has ds_read_b64_tr_b16 (dasr_probe_tr16 confirms it); the grouped 3x3 form exists for 16-bit tensors only
        return f32 == 2 ? launch_wgrad3_ld<true>(parts_dev, nparts, nsplit, ws, s) : launch_wgrad3_ld<false>(parts_dev, nparts, nsplit, ws, s);
    }
    if (kh == 3 && stride == 1) return dispatch_wgrad<3, 1>(parts_dev, nparts, nsplit, tr, f32, ws, s);
    if (kh == 4 && stride == 1) return dispatch_wgrad<4, 1>(parts_dev, nparts, nsplit, tr, f32, ws, s);
    if (kh == 4 && stride == 2) return dispatch_wgrad<4, 2>(parts_dev, nparts, nsplit, tr, f32, ws, s);
    if (kh == 5 && stride == 1) return dispatch_wgrad<5, 1>(parts_dev, nparts, nsplit, tr, f32, ws, s);
    if (kh == 1 && stride == 1) return dispatch_wgrad<1, 1>(parts_dev, nparts, nsplit, tr, f32, ws, s);
    if (kh == 3 && stride == 2) return dispatch_wgrad<3, 2>(parts_dev, nparts, nsplit, tr, f32, ws, s);
    return DASR_EINVAL;
}

extern "C" int dasr_wgrad_reduce(const dasr_wgrad_reduce_part* parts_dev, int32_t nparts, const float* ws, float* grad_flat,
                                 float scale, int32_t few_splits, void* stream) {
    if (nparts <= 0) return DASR_EINVAL;
    DASR_LAUNCH(wgrad_reduce_kernel, dim3(few_splits ? 32 : 128, nparts), dim3(256), 0, as_stream(stream), parts_dev, nparts, ws, grad_flat, scale);
    return (int)hipGetLastError();
}
