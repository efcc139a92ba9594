// ---------------------------------------------------------------------------------------------------
// rdb_is_kernel (round 6): the chained trunk launch, INPUT-STATIONARY form.  Included by conv.hip inside its anonymous namespace (it shares
// conv_epilogue, the LDS image and the flag protocol of conv_chain_kernel).
//
// Why.  conv_chain_kernel runs a dense block (ResidualDenseBlock_5C, codes/SRN/models/modules/block.py:254-286) layer by layer: conv_k re-stages its
// whole input, so per block and tile 40 activation chunks (640 channel planes) go L2 -> LDS for 12 distinct ones, and the Cout-32 layers (199 B of
// LDS-DMA per MFMA) run at 60-70 % of the matrix pipe (profiles/r05_chain_trace.txt).  Here every 16-channel chunk of the block's slab is staged ONCE
// and multiplied into the accumulators of EVERY conv that consumes it:
//     group x (4 chunks)  -> conv1 | conv2 conv3 conv4 conv5       group x1 (2 chunks) -> conv2 | conv3 conv4 conv5
//     group x2            -> conv3 | conv4 conv5                   group x3 -> conv4 | conv5          group x4 -> conv5
// ("pass A | pass B": the conv that the group completes runs first, its epilogue stores x_k, and the pass over the remaining convs covers the
// round trip store -> flag -> neighbours' flags -> halo DMA of x_k.)  12 activation chunks instead of 40 per block and tile; the weight volume is
// unchanged (479 KB per block and tile), so 714 KB instead of 1262 KB of LDS-DMA per block and tile.  The MFMA work, the order in which every accumulator
// receives its (chunk, tap) products and the epilogue arithmetic are those of conv_glds_kernel / conv_chain_kernel: results are BIT-IDENTICAL.
//
// Shape.  The five convs of a block need 32 + 32 + 32 + 32 + 64 = 192 output channels of accumulators per pixel; at most 160 are live at once (conv1 is
// finished before the others start).  One workgroup of 8 waves per CU (two waves per SIMD, 256 registers each), tile 16 x 32 pixels, wave w owns rows
// 2w, 2w + 1 (NT = 2 n-tiles of 32 pixels): 160 accumulator registers per lane.  LDS (160 KB, one workgroup): four activation-chunk slots (chunk c of
// the slab lives in slot c & 3: x in 0-3, x1 -> 0,1, x2 -> 2,3, x3 -> 0,1, x4 -> 2,3) + a ring of four 18-KB weight granules, filled three steps ahead.
// A STEP = one weight granule: one conv (9 KB, 18 MFMAs per wave), two Cout-32 convs on the same chunk (2 x 9 KB, the B fragments shared) or conv5
// (18 KB); one raw s_barrier per step, every wait a COUNTED vmcnt (nothing drains the LDS-DMA queue inside the launch).  34 steps per block and tile.
// A workgroup owns `tpw` tiles (tile j of tpw images of its XCD) and walks (block 0, tile 0), (block 0, tile 1), ..., (block 1, tile 0), ...: the wait
// of a tile for its neighbours' conv5 (conv_chain_kernel: 25 k cycles per block, exposed) is covered by the other tile's block.
// Flags, XCD placement (all tiles of image n on XCD n % 8), write-through stores, error word: as conv_chain_kernel.  A neighbour wait is a POLL
// PIPELINE, not a spin: wave 0 requests the nine flag words by LDS-DMA at the start of a step, every wave reads them from LDS after the step's barrier;
// only when a group is still missing where it is needed does the workgroup spin (and counts it: err stays 0, the time shows in the trace).
// ---------------------------------------------------------------------------------------------------
#ifndef IS_ABL
#define IS_ABL 0   // timing experiments with WRONG results (scripts/r06/is_ablate.sh): 1 no epilogues, 2 no weight DMA, 4 no MFMA, 8 no activation DMA, 16 no residual preload
#endif
#ifndef IS_BORDER
#define IS_BORDER 0   // 1: epilogues store the border pixels of their 16-bit planes first and the flag waits for those only (conv_epilogue BMODE).  Built, bit-identical, and
                      // SLOWER (9.9 / 10.6 ms per chain against 8.7 / 9.0): the second pass doubles the epilogue's store instructions and arithmetic -- profiles/r06_is_chain.txt
#endif
#ifndef IS_WDIST
#define IS_WDIST 3   // 16-row tiles: the weights of step g + IS_WDIST are requested in step g (ring of four granules: at most 3; 2 measured the same)
#endif
#ifndef IS_PUB_DELAY
#define IS_PUB_DELAY 1
#endif
struct ISC {   // what does not depend on the tile height
    static constexpr int TW = 32, IW = 34;
    static constexpr int WGRAN = 18432, NSLOT = 4;
    static constexpr int LDS_BYTES = 163840;
    static constexpr int X_OFF = LDS_BYTES - 8192;           // the last 8 KB: biases, polled flag words, decision words (the same offsets for every tile height)
    static constexpr int BIAS_OFF = X_OFF;                   // [2 item parities][5 layers][64 floats]
    static constexpr int POLL_OFF = X_OFF + 2560;            // [MAX_TPW][16 words]: the newest flag values seen of a tile's nine (self + 8 neighbours) words
    static constexpr int MAX_TPW = 8;
    static constexpr int F0_OFF = POLL_OFF + MAX_TPW * 64;   // f0[MAX_TPW], ticket
    static constexpr int DEC_OFF = F0_OFF + 64;              // decision words [2 step parities][2]: {target reached ? target : target - 1, tile slot}, written by wave 0
    static constexpr int TRC_OFF = DEC_OFF + 64;             // trace builds: 16 64-bit accumulators
    static constexpr int NSTEP = 34;
    static constexpr int PUB_DELAY = IS_PUB_DELAY;           // a conv's flag goes out at the end of the PUB_DELAY-th step behind its epilogue (its stores have that long to be acknowledged)
};
// NT = output rows per wave: 2 = tiles of 16 x 32 pixels (the full-chip shapes), 1 = tiles of 8 x 32 (launches that would otherwise fill less of the chip: the reference's
// shipped 32 x 32 crops run as 64 instead of 32 workgroups; half the MFMA work per step and workgroup)
// NW = waves per workgroup (round 6, last step): a chained launch is a SEQUENCE of 34 x 69 steps whose length does not shrink with the tile, and a step of the 8-row tiles
// is bound by the LDS port, not by the matrix pipe (8 waves x 18-27 KB of fragment reads per step against 9-18 MFMAs per wave: every weight fragment feeds ONE MFMA per
// wave).  Fewer waves per workgroup = fewer pixels per workgroup = more workgroups for the same image AND less LDS traffic per CU and step: tiles of 4 x 32 pixels (NT 1, NW 4)
// let the reference's shipped 16 crops of 32 x 32 run as 128 workgroups (2 x 32 with two waves was built too: slower, the neighbour-sync latency chain is the bound by then).
template <int NT_, int NW_ = 8>
struct ISCfg : ISC {
    static constexpr int NW = NW_, NTH = 64 * NW_;
    static constexpr int NT = NT_, TH = NW * NT, IH = TH + 2, NPIX = IH * IW;
    static constexpr int APIECE = NPIX * 2;                            // 16-byte pieces of one activation chunk (1224 / 680)
    static constexpr int AR = (APIECE + NTH - 1) / NTH;                // DMA rounds per chunk (the last one partial)
    static constexpr int ACT_SLOT = (APIECE * 16 + 1023) / 1024 * 1024;   // 20480 / 11264
    static constexpr int W_OFF = NSLOT * ACT_SLOT;
    // ring of weight granules and how many steps ahead they are requested.  (8-row tiles have room for six granules, five steps ahead -- the same lead TIME as three steps
    // of the 16-row tiles; measured: 2.53 / 2.59 ms per chain of the 32 x 32-crop launches against 2.43 / 2.52 ms with four / three.  Neither form waits for its weights.)
    static constexpr int NWG = 4, WDIST = IS_WDIST;
    static_assert(W_OFF + NWG * WGRAN <= X_OFF && WDIST < NWG && WDIST >= 2 && WDIST <= 5, "LDS budget / ring depth");
};

// the static program of one dense block on one tile: kind 1 = one Cout-32 conv u0, 2 = two Cout-32 convs u0, u1 on the same chunk, 5 = conv5; c = chunk of the slab
struct ISStep { int kind, u0, u1, c; };
constexpr ISStep IS_PROG[ISC::NSTEP] = {
    {1, 0, 0, 0}, {1, 0, 0, 1}, {1, 0, 0, 2}, {1, 0, 0, 3},                                                      // x -> conv1            | epilogue conv1 (x1)
    {2, 1, 2, 0}, {1, 3, 0, 0}, {5, 4, 4, 0}, {2, 1, 2, 1}, {1, 3, 0, 1}, {5, 4, 4, 1},
    {2, 1, 2, 2}, {1, 3, 0, 2}, {5, 4, 4, 2}, {2, 1, 2, 3}, {1, 3, 0, 3}, {5, 4, 4, 3},                          // x -> conv2-5
    {1, 1, 0, 4}, {1, 1, 0, 5},                                                                                  // x1 -> conv2           | epilogue conv2 (x2)
    {2, 2, 3, 4}, {5, 4, 4, 4}, {2, 2, 3, 5}, {5, 4, 4, 5},                                                      // x1 -> conv3-5
    {1, 2, 0, 6}, {1, 2, 0, 7},                                                                                  // x2 -> conv3           | epilogue conv3 (x3)
    {1, 3, 0, 6}, {5, 4, 4, 6}, {1, 3, 0, 7}, {5, 4, 4, 7},                                                      // x2 -> conv4, conv5
    {1, 3, 0, 8}, {1, 3, 0, 9},                                                                                  // x3 -> conv4           | epilogue conv4 (x4)
    {5, 4, 4, 8}, {5, 4, 4, 9},                                                                                  // x3 -> conv5
    {5, 4, 4, 10}, {5, 4, 4, 11}};                                                                               // x4 -> conv5           | epilogue conv5
// group k (x_k, k = 1..4; 5 = the next item's x): awaited during steps [IS_WIN0, IS_WIN1), its DMA may be issued from step IS_WIN0 on (the slots are free then), it is
// read first in step IS_WIN1
constexpr int IS_WIN0[6] = {0, 10, 18, 24, 30, 32}, IS_WIN1[6] = {0, 16, 22, 28, 32, 34};
constexpr int is_group_of_step(int t) {
    for (int k = 1; k <= 5; ++k)
        if (t >= IS_WIN0[k] && t < IS_WIN1[k]) return k;
    return 0;
}
constexpr int is_epi_after(int t) { return t == 3 ? 0 : t == 17 ? 1 : t == 23 ? 2 : t == 29 ? 3 : t == 33 ? 4 : -1; }

template <int TV>
struct ISInt { static constexpr int value = TV; };

__device__ __forceinline__ void is_wait_vm(int n) {   // s_waitcnt vmcnt(min(n, 31)), n wave-uniform (a smaller count than allowed only waits longer)
    switch (n) {
        case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;
        case 1: __builtin_amdgcn_s_waitcnt(0x0F71); break;
        case 2: __builtin_amdgcn_s_waitcnt(0x0F72); break;
        case 3: __builtin_amdgcn_s_waitcnt(0x0F73); break;
        case 4: __builtin_amdgcn_s_waitcnt(0x0F74); break;
        case 5: __builtin_amdgcn_s_waitcnt(0x0F75); break;
        case 6: __builtin_amdgcn_s_waitcnt(0x0F76); break;
        case 7: __builtin_amdgcn_s_waitcnt(0x0F77); break;
        case 8: __builtin_amdgcn_s_waitcnt(0x0F78); break;
        case 9: __builtin_amdgcn_s_waitcnt(0x0F79); break;
        case 10: __builtin_amdgcn_s_waitcnt(0x0F7A); break;
        case 11: __builtin_amdgcn_s_waitcnt(0x0F7B); break;
        case 12: __builtin_amdgcn_s_waitcnt(0x0F7C); break;
        case 13: __builtin_amdgcn_s_waitcnt(0x0F7D); break;
        case 14: __builtin_amdgcn_s_waitcnt(0x0F7E); break;
        case 15: __builtin_amdgcn_s_waitcnt(0x0F7F); break;
        case 16: __builtin_amdgcn_s_waitcnt(0x4F70); break;
        case 17: __builtin_amdgcn_s_waitcnt(0x4F71); break;
        case 18: __builtin_amdgcn_s_waitcnt(0x4F72); break;
        case 19: __builtin_amdgcn_s_waitcnt(0x4F73); break;
        case 20: __builtin_amdgcn_s_waitcnt(0x4F74); break;
        case 21: __builtin_amdgcn_s_waitcnt(0x4F75); break;
        case 22: __builtin_amdgcn_s_waitcnt(0x4F76); break;
        case 23: __builtin_amdgcn_s_waitcnt(0x4F77); break;
        case 24: __builtin_amdgcn_s_waitcnt(0x4F78); break;
        case 25: __builtin_amdgcn_s_waitcnt(0x4F79); break;
        case 26: __builtin_amdgcn_s_waitcnt(0x4F7A); break;
        case 27: __builtin_amdgcn_s_waitcnt(0x4F7B); break;
        case 28: __builtin_amdgcn_s_waitcnt(0x4F7C); break;
        case 29: __builtin_amdgcn_s_waitcnt(0x4F7D); break;
        case 30: __builtin_amdgcn_s_waitcnt(0x4F7E); break;
        default: __builtin_amdgcn_s_waitcnt(0x4F7F); break;
    }
}

// LDS-DMA helpers (free functions: see glds_dma_piece).  Every one returns the number of VMEM instructions THIS WAVE issued (wave-uniform).
template <int NTH>
__device__ __forceinline__ int is_dma_w(__amdgpu_buffer_rsrc_t rw, char* dst, unsigned src_off, int npieces, int wave, int tid) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    int n = 0;
#pragma unroll
    for (int r = 0; r < (1152 + NTH - 1) / NTH; ++r) {   // (a granule has at most 1152 pieces)
        if (r * NTH + wave * 64 < npieces) {   // (whole waves: npieces is a multiple of 64)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(dst + (r * NTH + wave * 64) * 16), 16, (unsigned)(tid + r * NTH) * 16u, src_off, 0, 0);
            ++n;
        }
    }
    return n;
}
template <class CC>
__device__ __forceinline__ int is_dma_act(__amdgpu_buffer_rsrc_t rin, char* slot, const unsigned* goff, unsigned src_off, int wave, int tid) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    int n = 0;
#pragma unroll
    for (int r = 0; r < CC::AR; ++r) {
        if (r * CC::NTH + wave * 64 < CC::APIECE) {
            if (tid + r * CC::NTH < CC::APIECE)   // (the last wave of the last round is partial: the pieces behind the chunk would land in the next slot)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_ptr)(slot + (r * CC::NTH + wave * 64) * 16), 16, goff[r], src_off, 0, 0);
            ++n;
        }
    }
    return n;
}
// nine flag words (self + eight neighbours; a missing neighbour = self) -> LDS, past the L1 / L2 (sc0 sc1); lanes 0-8 of one wave
__device__ __forceinline__ void is_dma_poll(__amdgpu_buffer_rsrc_t rflags, char* dst, unsigned voff) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rflags, (lds_ptr)dst, 4, voff, 0, 0, 17);
}
__device__ __forceinline__ void is_dma_bias(__amdgpu_buffer_rsrc_t rb, char* dst, unsigned voff) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)dst, 4, voff, 0, 0, 0);
}

template <class CC>
struct ISGeo {            // one tile of the workgroup's list
    int n, oy0, ox0;
    unsigned goff[CC::AR];   // per-thread source offsets of the activation DMA pieces
    unsigned nbo;             // lanes 0-8: byte offset of the flag word of neighbour `lane` (self where there is none)
    unsigned selfo;           // byte offset of the tile's own flag word
    unsigned f0;              // the tile's flag value at kernel start
    int slot;
};

#ifdef DASR_TRACE
#define IS_T() (g_trace && threadIdx.x == 0 ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
// (accumulated in LDS, copied out once at the end: a global read-modify-write per stamp would sit in the VMEM queue the kernel counts)
#define IS_ACC(k, val)                                                                                     \
    do {                                                                                                   \
        if (g_trace && threadIdx.x == 0) ((unsigned long long*)(smem + ISC::TRC_OFF))[k] += (val);           \
    } while (0)
#else
#define IS_T() 0ull
#define IS_ACC(k, val) do {} while (0)
#endif

// one step's MFMA body.  a0 / a1: the accumulators of the step's one or two m-tiles (NT = 2 n-tiles each); A fragment of (tap, m-tile mi) at
// wbuf + tap * TS + mi * MIS (+ lane * 16): conv5's granule is [tap][mi][1 KB] (TS 2048, MIS 1024), two Cout-32 convs are [conv][tap][1 KB] (TS 1024, MIS 9216).
// B fragments: the wave's four input rows per kx, reused across ky (as conv_glds_kernel).  mid(s) runs between the fragment requests and the MFMAs of tap step s.
template <int NT, int NU, int TS, int MIS, bool F16, class Mid>
__device__ __forceinline__ void is_body(f32x16 (&a0)[NT], f32x16 (&a1)[NT], const char* abuf, const char* wbuf, const int (&baddr)[NT + 2][3], const int aoff, Mid&& mid) {
    // PD = how many taps ahead the weight fragments are requested.  One-row tiles (NT = 1) issue one or two MFMAs per tap (32 - 64 cycles): one tap of lead does not
    // cover the LDS round trip, and with one wave per SIMD (the 4- and 2-wave workgroups) nothing else does -- three taps of lead there (a ring of four fragment sets);
    // the activation rows of the next kx are requested a whole kx (three taps) ahead.  Two-row tiles keep one tap (256 registers, none to spare).
    constexpr int PD = NT == 1 ? 3 : 1, R = PD + 1;
    bf16x8 fb[2][NT + 2], fa[R][NU];
#pragma unroll
    for (int rr = 0; rr < NT + 2; ++rr) fb[0][rr] = *(const bf16x8*)(abuf + baddr[rr][0]);
#pragma unroll
    for (int t = 0; t < PD; ++t)
#pragma unroll
        for (int mi = 0; mi < NU; ++mi) fa[t][mi] = *(const bf16x8*)(wbuf + aoff + ((t % 3) * 3 + t / 3) * TS + mi * MIS);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int kx = s / 3, ky = s - kx * 3;
        if (s + PD < 9) {
            const int kx1 = (s + PD) / 3, ky1 = (s + PD) - kx1 * 3;
#pragma unroll
            for (int mi = 0; mi < NU; ++mi) fa[(s + PD) % R][mi] = *(const bf16x8*)(wbuf + aoff + (ky1 * 3 + kx1) * TS + mi * MIS);
        }
        if (ky == (PD >= 3 ? 0 : 1) && kx < 2) {
#pragma unroll
            for (int rr = 0; rr < NT + 2; ++rr) fb[(kx + 1) & 1][rr] = *(const bf16x8*)(abuf + baddr[rr][kx + 1]);
        }
        mid(s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (IS_ABL & 4) {
                asm volatile("" ::"v"(fa[s % R][0]), "v"(fb[kx & 1][nt + ky]));
                if constexpr (NU == 2) asm volatile("" ::"v"(fa[s % R][1]));
            } else {
                a0[nt] = mfma16<F16>(fa[s % R][0], fb[kx & 1][nt + ky], a0[nt]);
                if constexpr (NU == 2) a1[nt] = mfma16<F16>(fa[s % R][1], fb[kx & 1][nt + ky], a1[nt]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}


template <bool F16, bool BWD, int NT, int NW>
__global__ __launch_bounds__(64 * NW, NW >= 8 ? 2 : 1) void rdb_is_kernel(const dasr_conv_params* __restrict__ layers, const int nrdb, const int tiles_y, const int tiles_x, const int tpw,
                                                        unsigned* flags, unsigned* tickets, int* err, const int stagger) {
    using C = ISCfg<NT, NW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = tiles_y * tiles_x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int xcd = (int)(xcc & 7u);
    const int quota = (int)(gridDim.x >> 3);
    int* xi = (int*)(smem + C::F0_OFF);   // [0 .. MAX_TPW) f0 of this workgroup's tiles, [MAX_TPW] ticket
    const __amdgpu_buffer_rsrc_t rflags = make_rsrc(flags);
    if (tid == 0) xi[C::MAX_TPW] = (int)(atomicAdd(tickets + xcd, 1u) % (unsigned)quota);
    __syncthreads();
    const int j = __builtin_amdgcn_readfirstlane(xi[C::MAX_TPW]);
    if (tid < tpw) {   // stage base of every tile this workgroup owns (the flag words count on from launch to launch)
        const int idx = j + quota * tid;
        const int img = idx / T, tile = idx - img * T;
        xi[tid] = (int)__builtin_amdgcn_raw_buffer_load_b32(rflags, (unsigned)((xcd + 8 * img) * T + tile) * 4u, 0, 17);
    }
    __syncthreads();
    if (tid < tpw * 16) ((int*)(smem + C::POLL_OFF))[tid] = xi[tid >> 4];   // "newest value seen" of every polled word: a lower bound of the truth at all times
    if (tid < 4) ((unsigned*)(smem + C::DEC_OFF))[tid] = 0xffffffffu;
#ifdef DASR_TRACE
    if (tid < 16) ((unsigned long long*)(smem + C::TRC_OFF))[tid] = 0ull;
#endif
    CH_WHERE(j, xcc);
    const int Hin = layers[0].Hin, Win = layers[0].Win;
    __syncthreads();

    auto geo_of = [&](int slot) {
        ISGeo<C> q;
        const int idx = j + quota * slot;
        const int img = idx / T, tile = idx - img * T;
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        q.n = xcd + 8 * img;   // all tiles of image n on XCD n % 8
        q.oy0 = ty * C::TH, q.ox0 = tx * C::TW;
        q.slot = slot;
#pragma unroll
        for (int r = 0; r < C::AR; ++r) {
            const int qq = tid + r * C::NTH;
            const int pp = qq >> 1, h = (qq & 1) ^ ((pp >> 3) & 1);
            const int iy = pp / C::IW, ix = pp - iy * C::IW;
            const int gy = q.oy0 - 1 + iy, gx = q.ox0 - 1 + ix;
            const bool ok = (pp < C::NPIX) & (gy >= 0) & (gy < Hin) & (gx >= 0) & (gx < Win);
            q.goff[r] = ok ? (unsigned)(((gy * Win + gx) * 16 + 8 * h) * 2) : OOB;
        }
        q.selfo = (unsigned)(q.n * T + tile) * 4u;
        {
            const int l9 = lane < 9 ? lane : 4;
            const int y = ty + l9 / 3 - 1, x = tx + l9 % 3 - 1;
            const bool in = (y >= 0) & (y < tiles_y) & (x >= 0) & (x < tiles_x);
            q.nbo = in ? (unsigned)(q.n * T + y * tiles_x + x) * 4u : q.selfo;
        }
        q.f0 = (unsigned)__builtin_amdgcn_readfirstlane(xi[slot]);
        return q;
    };

    // ---- per-lane fragment addresses (the image of conv_glds_kernel; wave w: output rows 2w, 2w + 1 = input rows 2w .. 2w + 3)
    const int nn = lane & 31, kh2 = lane >> 5;
    int baddr[NT + 2][3];
#pragma unroll
    for (int rr = 0; rr < NT + 2; ++rr)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int pp = (wave * C::NT + rr) * C::IW + nn + kx;
            baddr[rr][kx] = ((pp << 1) + (kh2 ^ ((pp >> 3) & 1))) << 4;
        }
    const int aoff = lane * 16;

    // ---- VMEM bookkeeping of this wave (LDS-DMA instructions only: epilogue loads / stores are not counted, which can only make a wait longer)
    int issued = 0;
    int mk[8] = {0, 0, 0, 0, 0, 0, 0, 0};           // mk[k]: `issued` at the end of step g - 1 - k (g: the current step)
    int g = 0;                                     // global step counter: the granule of step g lives in ring slot g % NWG
    bool pub_on = false;                           // a flag store is pending: value pub_val to word pub_off once every op up to pub_mark is acknowledged, at the end of step pub_due
    int pub_mark = 0, pub_due = 0, pub_late = 0;
    unsigned pub_off = 0, pub_val = 0;
    int own_mark = 0;                              // `issued` behind ALL stores of the newest epilogue
    int mark_x = 0;                                // `issued` behind the DMA of the current item's x chunks
    bool poll_pend = false;                        // wave 0: a poll of flag words is in flight (requested behind poll_mark)
    int poll_mark = 0;
    const int total_steps = nrdb * tpw * C::NSTEP;

    auto flag_store = [&]() {
        if (tid == 0) __builtin_amdgcn_raw_buffer_store_b32(pub_val, rflags, pub_off, 0, 16);
        pub_on = false;
    };
    auto flush_pub = [&]() {   // blocking form: every store of this workgroup acknowledged, then the flag
        if (pub_on) {
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
            flag_store();
        }
    };
    // (Round 6 also built the publish as a non-blocking queue -- every wave checks IB_STS.vm_cnt at the end of a step, wave 0 stores the flag once all eight have seen their
    // stores acknowledged: 9.7 ms per chain against 8.7 ms.  The flag then leaves up to two steps later, and what bounds this kernel is exactly that chain: epilogue stores
    // acknowledged -> flag -> the neighbours' poll -> their halo DMA; waiting for the stores at the end of the next step is the shortest form.  profiles/r06_is_chain.txt)
    // spin until they have (a group is not there where it is needed): the only place where the matrix pipe waits for a neighbour
    auto block_until = [&](const ISGeo<C>& q, unsigned target) {
        const unsigned long long t0 = IS_T();
        flush_pub();
        if (wave == 0 && lane < 9) {
            int spins = 0;
            unsigned v;
            while ((int)((v = __builtin_amdgcn_raw_buffer_load_b32(rflags, q.nbo, 0, 17)) - target) < 0) {
                __builtin_amdgcn_s_sleep(2);
                ++spins;
                if (spins > (1 << 21) || ((spins & 1023) == 0 && __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(err), 0, 0, 17) != 0)) {
                    atomicOr(err, 2);
                    break;
                }
            }
            ((unsigned*)(smem + C::POLL_OFF))[q.slot * 16 + lane] = v;
        }
        __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        IS_ACC(3, IS_T() - t0);
        IS_ACC(11, 1ull);
    };

    // XCD stagger (tuning key 9): XCD k starts k * stagger * ~4 us late, so that the eight XCDs (which never wait for each other: whole images per XCD) reach
    // their store / DMA bursts at different times
    for (int i = 0; i < xcd * stagger; ++i) __builtin_amdgcn_s_sleep(127);
    ISGeo<C> cur = geo_of(0);
    // ---- prologue: x of the first item, the weights of steps 0-2, the biases of the first item
    {
        const dasr_conv_params& p0 = layers[0];
        const __amdgpu_buffer_rsrc_t rin0 = make_rsrc((const bf16_t*)p0.in.p + (size_t)cur.n * p0.in.n_stride);
        const unsigned icb0 = (unsigned)(p0.in.cb_stride * 2);
#pragma unroll
        for (int c = 0; c < 4; ++c) issued += is_dma_act<C>(rin0, smem + c * C::ACT_SLOT, cur.goff, (unsigned)c * icb0, wave, tid);
        mark_x = issued;
        auto pro = [&](auto tc) {   // the granule of step t of the first item
            constexpr int t = decltype(tc)::value;
            if constexpr (t < C::WDIST) {
                constexpr ISStep e = IS_PROG[t];
                char* dst = smem + C::W_OFF + t * C::WGRAN;
                if constexpr (e.kind == 5) {
                    issued += is_dma_w<C::NTH>(make_rsrc(layers[4].w), dst, (unsigned)e.c * 18432u, 1152, wave, tid);
                } else {
                    issued += is_dma_w<C::NTH>(make_rsrc(layers[e.u0].w), dst, (unsigned)e.c * 9216u, 576, wave, tid);
                    if constexpr (e.kind == 2) issued += is_dma_w<C::NTH>(make_rsrc(layers[e.u1].w), dst + 9216, (unsigned)e.c * 9216u, 576, wave, tid);
                }
            }
        };
        pro(ISInt<0>{}), pro(ISInt<1>{}), pro(ISInt<2>{}), pro(ISInt<3>{}), pro(ISInt<4>{}), pro(ISInt<5>{}), pro(ISInt<6>{});
        if constexpr (!BWD) {
#pragma unroll
            for (int l = 0; l < 5; l += NW) {   // (layer l + wave: one wave per layer, the waves of a small workgroup take several)
                if (l + wave < 5) {
                    const dasr_conv_params& pb = layers[l + wave];
                    is_dma_bias(make_rsrc(pb.bias), smem + C::BIAS_OFF + (l + wave) * 256, (pb.bias != nullptr && lane < 32 * pb.mt) ? (unsigned)lane * 4u : OOB);
                    ++issued;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) mk[k] = issued;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // (the one full drain of the launch: granule 0 is read by the first step)
    }

    f32x16 A0[1][NT], A1[1][NT], A2[1][NT], A3[1][NT], A5[2][NT];

    int item = 0;
    for (int r = 0; r < nrdb; ++r) {
        for (int slot = 0; slot < tpw; ++slot, ++item) {
            const int L0 = 5 * r;
            const int par = item & 1;
            const dasr_conv_params& pl0 = layers[L0];
            const __amdgpu_buffer_rsrc_t rin = make_rsrc((const bf16_t*)pl0.in.p + (size_t)cur.n * pl0.in.n_stride);
            const unsigned icb = (unsigned)(pl0.in.cb_stride * 2);
            // the item behind this one
            const int sn = slot + 1 < tpw ? slot + 1 : 0, rn = slot + 1 < tpw ? r : r + 1;
            const bool has_next = rn < nrdb;
            const int L0n = has_next ? 5 * rn : L0;
            // packed weights of this item's five convs and of the next item's conv1 (scalar registers: no scalar-memory round trip inside a step)
            const void* wp0 = layers[L0].w;
            const void* wp1 = layers[L0 + 1].w;
            const void* wp2 = layers[L0 + 2].w;
            const void* wp3 = layers[L0 + 3].w;
            const void* wp4 = layers[L0 + 4].w;
            const void* wpn0 = layers[L0n].w;       // (a step requests the granule of WDIST <= 5 steps ahead: steps 0-4 of the next item = its conv1, conv2, conv3)
            const void* wpn1 = layers[L0n + 1].w;
            const void* wpn2 = layers[L0n + 2].w;
            auto wptr = [&](int u, bool next) -> const void* {
                return next ? (u == 0 ? wpn0 : u == 1 ? wpn1 : wpn2) : (u == 0 ? wp0 : u == 1 ? wp1 : u == 2 ? wp2 : u == 3 ? wp3 : wp4);
            };

            // x of this item is in LDS (requested at the end of the previous item / in the prologue)
            {
                const unsigned long long t0 = IS_T();
                is_wait_vm(__builtin_amdgcn_readfirstlane(issued - mark_x));
                __builtin_amdgcn_s_barrier();
                IS_ACC(4, IS_T() - t0);
            }
            // conv5's accumulators start from (beta1 / alpha) * x in fp32 (R1_PRE of conv_glds_kernel: the MFMAs accumulate on top, the epilogue scales by alpha).
            // Requested here, scaled in front of conv1's epilogue (step 3): four steps for the round trip.
            // (A conv5 WITHOUT a 16-bit shadow -- the last block of the trunk -- is one dasr_conv runs with its generic epilogue, alpha * acc + beta1 * x: same here.)
            const dasr_conv_params& p5 = layers[L0 + 4];
            const bool r1pre = p5.out_bf16.p != nullptr && !(IS_ABL & 16);
            if (r1pre) {
                const __amdgpu_buffer_rsrc_t rr1 = make_rsrc((const float*)p5.res1.p + (size_t)cur.n * p5.res1.n_stride);
                const unsigned r1_cb = (unsigned)p5.res1.cb_stride;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int oy = cur.oy0 + wave * NT + nt, ox = cur.ox0 + nn;
                    const bool pv = (oy < p5.Hout) & (ox < p5.Wout);
                    const unsigned pixel = (unsigned)(oy * p5.Wout + ox) * 16u;
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const int oc = mi * 32 + 8 * gq + 4 * kh2;
                            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rr1, pv ? ((unsigned)(oc >> 4) * r1_cb + pixel + (unsigned)(oc & 15)) * 4u : OOB, 0, 0);
#pragma unroll
                            for (int e = 0; e < 4; ++e) A5[mi][nt][4 * gq + e] = __uint_as_float(t[e]);
                        }
                }
            } else {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) A5[mi][nt][e] = 0.f;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) A0[0][nt][e] = 0.f, A1[0][nt][e] = 0.f, A2[0][nt][e] = 0.f, A3[0][nt][e] = 0.f;

            bool arr_issued = false;     // the group awaited in the current window has been requested
            int mark_act = 0;            // `issued` behind that request
            bool next_x_issued = false, next_rdy = false;
            const unsigned f0n = (unsigned)__builtin_amdgcn_readfirstlane(xi[sn]);

            auto request_group = [&](int c0) {   // chunks c0, c0 + 1 of this item's slab
                if constexpr (IS_BORDER) is_wait_vm(__builtin_amdgcn_readfirstlane(issued - own_mark));   // (the flag covered the border only: this tile's own interior pixels must be in memory)
                if constexpr (!(IS_ABL & 8)) issued += is_dma_act<C>(rin, smem + (c0 & 3) * C::ACT_SLOT, cur.goff, (unsigned)c0 * icb, wave, tid);
                if constexpr (!(IS_ABL & 8)) issued += is_dma_act<C>(rin, smem + ((c0 + 1) & 3) * C::ACT_SLOT, cur.goff, (unsigned)(c0 + 1) * icb, wave, tid);
                arr_issued = true;
                mark_act = issued;
            };
            auto request_next_x = [&](const ISGeo<C>& nxt) {
                const dasr_conv_params& pn = layers[L0n];
                const __amdgpu_buffer_rsrc_t rinn = make_rsrc((const bf16_t*)pn.in.p + (size_t)nxt.n * pn.in.n_stride);
                const unsigned icbn = (unsigned)(pn.in.cb_stride * 2);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if constexpr (!(IS_ABL & 8)) issued += is_dma_act<C>(rinn, smem + c * C::ACT_SLOT, nxt.goff, (unsigned)c * icbn, wave, tid);
                mark_x = issued;
                next_x_issued = true;
            };

            auto do_step = [&](auto tc) {
                constexpr int TT = decltype(tc)::value;
                constexpr ISStep d = IS_PROG[TT];
                constexpr int GK = is_group_of_step(TT);            // group awaited during this step (0: none)
                constexpr bool MUST = GK >= 1 && GK <= 4 && TT + 1 == IS_WIN1[GK];   // the group is read in the next step
                const char* abuf = smem + (d.c & 3) * C::ACT_SLOT;
                const char* wbuf = smem + C::W_OFF + (g % C::NWG) * C::WGRAN;
                char* wnext = smem + C::W_OFF + ((g + C::WDIST) % C::NWG) * C::WGRAN;
                if constexpr (GK != 0 && TT == IS_WIN0[GK > 0 ? GK : 1]) arr_issued = false;
                // ---- what this step requests besides its MFMAs.  Whether a group's nine flag words have reached their target is decided by WAVE 0 ALONE (at the end of
                // the step in which it polled, from the words its LDS-DMA brought) and handed to the other waves through a decision word in LDS across the step's
                // barrier: every wave takes the same branch (a wave reading the polled words itself could see a newer poll than its neighbours)
                bool act_now = false, poll_now = false;
                unsigned target = 0;
                int pslot = 0;
                auto decided = [&]() -> bool {   // wave 0's decision of the previous step: this (target, tile)?
                    const unsigned* dw = (const unsigned*)(smem + C::DEC_OFF) + ((g - 1) & 1) * 2;
                    return __builtin_amdgcn_readfirstlane(dw[0]) == target && __builtin_amdgcn_readfirstlane(dw[1]) == (unsigned)pslot;
                };
                bool need = false;   // a group is awaited and not yet known to be there
                if constexpr (GK >= 1 && GK <= 4) {
                    target = cur.f0 + (unsigned)(L0 + GK), pslot = cur.slot;
                    if (!arr_issued) {
                        if (decided()) act_now = true;
                        else need = true, poll_now = !pub_on && !poll_pend;   // (our own flag is one of the nine: nothing to see before it is out)
                    }
                } else if constexpr (GK == 5) {
                    if (has_next && rn > 0 && !next_rdy) {   // (block 0 reads what an earlier kernel wrote)
                        target = f0n + (unsigned)L0n, pslot = sn;
                        if (decided()) next_rdy = true;
                        else need = true, poll_now = !poll_pend;
                    }
                }
                constexpr int EU_AFTER = is_epi_after(TT);
                auto prefetch_w = [&]() {   // weights of step g + 3 (g: this step)
                    if (g + C::WDIST < total_steps && !(IS_ABL & 2)) {
                        constexpr int T3 = (TT + C::WDIST) % C::NSTEP;
                        constexpr ISStep e = IS_PROG[T3];
                        constexpr bool NX = TT + C::WDIST >= C::NSTEP;
                        if constexpr (e.kind == 5) {
                            issued += is_dma_w<C::NTH>(make_rsrc(wptr(4, NX)), wnext, (unsigned)e.c * 18432u, 1152, wave, tid);
                        } else {
                            issued += is_dma_w<C::NTH>(make_rsrc(wptr(e.u0, NX)), wnext, (unsigned)e.c * 9216u, 576, wave, tid);
                            if constexpr (e.kind == 2) issued += is_dma_w<C::NTH>(make_rsrc(wptr(e.u1, NX)), wnext + 9216, (unsigned)e.c * 9216u, 576, wave, tid);
                        }
                    }
                };
                auto mid = [&](int s) {
                    if (s == 0) {
                        if (poll_now && wave == 0) {
                            if constexpr (GK == 5) {
                                const ISGeo<C> nq = geo_of(sn);
                                if (lane < 9) is_dma_poll(rflags, smem + C::POLL_OFF + sn * 64, nq.nbo);
                            } else {
                                if (lane < 9) is_dma_poll(rflags, smem + C::POLL_OFF + cur.slot * 64, cur.nbo);
                            }
                            ++issued;
                            poll_mark = issued;
                            poll_pend = true;
                        }
                        if constexpr (EU_AFTER < 0) prefetch_w();   // (a step that ends in an epilogue requests them BEHIND the epilogue's stores: a store queued behind LDS-DMA
                                                                    // instructions waits for their data, profiles/r05_chain_trace.txt form 2 / r06_is_chain.txt)
                    }
                    if (s == 2) {
                        if constexpr (GK >= 1 && GK <= 4) {
                            if (act_now) request_group(4 + 2 * (GK - 1));
                        }
                    }
                };
                const unsigned long long t0 = IS_T();
                if constexpr (d.kind == 1) {
                    if constexpr (d.u0 == 0) is_body<NT, 1, 1024, 0, F16>(A0[0], A0[0], abuf, wbuf, baddr, aoff, mid);
                    else if constexpr (d.u0 == 1) is_body<NT, 1, 1024, 0, F16>(A1[0], A1[0], abuf, wbuf, baddr, aoff, mid);
                    else if constexpr (d.u0 == 2) is_body<NT, 1, 1024, 0, F16>(A2[0], A2[0], abuf, wbuf, baddr, aoff, mid);
                    else is_body<NT, 1, 1024, 0, F16>(A3[0], A3[0], abuf, wbuf, baddr, aoff, mid);
                } else if constexpr (d.kind == 2) {
                    if constexpr (d.u0 == 1) is_body<NT, 2, 1024, 9216, F16>(A1[0], A2[0], abuf, wbuf, baddr, aoff, mid);
                    else is_body<NT, 2, 1024, 9216, F16>(A2[0], A3[0], abuf, wbuf, baddr, aoff, mid);
                    static_assert(d.kind != 2 || (d.u0 == 1 && d.u1 == 2) || (d.u0 == 2 && d.u1 == 3), "pairs of the program");
                } else {
                    is_body<NT, 2, 2048, 1024, F16>(A5[0], A5[1], abuf, wbuf, baddr, aoff, mid);
                }
                // data gradient: the LeakyReLU' mask planes of the conv this step completes are requested HERE, behind the step's last MFMA and in front of its wait + barrier
                // (the fragment registers are dead, the weight requests of such a step are held back until after the epilogue: the loads have the queue to themselves and
                // fly during the barrier; fetched inside the epilogue they waited in order behind the LDS-DMA in flight -- 29 k against 17 k cycles per item, r06_is_chain.txt)
                MaskPre<NT> mpre;
                if constexpr (BWD && EU_AFTER >= 0 && EU_AFTER < 4 && !IS_BORDER && !(IS_ABL & 1)) mask_prefetch<1, NT>(layers[L0 + EU_AFTER], mpre, tid, 0, cur.n, cur.oy0, cur.ox0);
                const unsigned long long t1 = IS_T();
                // ---- end of the step: everything requested up to the end of step g - 2 has landed (the granule of step g + 1 among it); flag words / publish / group as due
                int nwait = issued - mk[C::WDIST - 2];   // everything requested up to the end of step g + 1 - WDIST: the granule of step g + 1 among it
                bool do_pub = false;
                if (pub_on && g >= pub_due) do_pub = true, nwait = min(nwait, issued - pub_mark + pub_late);
                if constexpr (MUST) {
                    if (arr_issued) nwait = min(nwait, issued - mark_act);
                }
                is_wait_vm(__builtin_amdgcn_readfirstlane(nwait));
                const unsigned long long t1b = IS_T();
                if (poll_pend && wave == 0) {   // have the polled words landed?  Asked, not waited for: the hardware's count of this wave's outstanding VMEM operations
                                                // (IB_STS.vm_cnt; it includes what `issued` does not count, so "landed" can only be late, never early)
                    unsigned ib;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS)" : "=s"(ib));
                    const int out = (int)((ib & 15u) | (((ib >> 22) & 3u) << 4));
                    if (out <= issued - poll_mark) {
                        poll_pend = false;
                        if (need) {   // decide for everybody
                            const unsigned v = ((const unsigned*)(smem + C::POLL_OFF))[pslot * 16 + (lane < 9 ? lane : 0)];
                            const bool rdy = __builtin_amdgcn_ballot_w64((int)(v - target) < 0) == 0ull;
                            if (lane < 2) ((unsigned*)(smem + C::DEC_OFF))[(g & 1) * 2 + lane] = lane == 0 ? (rdy ? target : target - 1u) : (unsigned)pslot;
                            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the decision is in LDS before the barrier
                        }
                    }
                }
                __builtin_amdgcn_s_barrier();
                if (do_pub) flag_store();
#pragma unroll
                for (int k = 7; k > 0; --k) mk[k] = mk[k - 1];
                mk[0] = issued;
                const unsigned long long t2 = IS_T();
                IS_ACC(d.kind == 1 ? 0 : 1, t1 - t0);
                IS_ACC(2, t2 - t1);
                IS_ACC(d.kind == 1 ? 12 : 13, t1b - t1);   // (of 2: the counted vmcnt wait alone, by step kind)
                IS_ACC(d.kind == 1 ? 14 : 15, t2 - t1b);   // (of 2: decision + barrier)
                IS_ACC(d.kind == 1 ? 8 : 9, 1ull);
                if constexpr (MUST) {
                    if (!arr_issued) {   // not there where it is needed: spin, then request and wait for it
                        block_until(cur, target);
                        request_group(4 + 2 * (GK - 1));
                        const unsigned long long t3 = IS_T();
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                        __builtin_amdgcn_s_barrier();
                        IS_ACC(3, IS_T() - t3);
                    }
                }
                // ---- epilogue of the conv this step completed
                constexpr int EU = is_epi_after(TT);
                if constexpr (EU == 0) {   // conv5's accumulators: (beta1 / alpha) * x (see above)
                    const unsigned long long t3 = IS_T();
                    const float c1 = r1pre ? p5.beta1 / p5.alpha : 1.f;
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            A5[mi][nt] *= c1;
                            asm volatile("" : "+v"(A5[mi][nt]));   // (here, not sunk behind the next step's DMA requests: the wait for the residual would drain them)
                        }
                    IS_ACC(7, IS_T() - t3);
                }
                if constexpr (EU >= 0) {
                    const unsigned long long t4 = IS_T();
                    const dasr_conv_params& p = layers[L0 + EU];
                    char* bl = smem + C::BIAS_OFF + par * 1280 + EU * 256;
                    int pmark = issued, nlate = 0;
                    if constexpr (EU == 4) {
                        if (has_next) {   // x of the next item (+ its biases) goes out in front of the conv5 epilogue when its neighbours are there already (the usual case: another tile's
                                          // block, finished an item ago); else behind it (a workgroup with one tile waits for its own conv5 there)
                            if (rn > 0 && !next_rdy) {   // (wave 0's decision at the end of this step)
                                const unsigned* dw = (const unsigned*)(smem + C::DEC_OFF) + (g & 1) * 2;
                                next_rdy = __builtin_amdgcn_readfirstlane(dw[0]) == f0n + (unsigned)L0n && __builtin_amdgcn_readfirstlane(dw[1]) == (unsigned)sn;
                            }
                        }
                        const bool two = p.res2.p != nullptr && !(IS_ABL & 1), sh = p.out_bf16.p != nullptr;
                        constexpr int E0 = BWD ? 160 : 161;   // alpha, fp32 out (+ bias forward); + 16 second residual, + 64 the 16-bit shadow
                        // BORDER FIRST (IS_BORDER): the neighbours read only the border pixels of the 16-bit planes, and the flag waits for the acknowledgement of what it covers:
                        // pass 1 stores those pixels alone (a few KB per tile), the flag's bookkeeping is taken behind it, pass 2 stores the rest (and the fp32 stream)
                        if constexpr (IS_ABL & 1) {
                            asm volatile("" ::"v"(A5[0][0]), "v"(A5[0][NT - 1]), "v"(A5[1][0]), "v"(A5[1][NT - 1]));
                        } else if (two && sh) {
                            if constexpr (IS_BORDER) conv_epilogue<false, 2, NT, 1, E0 + 80 - 32, F16 ? 1 : 0, false, true, true, true, 1, C::TH>(p, A5, bl, 0.f, tid, 0, cur.n, cur.oy0, cur.ox0);
                            pmark = issued;
                            conv_epilogue<false, 2, NT, 1, E0 + 80, F16 ? 1 : 0, false, true, true, true, IS_BORDER ? 2 : 0, C::TH>(p, A5, bl, 0.f, tid, 0, cur.n, cur.oy0, cur.ox0);
                            nlate = 12 * NT;
                        } else if (sh) {
                            if constexpr (IS_BORDER) conv_epilogue<false, 2, NT, 1, E0 + 64 - 32, F16 ? 1 : 0, false, true, true, true, 1, C::TH>(p, A5, bl, 0.f, tid, 0, cur.n, cur.oy0, cur.ox0);
                            pmark = issued;
                            conv_epilogue<false, 2, NT, 1, E0 + 64, F16 ? 1 : 0, false, true, true, true, IS_BORDER ? 2 : 0, C::TH>(p, A5, bl, 0.f, tid, 0, cur.n, cur.oy0, cur.ox0);
                            nlate = 12 * NT;
                        } else if (two) {   // (no 16-bit output: nothing a neighbour waits for)
                            conv_epilogue<false, 2, NT, 1, E0 + 8 + 16, F16 ? 1 : 0, false, true, true, true>(p, A5, bl, 0.f, tid, 0, cur.n, cur.oy0, cur.ox0);
                            pmark = issued, nlate = 0;
                        } else {
                            conv_epilogue<false, 2, NT, 1, E0 + 8, F16 ? 1 : 0, false, true, true, true>(p, A5, bl, 0.f, tid, 0, cur.n, cur.oy0, cur.ox0);
                            pmark = issued, nlate = 0;
                        }
                    } else {
                        auto epi14 = [&](f32x16 (&A)[1][NT]) {
                            if constexpr (IS_ABL & 1) {
                                asm volatile("" ::"v"(A[0][0]), "v"(A[0][NT - 1]));
                            } else {
                                if constexpr (IS_BORDER) conv_epilogue<false, 1, NT, 1, BWD ? 68 : 67, F16 ? 1 : 0, false, true, true, true, 1, C::TH>(p, A, bl, 0.f, tid, 0, cur.n, cur.oy0, cur.ox0);
                                pmark = issued;
                                if constexpr (BWD && !IS_BORDER) conv_epilogue<false, 1, NT, 1, 68, F16 ? 1 : 0, true, true, true, true>(p, A, bl, 0.f, tid, 0, cur.n, cur.oy0, cur.ox0, &mpre);
                                else conv_epilogue<false, 1, NT, 1, BWD ? 68 : 67, F16 ? 1 : 0, false, true, true, true, IS_BORDER ? 2 : 0, C::TH>(p, A, bl, 0.f, tid, 0, cur.n, cur.oy0, cur.ox0);
                                nlate = 2 * NT;
                            }
                        };
                        if constexpr (EU == 0) epi14(A0);
                        else if constexpr (EU == 1) epi14(A1);
                        else if constexpr (EU == 2) epi14(A2);
                        else epi14(A3);
                    }
                    // the flag of this conv: PUB_DELAY steps from now (an older one that is still pending goes out first)
                    flush_pub();
                    pub_on = true, pub_mark = pmark, pub_due = g + C::PUB_DELAY;
                    pub_late = IS_BORDER ? nlate : 0;   // store instructions of pass 2: issued behind the mark, not covered by the flag
                    pub_off = cur.selfo, pub_val = cur.f0 + (unsigned)(L0 + EU + 1);
                    own_mark = issued;                  // this tile's own planes are complete in memory once everything up to here is acknowledged (request_group waits for it)
                    // ... and only now the LDS-DMA requests this step held back: the weights of step g + 3, the next item's x
                    prefetch_w();
                    if constexpr (EU == 4) {
                        if (has_next && (rn == 0 || next_rdy)) request_next_x(geo_of(sn));
                        if constexpr (!BWD) {
                            if (has_next) {
#pragma unroll
                                for (int l = 0; l < 5; l += NW) {
                                    if (l + wave < 5) {
                                        const dasr_conv_params& pb = layers[L0n + l + wave];
                                        is_dma_bias(make_rsrc(pb.bias), smem + C::BIAS_OFF + (par ^ 1) * 1280 + (l + wave) * 256, (pb.bias != nullptr && lane < 32 * pb.mt) ? (unsigned)lane * 4u : OOB);
                                        ++issued;
                                    }
                                }
                            }
                        }
                    }
                    mk[0] = issued;   // (they count as requests of this step)
                    IS_ACC(EU == 4 ? 6 : 5, IS_T() - t4);
                }
                ++g;
            };
            // the 34 steps, unrolled at compile time
            auto run = [&](auto self, auto tc) -> void {
                constexpr int TT = decltype(tc)::value;
                if constexpr (TT < C::NSTEP) {
                    do_step(tc);
                    self(self, ISInt<TT + 1>{});
                }
            };
            run(run, ISInt<0>{});
            IS_ACC(10, 1ull);
            {
                const ISGeo<C> nxt = geo_of(sn);
                if (has_next && !next_x_issued) {   // the next item's neighbours were not there in front of the conv5 epilogue: wait for them now
                    block_until(nxt, nxt.f0 + (unsigned)L0n);
                    request_next_x(nxt);
                }
                cur = nxt;
            }
        }
    }
    flush_pub();
    __builtin_amdgcn_s_waitcnt(0x0F70);
#ifdef DASR_TRACE
    if (g_trace && tid < 16) g_trace[(size_t)(1 << 20) + (size_t)blockIdx.x * 64 + tid] = ((unsigned long long*)(smem + C::TRC_OFF))[tid];
#endif
}

template <bool F16, bool BWD, int NT, int NW>
int launch_rdb_is(const dasr_conv_params* dev_layers, int nrdb, int tiles_y, int tiles_x, int tpw, unsigned* flags, unsigned* tickets, int* err, hipStream_t s, const char* name, int stagger, int grid) {
    static bool attr_set = false;
    auto kfn = rdb_is_kernel<F16, BWD, NT, NW>;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, ISC::LDS_BYTES));
        attr_set = true;
    }
    DASR_LAUNCH_TAG(name, kfn, dim3((unsigned)grid), dim3(64 * NW), ISC::LDS_BYTES, s, dev_layers, nrdb, tiles_y, tiles_x, tpw, flags, tickets, err, stagger);
    return (int)hipGetLastError();
}
