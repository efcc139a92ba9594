// Micro-benchmark probes (NOT part of libdasr_hip.so): built into libdasr_bench.so by `python -m dasr_amd.build --bench`, declared in
// include/dasr_hip_bench.h, used by bench.py (roofline.peak_at_observed_clock / mfma_only_tflops_by_operand_data) and scripts/micro_*.py.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/dasr_hip_bench.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define DASR_EINVAL (-22)
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return (int)_e;           \
    } while (0)
static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

// MFMA-only micro-benchmark: every SIMD of the chip issues back-to-back v_mfma_f32_32x32x16_bf16 on four independent accumulators.
// What it sustains is the dense bf16 peak of THIS box at the clock the power state allows (spec: 2.5 PFLOP/s at 2.4 GHz).
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, float* sink) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (bf16_t)(0.001f * (float)((threadIdx.x * 7 + j * 13) % 31 - 15));
        b[j] = (bf16_t)(0.002f * (float)((threadIdx.x * 5 + j * 11) % 29 - 14));
    }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float t = 0.f;
    for (int j = 0; j < 16; ++j) t += c0[j] + c1[j] + c2[j] + c3[j];
    if (t == 12345.678f) sink[0] = t;  // keeps the chain live
}

// The same instruction stream with operands that toggle: MODE 0 bf16 / 1 f16 fragments drawn from a per-lane LCG (values in (-1, 1), re-drawn
// every 64 MFMA quads so that the loop stays issue-bound), 2 bf16 all-zero operands.  What differs between the modes is only the switching
// activity in the MFMA datapath, i.e. the clock the power management allows.
template <int MODE>
static __global__ __launch_bounds__(256) void mfma_data_kernel(int iters, float* sink) {
    unsigned st = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int o = 0; o < iters; o += 64) {
        float va[8], vb[8];
        for (int j = 0; j < 8; ++j) {
            st = st * 1664525u + 1013904223u;
            va[j] = MODE == 2 ? 0.f : (float)(int)(st >> 8) * (1.f / 8388608.f) - 1.f;
            st = st * 1664525u + 1013904223u;
            vb[j] = MODE == 2 ? 0.f : (float)(int)(st >> 8) * (1.f / 8388608.f) - 1.f;
        }
        if (MODE == 1) {
            f16x8 a, b;
            for (int j = 0; j < 8; ++j) { a[j] = (_Float16)va[j]; b[j] = (_Float16)vb[j]; }
            for (int i = 0; i < 64; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
            }
        } else {
            bf16x8 a, b;
            for (int j = 0; j < 8; ++j) { a[j] = (bf16_t)va[j]; b[j] = (bf16_t)vb[j]; }
            for (int i = 0; i < 64; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
        }
    }
    float t = 0.f;
    for (int j = 0; j < 16; ++j) t += c0[j] + c1[j] + c2[j] + c3[j];
    if (t == 12345.678f) sink[0] = t;
}

extern "C" int dasr_probe_mfma_data(int32_t iters, int32_t mode, float* tflops_out, void* stream) {
    if (iters < 64 || mode < 0 || mode > 2 || !tflops_out) return DASR_EINVAL;
    iters = iters / 64 * 64;
    float* sink = nullptr;
    HIP_TRY(hipMalloc(&sink, 16));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const int blocks = 256 * 2;
    for (int rep = 0; rep < 2; ++rep) {   // rep 0: warm-up (clock ramp), untimed
        hipEvent_t a = rep ? e0 : nullptr, b = rep ? e1 : nullptr;
        if (mode == 0) hipExtLaunchKernelGGL(mfma_data_kernel<0>, dim3(blocks), dim3(256), 0, as_stream(stream), a, b, 0, iters, sink);
        else if (mode == 1) hipExtLaunchKernelGGL(mfma_data_kernel<1>, dim3(blocks), dim3(256), 0, as_stream(stream), a, b, 0, iters, sink);
        else hipExtLaunchKernelGGL(mfma_data_kernel<2>, dim3(blocks), dim3(256), 0, as_stream(stream), a, b, 0, iters, sink);
    }
    hipError_t e = hipStreamSynchronize(as_stream(stream));
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return (int)e;
    *tflops_out = (float)((double)blocks * 4.0 * (double)iters * 4.0 * 2.0 * 32 * 32 * 16 / ((double)ms * 1e-3) / 1e12);
    return 0;
}

extern "C" int dasr_probe_mfma_peak(int32_t iters, float* tflops_out, void* stream) {
    if (iters <= 0 || !tflops_out) return DASR_EINVAL;
    float* sink = nullptr;
    HIP_TRY(hipMalloc(&sink, 16));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const int blocks = 256 * 2;  // 2 workgroups of 4 waves per CU: 2 waves per SIMD
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), iters / 8 + 1, sink);  // warm-up (clock ramp)
    hipExtLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), e0, e1, 0, iters, sink);
    hipError_t e = hipStreamSynchronize(as_stream(stream));
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return (int)e;
    const double flops = (double)blocks * 4.0 * (double)iters * 4.0 * 2.0 * 32 * 32 * 16;
    *tflops_out = (float)(flops / ((double)ms * 1e-3) / 1e12);
    return 0;
}

// Neighbour-flag synchronisation probe (feasibility of a persistent per-RDB kernel): `blocks` co-resident workgroups, each
// owns a tile of `tile_words` 32-bit words; per stage a workgroup rewrites its tile, publishes flag = stage, waits for its two ring
// neighbours' flags and reads one word per thread of each neighbour's tile.
//   nb_stride 8: the neighbours run on the same XCD (workgroup b is dispatched to XCD b % 8); 1: on other XCDs.
//   scope 0: no synchronisation (cost floor: the stores and loads alone); 1: agent-scope release / acquire fences around the flag (L2 write-back
//   and invalidate: what a kernel boundary does); 2: workgroup-scope fences (wait for the stores; the vector L1 is write-through) + agent-scope
//   relaxed atomics for the flag AND the neighbour reads (sc1: served by the L2) -- coherent only when producer and consumer share an L2.
// err[0] = a wait timed out, err[1] = number of stale neighbour reads.
static __global__ __launch_bounds__(256) void tile_sync_kernel(int stages, int nb_stride, int scope, unsigned* flags, unsigned* tiles, int tile_words,
                                                        unsigned* err) {
    const int b = blockIdx.x, nb = gridDim.x;
    const int left = (b + nb - nb_stride) % nb, right = (b + nb_stride) % nb;
    unsigned* my = tiles + (size_t)b * tile_words;
    const unsigned* lt = tiles + (size_t)left * tile_words + threadIdx.x % tile_words;
    const unsigned* rt = tiles + (size_t)right * tile_words + threadIdx.x % tile_words;
    unsigned bad = 0;
    for (int s = 1; s <= stages; ++s) {
        for (int i = threadIdx.x; i < tile_words; i += 256) my[i] = (unsigned)s;
        if (scope == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else if (scope == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        unsigned l, r;
        if (scope) {
            if (threadIdx.x == 0) __hip_atomic_store(&flags[b], (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < 2) {
                const int n = threadIdx.x ? right : left;
                int spins = 0;
                while (__hip_atomic_load(&flags[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)s) {
                    if (++spins > (1 << 21) || __hip_atomic_load(&err[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(&err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (scope == 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                l = *lt;
                r = *rt;
            } else {
                l = __hip_atomic_load(lt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                r = __hip_atomic_load(rt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (l < (unsigned)s || r < (unsigned)s) ++bad;
        } else {
            l = *lt;
            r = *rt;
            if (l + r == 0xffffffffu) ++bad;   // keeps the loads
        }
    }
    if (bad) atomicAdd(&err[1], bad);
}

extern "C" int dasr_probe_tile_sync(int32_t blocks, int32_t stages, int32_t nb_stride, int32_t scope, int32_t tile_words, float* us_per_stage,
                                    int32_t* timed_out, int32_t* stale_reads, void* stream) {
    if (blocks <= 0 || blocks > 2048 || stages <= 0 || nb_stride <= 0 || scope < 0 || scope > 2 || tile_words <= 0 || !us_per_stage) return DASR_EINVAL;
    unsigned* buf = nullptr;
    const size_t words = (size_t)blocks + (size_t)blocks * tile_words + 2;
    HIP_TRY(hipMalloc(&buf, words * 4));
    hipError_t e = hipMemsetAsync(buf, 0, words * 4, as_stream(stream));
    unsigned *flags = buf, *tiles = buf + blocks, *err = buf + blocks + (size_t)blocks * tile_words;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    hipExtLaunchKernelGGL(tile_sync_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), e0, e1, 0, stages, nb_stride, scope, flags, tiles, tile_words, err);
    if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
    float ms = 0.f;
    unsigned res[2] = {0, 0};
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess) e = hipMemcpy(res, err, 8, hipMemcpyDeviceToHost);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    if (e != hipSuccess) return (int)e;
    *us_per_stage = ms * 1e3f / (float)stages;
    if (timed_out) *timed_out = (int32_t)res[0];
    if (stale_reads) *stale_reads = (int32_t)res[1];
    return 0;
}


// `blocks` workgroups of 256 threads that do nothing but hold their workgroup slots for `micros` microseconds (s_memrealtime: the 100 MHz constant clock),
// asynchronous on `stream`.  Stand-in for a collective's kernels on the communication stream in tests/test_gpu_dp.py: whatever runs next to it finds
// `blocks` slots taken (the chained trunk launches, dasr_conv_chain, need every slot of the device).
__global__ void spin_kernel(long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" int dasr_probe_spin(int32_t blocks, int32_t micros, void* stream) {
    if (blocks <= 0 || blocks > 4096 || micros <= 0 || micros > 2000000) return DASR_EINVAL;
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (long long)micros * 100);
    return (int)hipGetLastError();
}

// Store-path probe (round 6): `blocks` workgroups of 512 threads, each writes `kb` KiB of its own region with 16-byte-per-lane stores (1 KiB per wave
// instruction, mode 0 plain / 1 sc1 write-through), `reps` times back to back with `gap` s_sleep(16) between the bursts; cycles of the burst
// (issue of the first store to vmcnt(0) of the last) are summed per workgroup.  What a CU / an XCD / the chip sustains when every epilogue stores at once.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
static __global__ __launch_bounds__(512) void store_probe_kernel(char* buf0, int kb, int mode, int reps, int gap, unsigned long long* cyc) {
    const bool roam = (mode & 4) != 0;   // mode & 4: every burst goes to a fresh region (reps regions of gridDim.x * kb KiB: past the 256 MB Infinity Cache)
    char* buf = buf0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int per_wave = kb / 8;   // KiB per wave = store instructions per wave
    u32x4_t v = {(unsigned)tid, 1u, 2u, 3u};
    unsigned long long total = 0;
    // mode & 2: the address pattern of rdb_is_kernel's conv5 epilogue instead of a linear region (kb must be 192): workgroup b = tile (b % 32) of image b / 32 of a
    // [N][4 planes][128][128][16] fp32 tensor followed by a [N][12 planes][128][128][16] 16-bit tensor; wave w owns rows 2w, 2w + 1 of the 16 x 32-pixel tile
    const int img = blockIdx.x / 32, tile = blockIdx.x % 32, ty = tile / 4, tx = tile % 4;
    const size_t nimg = (gridDim.x + 31) / 32;
    for (int rep = 0; rep < reps; ++rep) {
        __syncthreads();
        if (roam) buf = buf0 + (size_t)rep * gridDim.x * kb * 1024;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + (size_t)blockIdx.x * kb * 1024), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 0x7fffffff, 0x00020000);
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (mode & 2) {
            for (int nt = 0; nt < 2; ++nt)
                for (int pl = 0; pl < 4; ++pl) {
                    const unsigned pix = (unsigned)((ty * 16 + wave * 2 + nt) * 128 + tx * 32);
                    for (int hf = 0; hf < 2; ++hf) {
                        const unsigned off = (unsigned)(((size_t)(img * 4 + pl) * 16384 + pix + hf * 16) * 64) + lane * 16;
                        if (mode & 1) __builtin_amdgcn_raw_buffer_store_b128(v, rt, off, 0, 16);
                        else __builtin_amdgcn_raw_buffer_store_b128(v, rt, off, 0, 0);
                    }
                    const unsigned off2 = (unsigned)(nimg * 4 * 16384 * 64 + ((size_t)(img * 12 + pl) * 16384 + pix) * 32) + lane * 16;
                    if (mode & 1) __builtin_amdgcn_raw_buffer_store_b128(v, rt, off2, 0, 16);
                    else __builtin_amdgcn_raw_buffer_store_b128(v, rt, off2, 0, 0);
                }
        } else
        for (int i = 0; i < per_wave; ++i) {
            const unsigned off = (unsigned)((wave * per_wave + i) * 1024 + lane * 16);
            if (mode & 1) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        total += __builtin_readcyclecounter() - t0;
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(16);
        v[1] += 1u;
    }
    if (tid == 0) cyc[blockIdx.x] = total;
}

extern "C" int dasr_probe_store(void* buf, int32_t blocks, int32_t kb, int32_t mode, int32_t reps, int32_t gap, unsigned long long* cyc_out, void* stream) {
    if (!buf || blocks <= 0 || kb <= 0 || (kb & 7) || !cyc_out) return DASR_EINVAL;
    hipLaunchKernelGGL(store_probe_kernel, dim3((unsigned)blocks), dim3(512), 0, as_stream(stream), (char*)buf, kb, mode, reps, gap, cyc_out);
    HIP_TRY(hipGetLastError());
    return (int)hipStreamSynchronize(as_stream(stream));
}
