// Weight packing, HBM-bound elementwise kernels, fused Adam and the op-list executor (gfx950).
#include "common.h"
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>

namespace {

// ---------------------------------------------------------------------------------------------------
// fp32 master weights -> bf16 MFMA A-fragment order [mgroup][chunk][tap][mt][lane][8] (+ lo plane)
// ---------------------------------------------------------------------------------------------------
__global__ void pack_kernel(const dasr_pack_desc* __restrict__ descs, int ndesc, long long total,
                            const long long* __restrict__ prefix, const float* __restrict__ params, bf16_t* __restrict__ packed) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    int lo = 0, hi = ndesc - 1;  // find desc: prefix[d] <= gid < prefix[d+1]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (prefix[mid] <= gid) lo = mid; else hi = mid - 1;
    }
    const dasr_pack_desc& D = descs[lo];
    long long q = gid - prefix[lo];
    const int lane = q & 63;
    q >>= 6;
    const int mi = q % D.mt;
    q /= D.mt;
    const int tap = q % D.ntaps;
    q /= D.ntaps;
    const int nchunks = D.cin_pad >> 4;
    const int ckv = q % nchunks;
    const int mg = q / nchunks;
    const int oc = (mg * D.mt + mi) * 32 + (lane & 31);
    // fmt 3 / 4 (split 16-bit tensors, dasr_conv_params::in_wrap): cin_pad counts 3K virtual chunks [hi | hi | lo] of K real ones
    const int kreal = D.fmt >= 3 ? nchunks / 3 : nchunks;
    const int part = ckv / kreal, ck = ckv - part * kreal;
    const int c0 = ck * 16 + 8 * (lane >> 5);
    bf16x8 vh, vl;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int cin = c0 + e;
        float v = 0.f;
        if (oc < D.cout) {
            for (int s = 0; s < D.nseg; ++s) {
                const dasr_pack_seg& S = D.seg[s];
                const int ci = cin - S.cin_start;
                if (ci >= 0 && ci < S.cin_len) {
                    const int st = D.tapmap[tap];
                    const unsigned msk = tap < 16 ? D.tapmask[tap] : 0u;
                    const bool okc = !S.transpose || ci < S.src_cout;
                    const long long base = !S.transpose ? S.src_off + ((long long)oc * S.src_cin + S.src_c0 + ci) * D.src_ntaps
                                                        : S.src_off + ((long long)ci * S.src_cin + S.src_c0 + oc) * D.src_ntaps;
                    if (msk && okc) {  // sum of source taps (fixed order: deterministic)
                        float acc = 0.f;
                        for (int k = 0; k < D.src_ntaps && k < 16; ++k)
                            if (msk & (1u << k)) acc += params[base + k];
                        v = acc;
                    } else if (st >= 0 && okc) {
                        v = params[base + st];
                    }
                }
            }
        }
        bf16_t h, l;
        split_bf16(v, h, l);
        if (D.fmt == 1) h = __builtin_bit_cast(bf16_t, (f16_t)v);   // f16 bit pattern (prec 2 convs)
        if (D.fmt == 2 || D.fmt == 3) split_f16(v, h, l);             // f16 hi + lo planes (prec 4 convs) / virtual chunks (fmt 3)
        if (D.fmt >= 3 && part == 2) h = l;                           // third group of virtual chunks: the remainders
        vh[e] = h;
        vl[e] = l;
    }
    const long long o = D.dst_off + (gid - prefix[lo]) * 8;
    *(bf16x8*)(packed + o) = vh;
    if (D.lo_off) *(bf16x8*)(packed + o + D.lo_off) = vl;
}

// ---------------------------------------------------------------------------------------------------
__global__ void nchw_to_blocked_kernel(const float* __restrict__ src, int N, int C, int H, int W, dasr_tensor df, dasr_tensor db) {
    const int ncb = (C + 15) >> 4;
    const long long total = (long long)N * ncb * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = i % W;
    long long t = i / W;
    const int y = t % H;
    t /= H;
    const int cb = t % ncb;
    const int n = t / ncb;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = cb * 16 + j;
        v[j] = c < C ? src[(((long long)n * C + c) * H + y) * W + x] : 0.f;
    }
    const size_t po = ((size_t)y * W + x) * 16;
    if (df.p) {
        float* d = (float*)df.p + (size_t)n * df.n_stride + (size_t)cb * df.cb_stride + po;
#pragma unroll
        for (int j = 0; j < 4; ++j) ((f32x4*)d)[j] = f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
    }
    if (db.p) {
        bf16_t* d = (bf16_t*)db.p + (size_t)n * db.n_stride + (size_t)cb * db.cb_stride + po;
        bf16x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = (bf16_t)v[j];
            b[j] = (bf16_t)v[8 + j];
        }
        ((bf16x8*)d)[0] = a;
        ((bf16x8*)d)[1] = b;
    }
}

__global__ void blocked_to_nchw_kernel(dasr_tensor s, int N, int C, int H, int W, float* __restrict__ dst) {
    const long long total = (long long)N * C * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = i % W;
    long long t = i / W;
    const int y = t % H;
    t /= H;
    const int c = t % C;
    const int n = t / C;
    dst[i] = ((const float*)s.p)[(size_t)n * s.n_stride + (size_t)(c >> 4) * s.cb_stride + ((size_t)y * W + x) * 16 + (c & 15)];
}

// one thread per pixel (plane 0 only: C <= 16)
__global__ void l1_loss_kernel(dasr_tensor sr, const float* __restrict__ hr, const float* __restrict__ wm, int N, int C, int H, int W,
                               float coef, float* loss_acc, dasr_tensor grad, int accumulate, float gscale, dasr_red rs) {
    const long long total = (long long)N * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float part = 0.f;
    if (i < total) {
        const int x = i % W;
        long long t = i / W;
        const int y = t % H;
        const int n = t / H;
        const size_t po = ((size_t)y * W + x) * 16;
        const float* s = (const float*)sr.p + (size_t)n * sr.n_stride + po;
        const float wgt = wm ? wm[((long long)n * H + y) * W + x] : 1.f;
        float g[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) g[j] = 0.f;
        for (int c = 0; c < C; ++c) {
            const float d = s[c] - hr[(((long long)n * C + c) * H + y) * W + x];
            if (accumulate & 2) {   // nn.MSELoss (pixel_criterion 'l2', SR_model.py:33-36 / DASR_model.py:79-80)
                part += wgt * d * d;
                g[c] = 2.f * coef * wgt * d;
            } else {
                part += wgt * fabsf(d);
                g[c] = coef * wgt * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
            }
        }
        if (grad.p && (accumulate & 4)) {
            // f16 gradient tensor (round 6; the generator's f16 HR tail): f16(gscale * g) straight from here -- no padded fp32 image (64 B per pixel for 12 B of
            // payload) and no conversion pass behind it.  Only the channel groups that hold a real channel are written: the tensor is zero beyond them and stays so.
            f16_t* gp = (f16_t*)grad.p + (size_t)n * grad.n_stride + po;
            for (int j = 0; j < (C + 3) / 4; ++j) {
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (f16_t)(g[4 * j + e] * gscale);
                ((f16x4*)gp)[j] = o;
            }
        } else if (grad.p) {
            float* gp = (float*)grad.p + (size_t)n * grad.n_stride + po;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 o = {g[4 * j], g[4 * j + 1], g[4 * j + 2], g[4 * j + 3]};
                if (accumulate & 1) o += ((const f32x4*)gp)[j];
                ((f32x4*)gp)[j] = o;
            }
        }
    }
    part = wave_sum(part);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    const float v[1] = {red[0] + red[1] + red[2] + red[3]};
    float* const acc[1] = {loss_acc};
    const float cf[1] = {coef};
    grid_sum_commit<1>(rs, v, acc, cf);   // loss_acc += coef * (sum over the grid), summed in a fixed order
}

// thread per (n, cb, y, x) at the LOW resolution
__global__ void downsum2x_kernel(dasr_tensor src, int N, int C, int H, int W, dasr_tensor mask, int mask_f32, float slope,
                                 dasr_tensor df, dasr_tensor db) {
    const int ncb = (C + 15) >> 4;
    const long long total = (long long)N * ncb * H * W * 4;  // 4 threads per pixel, 4 channels each
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const int q = gi & 3;
    const long long i = gi >> 2;
    const int x = i % W;
    long long t = i / W;
    const int y = t % H;
    t /= H;
    const int cb = t % ncb;
    const int n = t / ncb;
    const float* s = (const float*)src.p + (size_t)n * src.n_stride + (size_t)cb * src.cb_stride + q * 4;
    const int W2 = 2 * W;
    f32x4 a = *(const f32x4*)(s + ((size_t)(2 * y) * W2 + 2 * x) * 16);
    a += *(const f32x4*)(s + ((size_t)(2 * y) * W2 + 2 * x + 1) * 16);
    a += *(const f32x4*)(s + ((size_t)(2 * y + 1) * W2 + 2 * x) * 16);
    a += *(const f32x4*)(s + ((size_t)(2 * y + 1) * W2 + 2 * x + 1) * 16);
    const size_t po = ((size_t)y * W + x) * 16 + q * 4;
    if (mask.p) {
        const size_t mo = (size_t)n * mask.n_stride + (size_t)cb * mask.cb_stride + po;
        float m[4];
        if (mask_f32) {
            const f32x4 mv = *(const f32x4*)((const float*)mask.p + mo);
            for (int j = 0; j < 4; ++j) m[j] = mv[j];
        } else {
            const bf16x4 mv = *(const bf16x4*)((const bf16_t*)mask.p + mo);
            for (int j = 0; j < 4; ++j) m[j] = (float)mv[j];
        }
        for (int j = 0; j < 4; ++j) a[j] = m[j] > 0.f ? a[j] : a[j] * slope;
    }
    if (df.p) *(f32x4*)((float*)df.p + (size_t)n * df.n_stride + (size_t)cb * df.cb_stride + po) = a;
    if (db.p) {
        bf16x4 o;
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)a[j];
        *(bf16x4*)((bf16_t*)db.p + (size_t)n * db.n_stride + (size_t)cb * db.cb_stride + po) = o;
    }
}

__global__ void axpby_kernel(dasr_tensor x, float a, dasr_tensor z, float b, int N, int C, int H, int W, dasr_tensor of,
                             dasr_tensor ob, float gamma, dasr_tensor mask, float slope, const float* slope_ptr) {
    const int ncb = (C + 15) >> 4;
    const long long per_plane = (long long)H * W * 4;  // f32x4 pieces
    const long long total = (long long)N * ncb * per_plane;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const long long e = (gi % per_plane) * 4;
    long long t = gi / per_plane;
    const int cb = t % ncb;
    const int n = t / ncb;
    f32x4 v = *(const f32x4*)((const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + e) * a;
    if (z.p) v += *(const f32x4*)((const float*)z.p + (size_t)n * z.n_stride + (size_t)cb * z.cb_stride + e) * b;
    if (mask.p) {  // (Leaky/P)ReLU' of the f32 activation `mask`, applied after the sum
        const float sl = slope_ptr ? *slope_ptr : slope;
        const f32x4 mv = *(const f32x4*)((const float*)mask.p + (size_t)n * mask.n_stride + (size_t)cb * mask.cb_stride + e);
        for (int j = 0; j < 4; ++j) v[j] = mv[j] > 0.f ? v[j] : v[j] * sl;
    }
    if (of.p) *(f32x4*)((float*)of.p + (size_t)n * of.n_stride + (size_t)cb * of.cb_stride + e) = v;
    if (ob.p) {
        bf16x4 o;
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(v[j] * gamma);
        *(bf16x4*)((bf16_t*)ob.p + (size_t)n * ob.n_stride + (size_t)cb * ob.cb_stride + e) = o;
    }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long long n, float step_size, float beta1, float beta2, float eps, float wd, float inv_sqrt_bc2, int* __restrict__ nonfinite,
                            const int* __restrict__ gate) {
    if (gate && *gate != 0) return;   // (wave-uniform scalar load) the gradients of this step are not valid -- e.g. a chained trunk launch flagged a broken
                                      // neighbour wait: weights and moments stay untouched until the host has looked at the word
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i];
        bad |= !(fabsf(gi) <= 3.0e38f);   // inf or NaN (e.g. an overflow of the f16-stored, pre-scaled HR-tail gradients): reported, not masked
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = pi - step_size * (mi / denom);
    }
    if (nonfinite && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(nonfinite, 1);   // one atomic per wave that saw one; normally none
}

__global__ void add_flat_kernel(float* __restrict__ y, const float* __restrict__ x, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] += x[i];
}

__global__ void fill_kernel(float* p, long long n, float v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

inline unsigned blocks_for(long long total, int bs = 256) { return (unsigned)((total + bs - 1) / bs); }

}  // namespace

extern "C" int dasr_pack_weights(const dasr_pack_desc* descs_dev, int32_t ndesc, int64_t total_pieces, const int64_t* piece_prefix_dev,
                                 const float* params_flat, void* packed, void* stream) {
    if (ndesc <= 0 || total_pieces <= 0) return DASR_EINVAL;
    DASR_LAUNCH(pack_kernel, dim3(blocks_for(total_pieces)), dim3(256), 0, as_stream(stream), descs_dev, ndesc,
                       (long long)total_pieces, (const long long*)piece_prefix_dev, params_flat, (bf16_t*)packed);
    return (int)hipGetLastError();
}

extern "C" int dasr_nchw_to_blocked(const float* src, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor dst_f32,
                                    dasr_tensor dst_bf16, void* stream) {
    const long long total = (long long)N * ((C + 15) / 16) * H * W;
    if (total <= 0) return DASR_EINVAL;
    DASR_LAUNCH(nchw_to_blocked_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), src, N, C, H, W, dst_f32, dst_bf16);
    return (int)hipGetLastError();
}

extern "C" int dasr_blocked_to_nchw(dasr_tensor src, int32_t N, int32_t C, int32_t H, int32_t W, float* dst, void* stream) {
    const long long total = (long long)N * C * H * W;
    if (total <= 0) return DASR_EINVAL;
    DASR_LAUNCH(blocked_to_nchw_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), src, N, C, H, W, dst);
    return (int)hipGetLastError();
}

extern "C" int dasr_l1_loss(dasr_tensor sr, const float* hr_nchw, const float* weight_map, int32_t N, int32_t C, int32_t H, int32_t W,
                            float coef, float* loss_acc, dasr_tensor grad, int32_t accumulate, float grad_scale, void* stream) {
    const long long total = (long long)N * H * W;
    if (total <= 0 || C > 16) return DASR_EINVAL;
    if ((accumulate & 4) && (accumulate & 1)) return DASR_EINVAL;   // the f16 gradient form does not accumulate
    const dasr_red rs = dasr_red_scratch(loss_acc, as_stream(stream), blocks_for(total), 1);
    if (loss_acc && !rs.part) return dasr_red_error();
    DASR_LAUNCH(l1_loss_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), sr, hr_nchw, weight_map, N, C, H, W, coef,
                       loss_acc, grad, accumulate, grad_scale != 0.f ? grad_scale : 1.f, rs);
    return (int)hipGetLastError();
}

extern "C" int dasr_downsum2x(dasr_tensor src, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor mask, int32_t mask_f32,
                              float slope, dasr_tensor dst_f32, dasr_tensor dst_bf16, void* stream) {
    const long long total = (long long)N * ((C + 15) / 16) * H * W * 4;
    if (total <= 0) return DASR_EINVAL;
    DASR_LAUNCH(downsum2x_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), src, N, C, H, W, mask, mask_f32, slope,
                       dst_f32, dst_bf16);
    return (int)hipGetLastError();
}

// ---- f16-storage helpers of the generator's HR tail ---------------------------------------------------------------------------
// y16 = f16(scale * x) over a blocked f32 tensor (dL/dSR -> pre-scaled f16 gradient)
__global__ void cvt_f16_kernel(dasr_tensor x, int N, int C, int H, int W, float scale, dasr_tensor y) {
    const int ncb = (C + 15) >> 4;
    const long long per_plane = (long long)H * W * 4, total = (long long)N * ncb * per_plane;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const long long e = (gi % per_plane) * 4;
    long long t = gi / per_plane;
    const int cb = t % ncb, n = t / ncb;
    const f32x4 v = *(const f32x4*)((const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + e);
    f16x4 o;
    for (int j = 0; j < 4; ++j) o[j] = (f16_t)(v[j] * scale);
    *(f16x4*)((f16_t*)y.p + (size_t)n * y.n_stride + (size_t)cb * y.cb_stride + e) = o;
}

// y (f32) = x - f16(scale * x) / scale: what an f16 operand rounding leaves behind.  The weight gradients of the BatchNorm discriminators run as
// g.x + g.x_lo + g_lo.x on the f16 MFMA (three parts of one launch, 22-bit operands) -- the residual tensors are these.
__global__ void f16_residual_kernel(dasr_tensor x, int N, int C, int H, int W, float scale, dasr_tensor y) {
    const int ncb = (C + 15) >> 4;
    const long long per_plane = (long long)H * W * 4, total = (long long)N * ncb * per_plane;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const long long e = (gi % per_plane) * 4;
    long long t = gi / per_plane;
    const int cb = t % ncb, n = t / ncb;
    const f32x4 v = *(const f32x4*)((const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + e);
    const float inv = 1.f / scale;
    f32x4 o;
    for (int j = 0; j < 4; ++j) o[j] = v[j] - (float)(f16_t)(v[j] * scale) * inv;
    *(f32x4*)((float*)y.p + (size_t)n * y.n_stride + (size_t)cb * y.cb_stride + e) = o;
}

// split 16-bit copy of scale * x: hi = round16(scale * x) into the first ncb planes of y, lo = round16(scale * x - hi) into the next ncb
template <typename T>
__global__ void cvt_split16_kernel(dasr_tensor x, int N, int C, int H, int W, float scale, dasr_tensor y) {
    const int ncb = (C + 15) >> 4;
    const long long per_plane = (long long)H * W * 4, total = (long long)N * ncb * per_plane;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const long long e = (gi % per_plane) * 4;
    long long t = gi / per_plane;
    const int cb = t % ncb, n = t / ncb;
    const f32x4 v = *(const f32x4*)((const float*)x.p + (size_t)n * x.n_stride + (size_t)cb * x.cb_stride + e);
    T* yp = (T*)y.p + (size_t)n * y.n_stride + (size_t)cb * y.cb_stride + e;
    const size_t lo = (size_t)ncb * y.cb_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float vv = v[j] * scale;
        const T h = (T)vv;
        yp[j] = h;
        yp[lo + j] = (T)(vv - (float)h);
    }
}

// backward of nn.Upsample(nearest, 2) on f16 tensors: dst[y][x] = out_scale * (LeakyReLU' mask) * sum of the 2x2 block of src;
// dst is f16 (df.p == NULL) or f32.  thread per (n, cb, y, x, 4-channel quad) at the LOW resolution
__global__ void downsum2x_f16_kernel(dasr_tensor src, int N, int C, int H, int W, dasr_tensor mask, float slope, float out_scale, dasr_tensor df,
                                     dasr_tensor dh) {
    const int ncb = (C + 15) >> 4;
    const long long total = (long long)N * ncb * H * W * 4;
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    const int q = gi & 3;
    const long long i = gi >> 2;
    const int x = i % W;
    long long t = i / W;
    const int y = t % H;
    t /= H;
    const int cb = t % ncb, n = t / ncb;
    const f16_t* s = (const f16_t*)src.p + (size_t)n * src.n_stride + (size_t)cb * src.cb_stride + q * 4;
    const int W2 = 2 * W;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            const f16x4 v = *(const f16x4*)(s + ((size_t)(2 * y + dy) * W2 + 2 * x + dx) * 16);
            for (int j = 0; j < 4; ++j) a[j] += (float)v[j];
        }
    const size_t po = ((size_t)y * W + x) * 16 + q * 4;
    if (mask.p) {
        const f16x4 mv = *(const f16x4*)((const f16_t*)mask.p + (size_t)n * mask.n_stride + (size_t)cb * mask.cb_stride + po);
        for (int j = 0; j < 4; ++j) a[j] = (float)mv[j] > 0.f ? a[j] : a[j] * slope;
    }
    for (int j = 0; j < 4; ++j) a[j] *= out_scale;
    if (df.p) *(f32x4*)((float*)df.p + (size_t)n * df.n_stride + (size_t)cb * df.cb_stride + po) = f32x4{a[0], a[1], a[2], a[3]};
    if (dh.p) {
        f16x4 o;
        for (int j = 0; j < 4; ++j) o[j] = (f16_t)a[j];
        *(f16x4*)((f16_t*)dh.p + (size_t)n * dh.n_stride + (size_t)cb * dh.cb_stride + po) = o;
    }
}

// nn.PixelShuffle(2) of pixelshuffle_block (block.py:838-851) on f16 tensors: dst[n][c][2y+dy][2x+dx] = src[n][4c + 2dy + dx][y][x].
// One thread per (n, 16-channel input plane p, y, x): the plane holds output channels 4p..4p+3 for the four (dy, dx).
__global__ void pixel_shuffle_f16_kernel(dasr_tensor src, int N, int C4, int H, int W, dasr_tensor dst) {
    const int ncb = C4 >> 4;
    const long long total = (long long)N * ncb * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = i % W;
    long long t = i / W;
    const int y = t % H;
    t /= H;
    const int p = t % ncb, n = t / ncb;
    const f16_t* s = (const f16_t*)src.p + (size_t)n * src.n_stride + (size_t)p * src.cb_stride + ((size_t)y * W + x) * 16;
    const f16x8 lo = *(const f16x8*)s, hi = *(const f16x8*)(s + 8);
    f16_t v[16];
    for (int j = 0; j < 8; ++j) { v[j] = lo[j]; v[8 + j] = hi[j]; }
    f16_t* d = (f16_t*)dst.p + (size_t)n * dst.n_stride + (size_t)(p >> 2) * dst.cb_stride + 4 * (p & 3);
    const int W2 = 2 * W;
    for (int k = 0; k < 4; ++k) {   // k = 2 dy + dx; channels 4p + cc take src channel 4 cc + k of the plane
        const f16x4 o = {v[k], v[4 + k], v[8 + k], v[12 + k]};
        *(f16x4*)(d + ((size_t)(2 * y + (k >> 1)) * W2 + 2 * x + (k & 1)) * 16) = o;
    }
}

// its adjoint with the LeakyReLU' of the (already activated) shuffle input folded in: gdst[n][4c+k][y][x] = m * gsrc[n][c][2y+dy][2x+dx],
// m = mask > 0 ? 1 : slope
__global__ void pixel_unshuffle_f16_kernel(dasr_tensor gsrc, dasr_tensor mask, float slope, int N, int C4, int H, int W, dasr_tensor gdst) {
    const int ncb = C4 >> 4;
    const long long total = (long long)N * ncb * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = i % W;
    long long t = i / W;
    const int y = t % H;
    t /= H;
    const int p = t % ncb, n = t / ncb;
    const f16_t* g = (const f16_t*)gsrc.p + (size_t)n * gsrc.n_stride + (size_t)(p >> 2) * gsrc.cb_stride + 4 * (p & 3);
    const int W2 = 2 * W;
    f16_t v[16];
    for (int k = 0; k < 4; ++k) {
        const f16x4 q = *(const f16x4*)(g + ((size_t)(2 * y + (k >> 1)) * W2 + 2 * x + (k & 1)) * 16);
        for (int cc = 0; cc < 4; ++cc) v[4 * cc + k] = q[cc];
    }
    const size_t po = (size_t)n * gdst.n_stride + (size_t)p * gdst.cb_stride + ((size_t)y * W + x) * 16;
    if (mask.p) {
        const f16_t* m = (const f16_t*)mask.p + (size_t)n * mask.n_stride + (size_t)p * mask.cb_stride + ((size_t)y * W + x) * 16;
        for (int j = 0; j < 16; ++j) v[j] = (float)m[j] > 0.f ? v[j] : (f16_t)((float)v[j] * slope);
    }
    f16x8 lo, hi;
    for (int j = 0; j < 8; ++j) { lo[j] = v[j]; hi[j] = v[8 + j]; }
    *(f16x8*)((f16_t*)gdst.p + po) = lo;
    *(f16x8*)((f16_t*)gdst.p + po + 8) = hi;
}

extern "C" int dasr_pixel_shuffle_f16(dasr_tensor src, int32_t N, int32_t C4, int32_t H, int32_t W, dasr_tensor dst, void* stream) {
    const long long total = (long long)N * (C4 / 16) * H * W;
    if (total <= 0 || (C4 & 63) || !src.p || !dst.p) return DASR_EINVAL;
    DASR_LAUNCH(pixel_shuffle_f16_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), src, N, C4, H, W, dst);
    return (int)hipGetLastError();
}

extern "C" int dasr_pixel_unshuffle_f16(dasr_tensor gsrc, dasr_tensor mask, float slope, int32_t N, int32_t C4, int32_t H, int32_t W, dasr_tensor gdst,
                                        void* stream) {
    const long long total = (long long)N * (C4 / 16) * H * W;
    if (total <= 0 || (C4 & 63) || !gsrc.p || !gdst.p) return DASR_EINVAL;
    DASR_LAUNCH(pixel_unshuffle_f16_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), gsrc, mask, slope, N, C4, H, W, gdst);
    return (int)hipGetLastError();
}

extern "C" int dasr_cvt_f16(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, float scale, dasr_tensor y, void* stream) {
    const long long total = (long long)N * ((C + 15) / 16) * H * W * 4;
    if (total <= 0 || !x.p || !y.p) return DASR_EINVAL;
    DASR_LAUNCH(cvt_f16_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), x, N, C, H, W, scale, y);
    return (int)hipGetLastError();
}

extern "C" int dasr_cvt_split16(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, float scale, dasr_tensor y, int32_t f16, void* stream) {
    const long long total = (long long)N * ((C + 15) / 16) * H * W * 4;
    if (total <= 0 || !x.p || !y.p) return DASR_EINVAL;
    if (f16) DASR_LAUNCH(cvt_split16_kernel<f16_t>, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), x, N, C, H, W, scale, y);
    else DASR_LAUNCH(cvt_split16_kernel<bf16_t>, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), x, N, C, H, W, scale, y);
    return (int)hipGetLastError();
}

extern "C" int dasr_f16_residual(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, float scale, dasr_tensor y, void* stream) {
    const long long total = (long long)N * ((C + 15) / 16) * H * W * 4;
    if (total <= 0 || !x.p || !y.p || !(scale > 0.f)) return DASR_EINVAL;
    DASR_LAUNCH(f16_residual_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), x, N, C, H, W, scale, y);
    return (int)hipGetLastError();
}

extern "C" int dasr_downsum2x_f16(dasr_tensor src, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor mask, float slope, float out_scale,
                                  dasr_tensor dst_f32, dasr_tensor dst_f16, void* stream) {
    const long long total = (long long)N * ((C + 15) / 16) * H * W * 4;
    if (total <= 0 || !src.p) return DASR_EINVAL;
    DASR_LAUNCH(downsum2x_f16_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), src, N, C, H, W, mask, slope, out_scale, dst_f32, dst_f16);
    return (int)hipGetLastError();
}

extern "C" int dasr_axpby(dasr_tensor x, float a, dasr_tensor z, float b, int32_t N, int32_t C, int32_t H, int32_t W,
                          dasr_tensor out_f32, dasr_tensor out_bf16, float gamma, dasr_tensor mask, float slope, const float* slope_ptr,
                          void* stream) {
    const long long total = (long long)N * ((C + 15) / 16) * H * W * 4;
    if (total <= 0) return DASR_EINVAL;
    DASR_LAUNCH(axpby_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), x, a, z, b, N, C, H, W, out_f32, out_bf16, gamma, mask, slope, slope_ptr);
    return (int)hipGetLastError();
}

extern "C" int dasr_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int32_t step, int32_t* nonfinite_flag, const int32_t* gate_flag, void* stream) {
    if (n <= 0 || step <= 0) return DASR_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    DASR_LAUNCH(adam_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), p, g, m, v, (long long)n, step_size, beta1, beta2, eps,
                       weight_decay, inv_sqrt_bc2, nonfinite_flag, gate_flag);
    return (int)hipGetLastError();
}

extern "C" int dasr_fill_f32(float* p, int64_t n, float value, void* stream) {
    if (n <= 0) return DASR_EINVAL;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    DASR_LAUNCH(fill_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), p, (long long)n, value);
    return (int)hipGetLastError();
}

extern "C" int dasr_add_flat(float* y, const float* x, int64_t n, void* stream) {
    if (n <= 0) return DASR_EINVAL;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    DASR_LAUNCH(add_flat_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), y, x, (long long)n);
    return (int)hipGetLastError();
}

// batch assembly on resident images: crop (+ optional bilinear resize of the source, cv2.INTER_LINEAR convention: half-pixel
// centres, edge clamp) + hflip / vflip / transpose; one thread per output element
__global__ void gather_crops_kernel(const dasr_crop_desc* __restrict__ descs, int n, int C, int size, float* __restrict__ dst) {
    const long long total = (long long)n * C * size * size;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = i % size;
    long long t = i / size;
    const int y = t % size;
    t /= size;
    const int c = t % C;
    const int k = t / C;
    const dasr_crop_desc D = descs[k];
    int ci = y, cj = x;
    if (D.flags & 4) { ci = x; cj = y; }           // transpose was applied last: undo first
    if (D.flags & 2) ci = size - 1 - ci;           // vflip
    if (D.flags & 1) cj = size - 1 - cj;           // hflip
    const int vy = D.y0 + ci, vx = D.x0 + cj;
    float v = 0.f;
    if (c < D.C && vy >= 0 && vy < D.vH && vx >= 0 && vx < D.vW) {
        const float* s = D.src + (size_t)c * D.H * D.W;
        if (D.vH == D.H && D.vW == D.W) {
            v = s[(size_t)vy * D.W + vx];
        } else {
            float fy = ((float)vy + 0.5f) * ((float)D.H / (float)D.vH) - 0.5f, fx = ((float)vx + 0.5f) * ((float)D.W / (float)D.vW) - 0.5f;
            int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
            const float wy = fy - (float)y0, wx = fx - (float)x0;
            const int y1 = min(max(y0 + 1, 0), D.H - 1), x1 = min(max(x0 + 1, 0), D.W - 1);
            y0 = min(max(y0, 0), D.H - 1);
            x0 = min(max(x0, 0), D.W - 1);
            v = (1.f - wy) * ((1.f - wx) * s[(size_t)y0 * D.W + x0] + wx * s[(size_t)y0 * D.W + x1]) +
                wy * ((1.f - wx) * s[(size_t)y1 * D.W + x0] + wx * s[(size_t)y1 * D.W + x1]);
        }
    }
    dst[i] = v;
}

extern "C" int dasr_gather_crops(const dasr_crop_desc* descs_dev, int32_t n, int32_t C, int32_t size, float* dst, void* stream) {
    const long long total = (long long)n * C * size * size;
    if (total <= 0 || !descs_dev || !dst) return DASR_EINVAL;
    DASR_LAUNCH(gather_crops_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), descs_dev, n, C, size, dst);
    return (int)hipGetLastError();
}

extern "C" void* dasr_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return (void*)e;
}

extern "C" int dasr_event_destroy(void* ev) { return ev ? (int)hipEventDestroy((hipEvent_t)ev) : 0; }

extern "C" int dasr_abi_version(void) { return DASR_ABI_VERSION; }

static thread_local int g_last_failed_op = -1;   // per enqueue thread; dasr_run_ops_mt copies the failing list's value to its caller
extern "C" int dasr_last_failed_op(void) { return g_last_failed_op; }

// ---- scratch rows of the deterministic grid sums (grid_sum_commit, common.h) -------------------------------------------------
// One row per (device, accumulator address, stream): launches that use the same row are ordered by their stream, so a row is never shared by two
// kernels in flight; rows are allocated on first use (the warm-up steps) and only ever grow (the old allocation of a grown row stays alive: a kernel
// enqueued earlier may still write it).  [ticket word | pad to 256 B | K x nblocks partials]
namespace {
struct RedRow {
    char* base;
    size_t floats;
};
std::mutex g_red_mu;
std::map<std::tuple<int, const void*, hipStream_t>, RedRow> g_red_rows;
thread_local bool g_red_capture_refused = false;
}  // namespace

// DASR_ECAPTURE if the last dasr_red_scratch of this thread refused to allocate under stream capture, else DASR_EINVAL (what a launcher without a row returns)
int dasr_red_error() {
    const bool c = g_red_capture_refused;
    g_red_capture_refused = false;
    return c ? DASR_ECAPTURE : DASR_EINVAL;
}

// Frees every scratch row (the rows are keyed by (device, accumulator address, stream) and otherwise live as long as the process: a process that builds and drops many
// models -- the test suite -- calls this between them).  The caller guarantees that no launch that uses a row is in flight.
extern "C" int dasr_red_release(void) {
    std::lock_guard<std::mutex> lock(g_red_mu);
    for (auto& kv : g_red_rows)
        if (kv.second.base) (void)hipFree(kv.second.base);
    g_red_rows.clear();
    return 0;
}

dasr_red dasr_red_scratch(const void* key_acc, hipStream_t s, unsigned nblocks, int k) {
    dasr_red r = {nullptr, nullptr};
    if (!key_acc || nblocks == 0 || k <= 0) return r;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return r;
    const size_t need = (size_t)k * nblocks;
    std::lock_guard<std::mutex> lock(g_red_mu);
    RedRow& row = g_red_rows[std::make_tuple(dev, key_acc, s)];
    if (!row.base || row.floats < need) {
        //  a row is allocated on the first launch of its (accumulator, stream): not under stream capture (hipMalloc is illegal there and would
        // invalidate the capture) -- the launcher then reports DASR_ECAPTURE; run the list once eagerly before capturing it, as engine.run_parallel does
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            g_red_capture_refused = true;
            return r;
        }
        const size_t cap = need < 4096 ? 4096 : need + need / 2;
        char* nb = nullptr;
        if (hipMalloc((void**)&nb, 256 + cap * sizeof(float)) != hipSuccess) return r;   // (a launcher that gets no row reports DASR_EINVAL)
        if (hipMemsetAsync(nb, 0, 256, s) != hipSuccess) return r;   // on the launch stream: ordered in front of the row's first kernel (a plain hipMemset
                                                                      // runs on the null stream, which a non-blocking stream does not wait for: the ticket of a
                                                                      // replica stream's first launch started from garbage and its sum was lost)
        row.base = nb;   // the previous allocation is kept (see above)
        row.floats = cap;
    }
    r.ticket = (unsigned*)row.base;
    r.part = (float*)(row.base + 256);
    return r;
}

// ---- profiling session (see DASR_LAUNCH in common.h) ------------------------------------------------------------------------
namespace {
struct ProfRec {
    hipEvent_t e0, e1;
    const char* tag;
    double flops, bytes;
    int op;
};
ProfRec* g_prof = nullptr;
int g_prof_cap = 0;
std::atomic<int> g_prof_n{0};
std::atomic<bool> g_prof_on{false};
// algorithmic work of the op being dispatched (consumed by its first launch); per enqueue thread (dasr_run_ops_mt)
thread_local double g_prof_flops = 0.0, g_prof_bytes = 0.0;
thread_local int g_prof_op = 0;
char g_prof_filter[128] = {0};   // non-empty: only launches whose tag contains it get events (dasr_prof_filter)
}  // namespace

bool dasr_prof_slot(const char* tag, hipEvent_t* e0, hipEvent_t* e1) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return false;
    if (g_prof_filter[0] && !strstr(tag, g_prof_filter)) return false;
    const int slot = g_prof_n.fetch_add(1, std::memory_order_relaxed);
    if (slot >= g_prof_cap) {
        g_prof_n.store(g_prof_cap, std::memory_order_relaxed);
        return false;
    }
    ProfRec& r = g_prof[slot];
    r.tag = tag;
    r.flops = g_prof_flops;
    r.bytes = g_prof_bytes;
    r.op = g_prof_op;
    g_prof_flops = g_prof_bytes = 0.0;
    *e0 = r.e0;
    *e1 = r.e1;
    return true;
}

extern "C" int dasr_prof_filter(const char* substr) {
    if (g_prof_on.load()) return DASR_EINVAL;   // between sessions only
    const size_t n = substr ? strlen(substr) : 0;
    if (n >= sizeof(g_prof_filter)) return DASR_EINVAL;
    memset(g_prof_filter, 0, sizeof(g_prof_filter));
    if (n) memcpy(g_prof_filter, substr, n);
    return 0;
}

extern "C" int dasr_prof_begin(int32_t capacity) {
    if (capacity <= 0) return DASR_EINVAL;
    if (capacity > g_prof_cap) {
        ProfRec* np = (ProfRec*)realloc(g_prof, sizeof(ProfRec) * (size_t)capacity);
        if (!np) return DASR_EINVAL;
        g_prof = np;
        for (int i = g_prof_cap; i < capacity; ++i) {
            HIP_TRY(hipEventCreate(&g_prof[i].e0));
            HIP_TRY(hipEventCreate(&g_prof[i].e1));
            g_prof_cap = i + 1;
        }
    }
    g_prof_n.store(0);
    g_prof_on.store(true);
    return 0;
}

extern "C" int dasr_prof_end(int32_t max_out, float* us_out, double* flops_out, double* bytes_out, int32_t* op_out, const char** tag_out) {
    g_prof_on.store(false);
    HIP_TRY(hipDeviceSynchronize());
    const int recorded = g_prof_n.load() < g_prof_cap ? g_prof_n.load() : g_prof_cap;
    const int n = recorded < max_out ? recorded : max_out;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g_prof[i].e0, g_prof[i].e1));
        us_out[i] = ms * 1e3f;
        flops_out[i] = g_prof[i].flops;
        bytes_out[i] = g_prof[i].bytes;
        op_out[i] = g_prof[i].op;
        tag_out[i] = g_prof[i].tag;
    }
    return n;
}

extern "C" int dasr_run_ops(const dasr_op* ops, int32_t n, void* stream0) {
    void* stream = stream0;  // DASR_OP_SET_STREAM redirects the following ops
    for (int k = 0; k < n; ++k) {
        const dasr_op& o = ops[k];
        int rc = 0;
        g_prof_flops = o.flops;
        g_prof_bytes = o.bytes;
        g_prof_op = o.op | ((o.i[7] & 0xff) << 8);   // i[7]: the plan builder's time-bucket tag (no kernel reads it), returned by dasr_prof_end
        switch (o.op) {
            case DASR_OP_CONV: rc = dasr_conv(&o.conv, stream); break;
            case DASR_OP_WGRAD:
                rc = dasr_wgrad((const dasr_wgrad_part*)o.p[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], (float*)o.p[1], stream);
                break;
            case DASR_OP_WGRAD_REDUCE:
                rc = dasr_wgrad_reduce((const dasr_wgrad_reduce_part*)o.p[0], o.i[0], (const float*)o.p[1], (float*)o.p[2],
                                       o.f[1] != 0.f ? o.f[0] * o.f[1] : o.f[0], o.i[1], stream);   // f[1]: inverse of the f16 operand pre-scale; i[1]: few_splits
                break;
            case DASR_OP_PACK:
                rc = dasr_pack_weights((const dasr_pack_desc*)o.p[0], o.i[0], o.l[0], (const int64_t*)o.p[1], (const float*)o.p[2], o.p[3], stream);
                break;
            case DASR_OP_DOWNSUM: rc = dasr_downsum2x(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[1], o.i[4], o.f[0], o.t[2], o.t[3], stream); break;
            case DASR_OP_AXPBY: rc = dasr_axpby(o.t[0], o.f[0], o.t[1], o.f[1], o.i[0], o.i[1], o.i[2], o.i[3], o.t[2], o.t[3], o.f[2], o.t[4], o.f[3], (const float*)o.p[0], stream); break;
            case DASR_OP_FILL: rc = dasr_fill_f32((float*)o.p[0], o.l[0], o.f[0], stream); break;
            case DASR_OP_L1LOSS:
                rc = dasr_l1_loss(o.t[0], (const float*)o.p[0], (const float*)o.p[1], o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], (float*)o.p[2], o.t[1],
                                  o.i[4], o.f[1], stream);
                break;
            case DASR_OP_NCHW2B: rc = dasr_nchw_to_blocked((const float*)o.p[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[0], o.t[1], stream); break;
            case DASR_OP_B2NCHW: rc = dasr_blocked_to_nchw(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], (float*)o.p[0], stream); break;
            case DASR_OP_INORM_FWD: rc = dasr_inorm_lrelu_fwd(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], o.f[1], o.t[1], (float*)o.p[0], stream); break;
            case DASR_OP_INORM_BWD: rc = dasr_inorm_lrelu_bwd(o.t[0], o.t[1], o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], (const float*)o.p[0], o.t[2], stream); break;
            case DASR_OP_BCE:
                rc = dasr_gan_loss(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.f[0], o.f[1], o.f[2], (float*)o.p[0], (float*)o.p[1], o.f[3], o.t[1], stream);
                break;
            case DASR_OP_DWT_FWD: rc = dasr_dwt_fwd(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.t[1], o.t[2], stream); break;
            case DASR_OP_DWT_BWD: rc = dasr_dwt_bwd(o.t[0], o.t[1], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.t[2], o.i[5], stream); break;
            case DASR_OP_LOWPASS:
                rc = dasr_lowpass(o.t[0], o.t[1], (const float*)o.p[0], o.i[4], o.i[0], o.i[1], o.i[2], o.i[3], o.i[5], o.f[0], o.f[1], o.t[2], o.t[3], o.i[6], stream);
                break;
            case DASR_OP_MAXPOOL: rc = dasr_maxpool2(o.t[0], o.i[4], o.i[0], o.i[1], o.i[2], o.i[3], o.t[1], o.i[6], stream); break;   // i[6]: input width (0 = 2 * Wo)
            case DASR_OP_MAXPOOL_BWD: rc = dasr_maxpool2_bwd(o.t[0], o.t[1], o.i[4], o.i[0], o.i[1], o.i[2], o.i[3], o.t[2], o.i[5], o.i[6], stream); break;
            case DASR_OP_L1DIFF:
                rc = dasr_l1_diff(o.t[0], o.t[1], o.i[4], o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], o.f[1], (float*)o.p[0], o.t[2], stream);
                break;
            case DASR_OP_AFFINE4: rc = dasr_affine4(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], &o.f[0], (const float*)o.l, o.t[1], o.i[4], o.i[5], stream); break;
            case DASR_OP_BILINEAR: rc = dasr_bilinear_up((const float*)o.p[0], o.i[0], o.i[1], o.i[2], o.i[3], (float*)o.p[1], stream); break;
            case DASR_OP_LOGLOSS:
                rc = dasr_logloss(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], o.f[1], o.f[2], (float*)o.p[0], (float*)o.p[1], o.f[3], o.t[1], o.i[4], stream);
                break;
            case DASR_OP_SIGMOID_BWD: rc = dasr_sigmoid_bwd(o.t[0], o.t[1], o.i[0], o.i[1], o.i[2], o.i[3], o.t[2], stream); break;
            case DASR_OP_PRELU_GRAD:
                // i[4] != 0: y and gx are f16 tensors, gx pre-scaled by 1 / f[1]
                rc = o.i[4] ? dasr_prelu_grad_f16(o.t[0], o.t[1], o.i[0], o.i[1], o.i[2], o.i[3], (const float*)o.p[0], (float*)o.p[1], (float*)o.p[2],
                                                  o.f[0] * (o.f[1] != 0.f ? o.f[1] : 1.f), stream)
                            : dasr_prelu_grad(o.t[0], o.t[1], o.i[0], o.i[1], o.i[2], o.i[3], (const float*)o.p[0], (float*)o.p[1], (float*)o.p[2], o.f[0], stream);
                break;
            case DASR_OP_LOWPASS_VALID: rc = dasr_lowpass_valid(o.t[0], (const float*)o.p[0], o.i[4], o.i[0], o.i[1], o.i[2], o.i[3], o.i[5], o.t[1], o.i[6], stream); break;
            case DASR_OP_ADD_FLAT: rc = dasr_add_flat((float*)o.p[0], (const float*)o.p[1], o.l[0], stream); break;
            case DASR_OP_SIGMOID_FWD: rc = dasr_sigmoid_fwd(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[1], stream); break;
            case DASR_OP_EVENT_RECORD: rc = (int)hipEventRecord((hipEvent_t)o.p[0], as_stream(stream)); break;
            case DASR_OP_STREAM_WAIT: rc = (int)hipStreamWaitEvent(as_stream(stream), (hipEvent_t)o.p[0], 0); break;
            case DASR_OP_SET_STREAM: stream = o.p[0] ? o.p[0] : stream0; break;
            case DASR_OP_PIXSHUF: rc = dasr_pixel_shuffle_f16(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[1], stream); break;
            case DASR_OP_PIXUNSHUF: rc = dasr_pixel_unshuffle_f16(o.t[0], o.t[1], o.f[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[2], stream); break;
            case DASR_OP_CVT_F16:   // i[4]: 0 plain f16 copy, 1 split f16, 2 split bf16, 3 f32 residual of the f16 rounding
                rc = o.i[4] == 3 ? dasr_f16_residual(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], o.t[1], stream)
                     : o.i[4]    ? dasr_cvt_split16(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], o.t[1], o.i[4] == 1, stream)
                                 : dasr_cvt_f16(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], o.t[1], stream);
                break;
            case DASR_OP_DOWNSUM_F16: rc = dasr_downsum2x_f16(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[1], o.f[0], o.f[1], o.t[2], o.t[3], stream); break;
            case DASR_OP_BNORM_FWD:
                rc = dasr_bnorm_lrelu_fwd(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.f[0], o.f[1], (const float*)o.p[0], (const float*)o.p[1], o.t[1],
                                          (float*)o.p[2], stream);
                break;
            case DASR_OP_BNORM_BWD:
                rc = dasr_bnorm_lrelu_bwd(o.t[0], o.t[1], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.f[0], (const float*)o.p[0], (const float*)o.p[1],
                                          (const float*)o.p[2], o.t[2], (float*)o.p[3], (float*)o.l[0], o.f[1], stream);
                break;
            case DASR_OP_BNORM_RUNNING:
                rc = dasr_bnorm_running((const float*)o.p[0], o.i[0], o.i[1], o.i[2], o.f[0], (float*)o.p[1], (float*)o.p[2], (float*)o.p[3], stream);
                break;
            case DASR_OP_INORM_JVP: rc = dasr_inorm_lrelu_jvp(o.t[0], o.t[1], o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], (const float*)o.p[0], o.t[2], stream); break;
            case DASR_OP_INORM_SECOND:
                rc = dasr_inorm_second(o.t[0], o.t[1], o.t[2], o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], (const float*)o.p[0], o.t[3], o.i[4], stream);
                break;
            case DASR_OP_PRELU_FINAL:
                rc = dasr_prelu_final((const float*)o.p[0], o.i[0], o.l[0], o.i[1], (const float* const*)o.p[1], (float* const*)o.p[2], o.f[0], stream);
                break;
            case DASR_OP_BNORM_JVP:
                rc = dasr_bnorm_lrelu_jvp(o.t[0], o.t[1], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.f[0], (const float*)o.p[0], (const float*)o.p[1], (const float*)o.p[2],
                                          o.t[2], stream);
                break;
            case DASR_OP_BNORM_SECOND:
                rc = dasr_bnorm_second(o.t[0], o.t[1], o.t[2], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.f[0], (const float*)o.p[0], (const float*)o.p[1],
                                       (const float*)o.p[2], o.t[3], o.i[5], (float*)o.p[3], o.f[1], stream);
                break;
            case DASR_OP_GRAD_PENALTY: rc = dasr_grad_penalty(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], (float*)o.p[0], (float*)o.p[1], (float*)o.p[2], o.i[4], o.i[5] > 0 ? o.i[5] : 1, stream); break;
            case DASR_OP_FILL_SCALED: rc = dasr_fill_scaled(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], (const float*)o.p[0], o.f[0], stream); break;
            case DASR_OP_CONV_CHAIN: rc = dasr_conv_chain((const dasr_conv_params*)o.p[0], (const dasr_conv_params*)o.p[1], (const int32_t*)o.p[2], o.i[0], (uint32_t*)o.p[3], (int32_t*)o.l[0], stream); break;
            case DASR_OP_RDB_CHAIN: rc = dasr_rdb_chain((const dasr_conv_params*)o.p[0], (const dasr_conv_params*)o.p[1], o.i[0], (uint32_t*)o.p[3], (int32_t*)o.l[0], stream); break;
            case DASR_OP_DDM_SPREAD: rc = dasr_ddm_spread(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], o.i[6], o.f[0], o.t[1], stream); break;
            case DASR_OP_RAGAN:
                rc = dasr_ragan(o.t[0], o.t[1], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], o.f[0], o.f[1], o.f[2], o.f[3], *(const float*)&o.l[2],
                                (float*)o.p[0], (float*)o.p[1], (float*)o.p[2], (float*)o.p[3], (float*)o.l[0], *(const float*)&o.l[1], o.t[2], o.t[3],
                                stream);
                break;
            case DASR_OP_LPIPS_S2D: rc = dasr_lpips_s2d(o.t[0], o.i[0], o.i[1], o.i[2], &o.f[0], (const float*)o.l, o.t[1], o.i[3], stream); break;
            case DASR_OP_MAXPOOL3: rc = dasr_maxpool3s2(o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[1], stream); break;
            case DASR_OP_MAXPOOL3_BWD: rc = dasr_maxpool3s2_bwd(o.t[0], o.t[1], o.i[0], o.i[1], o.i[2], o.i[3], o.t[2], o.i[4], o.i[5], stream); break;
            case DASR_OP_LPIPS_HEAD:
                rc = dasr_lpips_head(o.t[0], o.l[0], o.i[0], o.i[1], o.i[2], o.i[3], (const float*)o.p[0], o.f[0], o.f[1], o.f[2], (float*)o.p[1], o.t[1],
                                     o.i[4], stream);
                break;
            default: rc = DASR_EINVAL;
        }
        if (rc != 0) {
            g_last_failed_op = k;
            return rc;
        }
    }
    return 0;
}

// ---- multi-threaded enqueue ---------------------------------------------------------------------------------------------------
// The sub-batch replicas of a training step are independent op lists on their own HIP streams.  One host thread enqueues ~250-400 k
// launches/s; with three or four replica streams (5 000+ launches per step at configs[1]) a single enqueuer is what the GPU waits for.
// dasr_run_ops_mt gives every list its own enqueue thread (a persistent pool: list 0 runs on the caller, lists 1.. on workers bound to the
// caller's device), and returns when every list has been ENQUEUED (not executed).  Events recorded / waited inside the lists keep their meaning:
// hipEventRecord / hipStreamWaitEvent are ordered per stream, and an event another list waits for must have been recorded by an EARLIER call
// (the trainers only wait for replica events on the communication stream after this function has returned).
namespace {
struct EnqWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    const dasr_op* ops = nullptr;
    int n = 0, dev = 0, rc = 0, failed = -1;
    void* stream = nullptr;
    bool has_work = false, done = false, quit = false;
    void loop() {
        int cur_dev = -1;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return has_work || quit; });
            if (quit) return;
            has_work = false;
            lk.unlock();
            int r = 0;
            if (cur_dev != dev) {
                r = (int)hipSetDevice(dev);
                cur_dev = dev;
            }
            if (r == 0) r = dasr_run_ops(ops, n, stream);
            const int f = dasr_last_failed_op();
            lk.lock();
            rc = r;
            failed = f;
            done = true;
            cv.notify_all();
        }
    }
};
constexpr int kMaxEnqWorkers = 7;
EnqWorker* g_enq[kMaxEnqWorkers] = {};
std::mutex g_enq_mutex;   // one dasr_run_ops_mt call at a time
struct EnqPoolReaper {
    ~EnqPoolReaper() {
        for (auto*& w : g_enq) {
            if (!w) continue;
            {
                std::lock_guard<std::mutex> lk(w->m);
                w->quit = true;
            }
            w->cv.notify_all();
            if (w->th.joinable()) w->th.join();
            delete w;
            w = nullptr;
        }
    }
} g_enq_reaper;
}  // namespace

extern "C" int dasr_run_ops_mt(const dasr_op* const* lists, const int32_t* counts, void* const* streams, int32_t nlists) {
    if (nlists <= 0 || nlists > kMaxEnqWorkers + 1 || !lists || !counts || !streams) return DASR_EINVAL;
    std::lock_guard<std::mutex> guard(g_enq_mutex);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    for (int i = 1; i < nlists; ++i) {
        EnqWorker*& w = g_enq[i - 1];
        if (!w) {
            w = new EnqWorker();
            w->th = std::thread([w] { w->loop(); });
        }
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->ops = lists[i];
            w->n = counts[i];
            w->stream = streams[i];
            w->dev = dev;
            w->done = false;
            w->has_work = true;
        }
        w->cv.notify_all();
    }
    int rc = dasr_run_ops(lists[0], counts[0], streams[0]);
    int failed = rc ? dasr_last_failed_op() : -1;
    for (int i = 1; i < nlists; ++i) {
        EnqWorker* w = g_enq[i - 1];
        std::unique_lock<std::mutex> lk(w->m);
        w->cv.wait(lk, [&] { return w->done; });
        if (rc == 0 && w->rc != 0) {
            rc = w->rc;
            failed = w->failed | (i << 24);   // bits 24..: index of the failing list
        }
    }
    if (rc != 0) g_last_failed_op = failed;
    return rc;
}
