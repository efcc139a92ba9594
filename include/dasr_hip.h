/*
 * dasr_hip.h -- C ABI of libdasr_hip.so, the MI355X (gfx950) kernels behind the DASR SRN
 * training step (ShuhangGu/DASR, codes/SRN).  Plain pointers and sizes only; no torch types.
 *
 * The reference has NO native/FFI layer: every op below replaces a stock
 * PyTorch op that the reference reaches through nn.Module graphs.  The citation on each entry
 * point names the reference construct whose arithmetic it takes over.
 *
 * Activation layout ("NC16HW16"): T[n][cb][y][x][16], 16 channels innermost, bf16 or f32.
 * A dasr_tensor is a view: base pointer of plane cb=0 of the slice, element strides between
 * images and between 16-channel planes.  H, W travel in the op parameters.
 *
 * All launchers are asynchronous on `stream` (a hipStream_t passed as void*), own no memory,
 * and return 0 on success or a hipError_t / negative DASR_E* code.
 */
#ifndef DASR_HIP_H
#define DASR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DASR_ABI_VERSION 20
#define DASR_EINVAL (-22)
#define DASR_ECAPTURE (-16)   /* a launch needed a device allocation (the scratch row of a deterministic grid sum, first use per accumulator and stream) while its stream was being captured */

typedef struct {
    void*   p;          /* base of plane 0 (device pointer) */
    int64_t n_stride;   /* elements between images */
    int64_t cb_stride;  /* elements between 16-channel planes */
} dasr_tensor;

/* ---- implicit-GEMM convolution, forward and data-gradient ------------------------------------
 * Replaces nn.Conv2d fwd (+bias, +LeakyReLU/ReLU, + residual scale-add) of conv_block
 * (codes/SRN/models/modules/block.py:130-156), the torch.cat of ResidualDenseBlock_5C
 * (block.py:280-286: input = first cin/16 planes of a dense slab, output written into its own
 * planes), nn.Upsample(nearest,2) of upconv_blcok (block.py:854-861, `ups`), the 4x4 convs of
 * NLayerDiscriminator (architecture.py:983-1024) and VGG19 convs (architecture.py:1060-1088).
 * The data-gradient of a stride-1 conv is the same op on flipped/transposed packed weights
 * with `mask` = forward activation (LeakyReLU'/ReLU').
 *   v = acc + bias[oc]; if act: v = v>0 ? v : slope*v; if mask.p: v *= (mask>0 ? 1 : slope)
 *   v = alpha*v + beta1*res1 + beta2*res2;  out_f32 = v;  out_bf16 = bf16(gamma*v)
 */
typedef struct {
    dasr_tensor in;  int32_t in_f32;  int32_t Hin, Win;  int32_t ups;  int32_t cin; /* cin % 16 == 0 */
    const void* w;   int64_t w_lo_off;      /* packed bf16 weights (dasr_pack_weights); lo plane offset (elements) for prec 3 */
    const float* bias;                      /* [cout] or NULL */
    int32_t cout, Hout, Wout, N;
    int32_t kh, stride, pad;                /* 3/1/1 or 4/2/1 or 4/1/1; the f32-tensor 3x3 stride-1 conv also takes pad 0 / 2 (LPIPS conv1 and its adjoint) */
    int32_t prec;                           /* 1: bf16 operands; 2: f16 operands, one MFMA pass (f32 input, see in_scale); 3: split-bf16 (hi*hi+hi*lo+lo*hi), ~fp32;
                                             * 4: split-f16 (same three passes on f16 hi/lo pairs of the in_scale'd f32 input: 22-bit operands) */
    int32_t mt;                             /* 32-oc tiles per workgroup (1 or 2); must match the packing */
    int32_t act;  float slope;
    dasr_tensor mask;  int32_t mask_f32;
    float alpha;  dasr_tensor res1;  float beta1;  dasr_tensor res2;  float beta2;
    dasr_tensor out_f32;  dasr_tensor out_bf16;  float gamma;
    int32_t xcd_remap;                      /* set by the library (XCD-aware tile order); callers leave it 0 */
    /* extensions used by the stride-2 data-gradient (four 2x2 parity sub-convolutions writing interleaved sub-grids):
     * pad_x < 0: same as pad; out_stride 0/1: dense output, 2: output pixel (oy,ox) lands at (2*oy+out_oy, 2*ox+out_ox)
     * of a tensor of width out_W (mask / res / out tensors are all indexed at that full-resolution position). */
    int32_t pad_x, out_stride, out_oy, out_ox, out_W;
    /* act: 0 none, 1 (Leaky/P)ReLU with `slope`, 2 sigmoid.  slope_ptr != NULL: slope read from device memory
     * (nn.PReLU's learned parameter, codes/DSN/model.py:29,215), used for `act` and for the `mask` derivative. */
    const float* slope_ptr;
    /* strided INPUT view (sub-pixel data-gradient of nearest-x2 + 3x3: the parity sub-grids of the gradient are read in place):
     * in_stride 0/1: dense; 2: input pixel (y,x) of the Hin x Win grid lives at (2*y+in_oy, 2*x+in_ox) of a tensor of width in_W.
     * conv_kernel variants only (not the LDS-DMA dense-block kernel). */
    int32_t in_stride, in_oy, in_ox, in_W;
    /* prec 2: the f32 input is multiplied by in_scale (a power of two; 0 = 1) before it is rounded to f16, and the accumulator by
     * 1/in_scale before the epilogue: keeps tiny loss gradients (1e-8..1e-5) inside f16's normal range, exact otherwise. */
    float in_scale;
    /* non-zero: the 16-bit output tensor (`out_bf16`) is written as f16 instead of bf16 (activations of the generator's HR tail are
     * kept in f16: 11-bit mantissa, one MFMA pass in the consuming conv, prec 2 with in_f32 = 0) */
    int32_t out16_f16;
    /* SPLIT 16-BIT TENSORS (round 3): a tensor of C channels held as 2K planes of 16-bit elements (K = ceil(C/16)): planes [0, K) the rounded
     * value `hi`, planes [K, 2K) the remainder lo = round16(v - hi) -- 22 (f16) or 16 (bf16) mantissa bits in the same 4 bytes per element as
     * f32, but in the layout the LDS-DMA dense-conv kernel streams.  The split-precision product hi*hi + lo*hi + hi*lo (prec 3 / 4 on f32
     * tensors: three MFMA passes over operands split on the fly) then is ONE launch of that kernel over 3K "virtual" chunks:
     *   in_wrap = 2K, cin = 3 * 16K: chunk c reads plane (c < in_wrap ? c : c - in_wrap) of `in`, i.e. hi, lo, hi again; the weights are
     *   packed as [hi | hi | lo] (dasr_pack_weights fmt 3 / 4).  0: plain tensor.
     *   out16_lo = K' > 0: the 16-bit output is written split: hi into plane cb, lo = round16(gamma * v - hi) into plane cb + K'.
     *   res1_lo = K' > 0: `res1` is a split 16-bit tensor (same element format as the 16-bit output), read as hi + lo.
     * LDS-DMA kernel only (16-bit input, 3x3 / stride 1 / pad 1, prec 1 or 2). */
    int32_t in_wrap, out16_lo, res1_lo;
    /* (ABI 20) non-NULL, 64-channel mask-only data-gradient convs of the LDS-DMA kernel (16-bit input, `mask` a 16-bit tensor in the format of the 16-bit output, no other
     * epilogue term): every workgroup also writes prelu_part[workgroup] = slope * sum over its outputs with mask <= 0 of (conv result * mask) -- the per-workgroup partials
     * of dL/dslope of the nn.PReLU() (one shared slope, codes/DSN/model.py:29,215) whose output the mask is.  The buffer holds >= N * ceil(H/4) * ceil(W/16) floats (one per workgroup of any tile shape the launcher may choose), zero at
     * allocation (entries no workgroup owns stay zero); dasr_prelu_final finishes the sum.  Elsewhere the field must be NULL. */
    float* prelu_part;
} dasr_conv_params;

int dasr_conv(const dasr_conv_params* p, void* stream);
/* A chain of dense-block convs in ONE persistent launch (round 4; replaces the per-layer launches of ResidualDenseBlock_5C / RRDB forward,
 * codes/SRN/models/modules/block.py:254-309, for the trunk of RRDBNet, architecture.py:174-205).  Layer L is dasr_conv(layers[L]) -- same arithmetic,
 * bit-identical results -- but a tile of the image goes from layer to layer inside the launch and waits for its eight neighbour tiles (flags in
 * `flags`, one word per tile, zero at allocation) only before it reads the first input chunk that holds layer L-1's output (dep_chunk[L], in
 * 16-channel chunks; <= 0: every chunk).  Constraints (DASR_EINVAL otherwise): 3x3 / stride 1 / pad 1 on 16-bit tensors of ONE geometry,
 * cout == 32 * mt, the term sets of the dense-block forward (bias + LeakyReLU -> 16-bit planes; bias, alpha, one or two fp32 residuals -> fp32 +
 * 16-bit), N a multiple of 8 (whole images per XCD) and N * tiles == 512 (the launch fills the chip exactly: every workgroup resident, every XCD
 * hosts the tiles of its own images).
 * `flags`: N * tiles + 8 words (the last eight: per-XCD ticket counters).  `err` (device word, zero at allocation): bit 1 a neighbour wait gave
 * up -- the results are then not valid (pass the word to dasr_adam as gate_flag: such a step then never reaches the weights).
 * OPERATIONAL CONTRACT: the launch needs the device to itself -- a whole 256-CU MI355X (no CPX / DPX partition), no other process on it, and no
 * other kernel of this process in flight on another stream (a collective's kernels on a communication stream included) while it runs: all 512
 * workgroups must be resident at once, 64 per XCD.  See INTEGRATION.md "Chained launches: operational contract".
 * dev_layers / dev_dep_chunk: device copies the kernel reads; host_layers: the same blocks in host memory (validated by the launcher). */
int dasr_conv_chain(const dasr_conv_params* dev_layers, const dasr_conv_params* host_layers, const int32_t* dev_dep_chunk, int32_t nlayers,
                    uint32_t* dev_flags, int32_t* dev_err, void* stream);
/* The same chain, INPUT-STATIONARY form (round 6; replaces ResidualDenseBlock_5C.forward, codes/SRN/models/modules/block.py:280-286, and its data
 * gradient, for the whole trunk of RRDBNet, architecture.py:174-205, in one launch).  layers[5 b + k], k = 0..4, are the five convs of dense block b
 * exactly as dasr_conv would run them (cin = 64 + 32 k on ONE slab, cout 32 / 32 / 32 / 32 / 64; forward: bias + LeakyReLU -> 16-bit planes, conv5 bias,
 * alpha, one or two fp32 residuals -> fp32 stream (+ 16-bit planes of the next slab, may be absent on the last block); data gradient: LeakyReLU' mask ->
 * 16-bit planes, alpha, residuals -> fp32 (+ 16-bit)) -- bit-identical results -- but every 16-channel chunk of the slab is staged in LDS ONCE and
 * multiplied into the accumulators of every conv of the block that consumes it (12 chunks per block and tile instead of 40), one workgroup of 8 waves per
 * CU.  Geometry: tiles of 16, 8 or 4 rows x 32 pixels (8 waves x 2 rows, 8 x 1, 4 x 1), T per image; image n on XCD n % 8; q workgroups per XCD (the largest multiple of
 * T that is <= 32, divides the N T / 8 tiles of an XCD and leaves at most 8 tiles per workgroup), i.e. a launch of 8 q <= 256 workgroups (csrc/rdb_is.h); the launcher takes
 * the height that minimises (tiles per workgroup) x (measured chain time of one tile at that height).  Constraints (DASR_EINVAL otherwise): bf16 storage, nlayers a multiple
 * of 5, N a multiple of 8, T <= 32 and such a q exists for one of the heights (16 x 128^2: 16 rows, q 32, two tiles each; 8 x 128^2: 16 rows, one tile; 16 crops of
 * 32 x 32: 4 rows, q 16).  flags: N * ceil(H / 4) * ceil(W / 32) + 8 words, zero at allocation (sized for the finest tiles whichever height runs; the last eight: per-XCD
 * ticket counters); err / operational contract: as dasr_conv_chain (the launch needs all 256 CUs of the device to itself: up to 256 workgroups with 160 KB of LDS each, all
 * resident). */
int dasr_rdb_chain(const dasr_conv_params* dev_layers, const dasr_conv_params* host_layers, int32_t nlayers, uint32_t* dev_flags, int32_t* dev_err,
                   void* stream);
/* kernel-variant knobs for A/B runs (bench.py --sweep / --tune); defaults are the tuned choice.
 * key 1 / 2: dense-block conv with Cout = 32 / 64: 12 = LDS-DMA kernel (default for Cout 32), 13 = its 8-wave 32x32-tile form (Cout 64 default: chosen per launch when the 4-wave grid has <= 256 workgroups), 0 = first-generation register-staged kernel,
 *            1 double-buffered LDS, 4/5 8x32 tiles, 6 4x32 tiles, 8/9 row reuse, 10/11 register-staged pipeline;
 * key 3: split-bf16 stream conv (0 single / 1 double LDS buffer, 4 8x32 tiles); key 4: XCD-aware tile order on/off;
 * key 5: compile-time specialised epilogues on/off;
 *            key 1 also: 15 / 16 / 17 = ring of three LDS images with counted vmcnt / + one loader wave / + LDS flags instead of the chunk barrier
 *            (round 3: built, parity-tested, measured flat -- profiles/r03_conv_ablation.txt);
 * key 7 (-DDASR_BENCH library only): form of the chained launches: 1 = conv_chain_kernel for 512 tiles and conv_chain2_kernel (round 5: workgroups that own
 *        several tiles; measured slower, profiles/r05_chain_trace.txt) for multiples, 2 = conv_chain2_kernel always;
 * (round 3: the Cout-32 dense-block convs store their output `sc1`, written through -- measured with a run-time switch, now compile time.) */
int dasr_set_tuning(int32_t key, int32_t value);

/* ---- weight gradient ---------------------------------------------------------------------------
 * Replaces autograd's convolution_backward (weight, bias) for the convs above.  One launch covers
 * a list of `parts` (each: one 32-oc tile x up to two 32-cin tiles x all taps of one conv) times
 * `nsplit` pixel splits; partial sums go to a workspace, dasr_wgrad_reduce sums them
 * deterministically into the flat gradient buffer in the reference layout [cout][cin][kh][kw].
 */
typedef struct {
    dasr_tensor g;   int32_t g_f32;         /* output-gradient, planes starting at this part's oc tile */
    dasr_tensor in;  int32_t in_f32;        /* layer input, planes starting at this part's first cin tile */
    int32_t ups;                            /* input is nearest-x2 upsampled on the fly */
    int32_t n_ctiles;                       /* 1 or 2 valid 32-cin tiles */
    int32_t g_planes, in_planes;            /* 16-ch planes that really exist (g: 1..2, in: 1..4); the rest reads as zero */
    int32_t Hin, Win, Hout, Wout, N;
    int32_t kh, stride, pad;
    int32_t want_bias;                      /* this part also accumulates sum_p g[oc] */
    int64_t ws_off;                         /* float offset into workspace: [nsplit][taps of the part][32][64] (+ bias [nsplit][32]) */
    int64_t ws_bias_off;
    int32_t tap0;                           /* first tap of this part (5x5 kernels are split into parts of <= 10 taps) */
    float g_scale;                          /* f16 staging (f32 & 2): g is multiplied by g_scale (power of two, 0 = 1) before rounding; the caller folds
                                             * 1/g_scale into dasr_wgrad_reduce's scale */
} dasr_wgrad_part;

/* f32 bit 0: g and in tensors of ALL parts are f32 (rounded while staging) instead of bf16; bit 1: ... rounded to f16 (11-bit mantissa,
 * g pre-scaled by part.g_scale) and multiplied with v_mfma_f32_32x32x16_f16 instead of bf16.
 * kh = 33 selects the 6-wave 3x3 kernel: a part is one 64-channel input block x up to three 32-oc tiles
 * (g_planes = 2/4/6), workspace [split][tap][3][32][64], bias [split][96]. */
int dasr_wgrad(const dasr_wgrad_part* parts_dev, int32_t nparts, int32_t nsplit, int32_t kh, int32_t stride, int32_t f32,
               float* ws, void* stream);
/* bit 0: 1 = ds_read_b64_tr_b16 gathers, 0 = scalar LDS gathers (dasr_probe_tr16 sets it from the device);
 * bit 1: dense-block wgrad3 staged by LDS-DMA instead of registers (A/B; default off); bit 7: kh = 33 launches on 16-bit tensors run
 * wgrad4_kernel (4 waves, LDS-DMA ring of three tiles, register window of X fragments; measured slower, kept as a tested alternative)
 * instead of wgrad3_kernel (12 waves, register-staged). */
int dasr_wgrad_set_mode(int32_t use_tr);

typedef struct {
    int64_t ws_off;  int64_t ws_bias_off;  int32_t nsplit;  int32_t ntaps;
    int32_t oc0, c0;                        /* tile origin inside the conv */
    int32_t cout, cin, n_ctiles;            /* real (unpadded) sizes of the conv */
    int64_t dst_w_off;  int64_t dst_b_off;  /* float offsets into the flat grad buffer; dst_b_off < 0: no bias */
    int32_t flip_io;                        /* reserved */
    /* workspace strides (floats); 0 = the 4-wave layout [split][tap][32][64] / bias [split][32] */
    int64_t split_stride, tap_stride, bias_stride;
    int32_t tap0, ntaps_total;              /* part covers taps [tap0, tap0+ntaps) of a kernel with ntaps_total taps (0: = ntaps) */
    int32_t bias_nsplit, reserved_;         /* splits that carry a bias partial (0: = nsplit).  Split-operand weight gradients (three wgrad parts
                                             * g.x, g.x_lo, g_lo.x laid out as 3 * nsplit consecutive splits) sum the bias over the first nsplit only */
} dasr_wgrad_reduce_part;

/* few_splits != 0 (ABI 17): the caller guarantees nsplit <= 4 and ntaps <= 16 for EVERY part (the grouped dense-block launches): one workgroup per
 * output channel is launched instead of four (same results; the other three only returned). */
int dasr_wgrad_reduce(const dasr_wgrad_reduce_part* parts_dev, int32_t nparts, const float* ws, float* grad_flat,
                      float scale, int32_t few_splits, void* stream);

/* ---- weight packing ------------------------------------------------------------------------------
 * fp32 master weights (reference layout [cout][cin][kh][kw], nn.Conv2d) -> bf16 MFMA-fragment order
 * [mgroup][chunk][tap][mt][lane][8] (+ lo plane for prec 3).  A packed conv is assembled from up to
 * 5 source segments so that the dense-block data-gradient (transposed, tap-flipped, concatenated
 * over the later convs of the block) is just another packed conv.
 */
typedef struct {
    int64_t src_off;      /* float offset of the source conv weight in the flat param buffer */
    int32_t src_cout, src_cin;
    int32_t cin_start, cin_len;   /* range of packed input channels fed by this segment */
    int32_t src_c0;       /* fwd: source cin offset; bwd: source cin (= packed oc) offset */
    int32_t transpose;    /* 0: W[oc][src_c0+ci][tapmap[t]];  1: W[ci][src_c0+oc][tapmap[t]] */
} dasr_pack_seg;

typedef struct {
    int64_t dst_off;      /* bf16 element offset of the hi plane; lo plane at dst_off + lo_off */
    int64_t lo_off;       /* 0 when prec 1 */
    int32_t cout, cin_pad, ntaps, mt, nseg;
    int32_t src_ntaps;    /* taps of the source weight (kh*kw of the nn.Conv2d) */
    int32_t fmt;          /* 0: bf16 (hi plane, + lo plane when lo_off != 0); 1: f16 (prec 2 convs); 2: f16 hi + f16 lo planes (prec 4);
                           * 3 (f16) / 4 (bf16): cin_pad = 3 * 16K virtual channels [hi | hi | lo] of 16K real ones, one plane (split 16-bit tensors,
                           * dasr_conv_params::in_wrap; segments address the real channels) */
    int8_t  tapmap[32];   /* packed tap -> source tap (identity: forward; reversed: stride-1 dgrad; parity subset: stride-2 dgrad) */
    uint16_t tapmask[16]; /* non-zero: packed tap t (< 16) = SUM of the source taps whose bits are set (sub-pixel form of nearest-x2 + 3x3) */
    dasr_pack_seg seg[5];
} dasr_pack_desc;

/* piece_prefix_dev[ndesc+1]: cumulative count of 16-byte output pieces (= mgroups*chunks*ntaps*mt*64 per desc) */
int dasr_pack_weights(const dasr_pack_desc* descs_dev, int32_t ndesc, int64_t total_pieces, const int64_t* piece_prefix_dev,
                      const float* params_flat, void* packed, void* stream);

/* ---- elementwise / reductions --------------------------------------------------------------------*/
/* NCHW f32 [N][C][H][W] (reference tensor layout at the trainer boundary) <-> NC16HW16 f32 (+ optional bf16 copy) */
int dasr_nchw_to_blocked(const float* src, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor dst_f32,
                         dasr_tensor dst_bf16, void* stream);
int dasr_blocked_to_nchw(dasr_tensor src, int32_t N, int32_t C, int32_t H, int32_t W, float* dst, void* stream);

/* L1 pixel loss of SRModel (codes/SRN/models/SR_model.py:80) / DASR_Model (DASR_model.py:212-222):
 *   loss_acc[0] += coef * sum wm*|sr - hr|,   grad (blocked f32, zero beyond C) (+)= coef * wm * sign(sr - hr)
 * coef = weight / element count is computed by the caller; `weight_map` (NCHW [N][1][H][W]) is the optional
 * domain-distance map of the multiweights pixel loss (DASR_model.py:213-215).
 * accumulate bit 0: grad += instead of grad =; bit 1: squared error (nn.MSELoss, pixel_criterion 'l2', SR_model.py:33-36):
 * loss_acc[0] += coef * sum wm*(sr - hr)^2, grad (+)= 2 coef wm (sr - hr);
 * bit 2 (round 6; not with bit 0): `grad` is an f16 tensor and receives f16(grad_scale * g) -- the generator's f16 HR tail takes dL/dSR in that form (a power-of-two
 * pre-scale keeps it inside f16's range); only the 4-channel groups that hold a real channel are written.  grad_scale 0 = 1. */
int dasr_l1_loss(dasr_tensor sr, const float* hr_nchw, const float* weight_map, int32_t N, int32_t C, int32_t H, int32_t W,
                 float coef, float* loss_acc, dasr_tensor grad, int32_t accumulate, float grad_scale, void* stream);

/* backward of nn.Upsample(nearest,2) (block.py:857): dst[y][x] = sum of the 2x2 block of src (Hs=2H, Ws=2W);
 * optional LeakyReLU' mask (mask>0 ? 1 : slope) from the forward activation at the low resolution. */
int dasr_downsum2x(dasr_tensor src, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor mask, int32_t mask_f32,
                   float slope, dasr_tensor dst_f32, dasr_tensor dst_bf16, void* stream);

/* f16 storage of the generator's HR tail: y (f16, blocked) = f16(scale * x) of a blocked f32 tensor (dL/dSR -> pre-scaled f16 gradient), and
 * the backward of nn.Upsample(nearest,2) on f16 tensors: dst = out_scale * mask' * (2x2 block sum of src); dst_f32 and/or dst_f16 */
int dasr_cvt_f16(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, float scale, dasr_tensor y, void* stream);
/* split 16-bit copy (dasr_conv_params::in_wrap): y planes [0, K) = round16(scale * x), planes [K, 2K) = round16(scale * x - hi); f16 != 0: IEEE half, else bfloat16 */
int dasr_cvt_split16(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, float scale, dasr_tensor y, int32_t f16, void* stream);
/* y (f32) = x - f16(scale * x) / scale on a blocked f32 tensor: the part of x an f16 operand rounding drops.  Used to run the weight gradients of
 * the BatchNorm discriminators (Discriminator_VGG_128 architecture.py:442-495, FSD-Batch codes/DSN/model.py:176-189) with 22-bit operands on the
 * f16 MFMA: dW = g.x + g.x_lo + g_lo.x as three parts of one dasr_wgrad launch (their D gradients were 1.3e-2 off an fp64 run). */
int dasr_f16_residual(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, float scale, dasr_tensor y, void* stream);
int dasr_downsum2x_f16(dasr_tensor src, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor mask /* f16, optional */, float slope,
                       float out_scale, dasr_tensor dst_f32, dasr_tensor dst_f16, void* stream);

/* nn.PixelShuffle(2) of pixelshuffle_block (codes/SRN/models/modules/block.py:838-851) on f16 blocked tensors: src has C4 = 4 C channels at H x W,
 * dst C channels at 2H x 2W, dst[c][2y+dy][2x+dx] = src[4c+2dy+dx][y][x]; and its adjoint with the LeakyReLU' of the (activated) src as mask */
int dasr_pixel_shuffle_f16(dasr_tensor src, int32_t N, int32_t C4, int32_t H, int32_t W, dasr_tensor dst, void* stream);
int dasr_pixel_unshuffle_f16(dasr_tensor gsrc, dasr_tensor mask /* f16 [C4 @ HxW], optional */, float slope, int32_t N, int32_t C4, int32_t H,
                             int32_t W, dasr_tensor gdst, void* stream);

/* out = a*x + b*z (z optional) over blocked f32 tensors, optional bf16 copy scaled by gamma
 * (ShortcutBlock / RRDB residual bookkeeping, block.py:97-105,305-309) */
int dasr_axpby(dasr_tensor x, float a, dasr_tensor z, float b, int32_t N, int32_t C, int32_t H, int32_t W,
               dasr_tensor out_f32, dasr_tensor out_bf16, float gamma, dasr_tensor mask /* optional (P)ReLU' mask, f32 */, float slope,
               const float* slope_ptr, void* stream);

/* torch.optim.Adam step (DASR_model.py:129-143; SR_model.py:50-51) on flat fp32 buffers:
 * g += wd*p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * nonfinite_flag (optional, device int): bit 0 is OR-ed in when any gradient element is inf / NaN (the update itself is what torch would do);
 * the trainers read it with their lazily synchronised log and raise -- e.g. an overflow of the f16-stored HR-tail gradients (DASR_HR_PREC=3
 * keeps that tail in split-bf16 on f32 tensors).
 * gate_flag (optional, device int; ABI 17): when the word is non-zero the launch changes NOTHING (weights and moments untouched).  The trainers pass the
 * error word of the generator's chained trunk launches (dasr_conv_chain `err`): gradients computed behind a broken neighbour wait never reach the
 * weights; the host raises at its next synchronisation point (log interval, checkpoint). */
int dasr_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int32_t step, int32_t* nonfinite_flag, const int32_t* gate_flag, void* stream);

int dasr_fill_f32(float* p, int64_t n, float value, void* stream);
/* y += x over flat fp32 buffers (sum of the gradient buffers of concurrently processed sub-batches) */
int dasr_add_flat(float* y, const float* x, int64_t n, void* stream);


/* ---- GAN-step kernels (csrc/gan.hip) ---------------------------------------------------------------------*/
/* nn.InstanceNorm2d(affine=False, eps) + LeakyReLU of NLayerDiscriminator (architecture.py:1003-1015), fused;
 * stats[N][Cpad][2] = (mean, rstd).  Backward takes the saved forward output a and dL/da, returns dL/dx. */
int dasr_inorm_lrelu_fwd(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, float eps, float slope, dasr_tensor y,
                         float* stats, void* stream);
int dasr_inorm_lrelu_bwd(dasr_tensor a, dasr_tensor ga, int32_t N, int32_t C, int32_t H, int32_t W, float slope,
                         const float* stats, dasr_tensor gx, void* stream);
/* Second-order pieces for the DSN's --wgan gradient penalty (codes/DSN/train.py:231-236: torch.autograd.grad(..., create_graph=True) through the
 * discriminator and the backward of that).  InstanceNorm's Jacobian is symmetric, J t = rstd (t - mean t - xhat mean(xhat t)):
 * dasr_inorm_lrelu_jvp: forward-mode tangent  out = lrelu'(a) * J t  (a = saved forward output, t = tangent of the conv output);
 * dasr_inorm_second: adjoint of z -> J(z) t for fixed t, upstream w = lrelu'(a) * ga:
 *   out (+)= -rstd^2 [ xhat (mean(w t) - mean w mean t - 3 mean(w xhat) mean(xhat t)) + mean(xhat t) (w - mean w) + mean(w xhat) (t - mean t) ];
 * dasr_grad_penalty: nrm = ||g||_2 over the C (<= 16) real channels of ALL images, out3 (FOUR floats) = {nrm, weight (nrm - 1)^2, 2 weight (nrm - 1) / nrm, -},
 *   loss_acc[0] += the penalty; part256 = 256 floats of scratch (deterministic two-stage sum).  stage 0 / world 1: all of it.  Data parallel (the
 *   reference's norm is over the global batch): stage 1 leaves the local sum of squares in out3[3], the caller SUM-all-reduces that word, stage 2 finishes
 *   with nrm^2 = out3[3] / world^2 and out3[2] scaled by 1 / world (the reverse pass's weight-gradient reductions carry the other 1 / world; the ranks' gradients are summed);
 * dasr_fill_scaled: x = factor * scalar[0] on the C real channels (a constant upstream gradient whose value was computed on the device). */
int dasr_inorm_lrelu_jvp(dasr_tensor a, dasr_tensor t, int32_t N, int32_t C, int32_t H, int32_t W, float slope, const float* stats, dasr_tensor out, void* stream);
int dasr_inorm_second(dasr_tensor a, dasr_tensor t, dasr_tensor ga, int32_t N, int32_t C, int32_t H, int32_t W, float slope, const float* stats,
                      dasr_tensor out, int32_t accumulate, void* stream);
int dasr_grad_penalty(dasr_tensor g, int32_t N, int32_t C, int32_t H, int32_t W, float weight, float* part256, float* out3, float* loss_acc, int32_t stage,
                      int32_t world, void* stream);
int dasr_fill_scaled(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, const float* scalar, float factor, void* stream);
/* nn.BatchNorm2d in TRAINING mode (batch statistics, affine gamma / beta) + LeakyReLU of Discriminator_VGG_128 (architecture.py:442-495), fused.
 * The N images are normalised in groups of `group` consecutive images with their own statistics (the reference runs the discriminator on the
 * fake and the real half in separate calls, DASR_model.py:251,288-289).  stats[N/group][Cpad][3] = (mean, rstd, biased variance).
 * Backward recomputes xhat / z from the saved conv output x: gx per group; dgamma / dbeta (optional, both or none) = pscale * sums over all groups.
 * dasr_bnorm_running: running_mean / running_var (unbiased, `count` = elements per channel of the group) / num_batches_tracked after one forward on group g */
int dasr_bnorm_lrelu_fwd(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, int32_t group, float eps, float slope, const float* gamma,
                         const float* beta, dasr_tensor y, float* stats, void* stream);
int dasr_bnorm_lrelu_bwd(dasr_tensor x, dasr_tensor ga, int32_t N, int32_t C, int32_t H, int32_t W, int32_t group, float slope, const float* gamma,
                         const float* beta, const float* stats, dasr_tensor gx, float* dgamma, float* dbeta, float pscale, void* stream);
/* Second-order pieces of BatchNorm2d (training mode) + LeakyReLU for `--wgan` with `--norm_layer Batch` (ABI 19; codes/DSN/train.py:231-236 through
 * model.py:136-160,176-189): the InstanceNorm pair above with group means, gamma behind the normalisation and lrelu' read from y = gamma xhat + beta
 * (x = saved conv output).  dasr_bnorm_lrelu_jvp: out = lrelu'(y) gamma r (t - mean t - xhat mean(xhat t)).  dasr_bnorm_second (u = lrelu'(y) ga, w = gamma u):
 * out (+)= -r^2 [xhat (mean(w t) - mean w mean t - 3 mean(w xhat) mean(xhat t)) + mean(xhat t) (w - mean w) + mean(w xhat) (t - mean t)];
 * dgamma (optional; += when `accumulate`) = pscale * sum u xhat_dot -- the tangent output is linear in gamma; beta has no term. */
int dasr_bnorm_lrelu_jvp(dasr_tensor x, dasr_tensor t, int32_t N, int32_t C, int32_t H, int32_t W, int32_t group, float slope, const float* gamma,
                         const float* beta, const float* stats, dasr_tensor out, void* stream);
int dasr_bnorm_second(dasr_tensor x, dasr_tensor t, dasr_tensor ga, int32_t N, int32_t C, int32_t H, int32_t W, int32_t group, float slope,
                      const float* gamma, const float* beta, const float* stats, dasr_tensor out, int32_t accumulate, float* dgamma, float pscale,
                      void* stream);
int dasr_bnorm_running(const float* stats, int32_t g, int32_t C, int32_t count, float momentum, float* running_mean, float* running_var,
                       float* num_batches_tracked, void* stream);
/* GANLoss('vanilla') = BCEWithLogitsLoss vs a constant target (loss.py:8-40): loss_acc += coef*sum(bce),
 * score_acc += score_coef*sum(x) (the disc_Score log), grad = gcoef*(sigmoid(x)-target) */
int dasr_bce_logits(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, float target, float coef, float gcoef,
                    float* loss_acc, float* score_acc, float score_coef, dasr_tensor grad, void* stream);
/* GANLoss(gan_type) (loss.py:8-40; DASR_model.py:35): gan_type 0 'vanilla' (= dasr_bce_logits), 1 'lsgan' (MSELoss: (x-target)^2, grad
 * gcoef*2(x-target)), 2 'wgan-gp' (-x when target is the real label (> 0.5), +x otherwise; the reference builds but never applies its
 * gradient penalty, DASR_model.py:114-118).  Accumulators and grad as in dasr_bce_logits. */
int dasr_gan_loss(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, int32_t gan_type, float target, float coef, float gcoef,
                  float* loss_acc, float* score_acc, float score_coef, dasr_tensor grad, void* stream);
/* relativistic average GAN loss (`ragan: true`, DASR_model.py:240-244,273-275): a, b = logit maps [N][1][H][W] of the two halves,
 *   L = coef * sum_{n,p} [ bce(a - mean_n(b), ta) + bce(b - mean_n(a), tb) ],  means per pixel over the GLOBAL batch (n_glob >= N samples).
 * Three stages so that data-parallel ranks can all-reduce (SUM) the two tiny per-pixel buffers in between (2*H*W floats each):
 *   stage 0: sums = [sum_n a ; sum_n b];   stage 1: loss_acc += coef * local loss, score_a/b += score_coef * sum a / b,
 *   part = [sum_n (sigmoid(za) - ta) ; sum_n (sigmoid(zb) - tb)];   stage 2: ga / gb (optional) = gcoef * d(sum)/da, /db incl. the mean terms
 * form 0: the SRN form above (score = mean logit); form 2 / 3: the same with the GANLoss('lsgan') / ('wgan-gp') term instead of bce.  form 1: the DSN's `--ragan` (codes/DSN/train.py:221-223, model.py:98-106, loss.py:11-41):
 *   term(z, t) = -log(sigmoid(z) + eps) for t > 0.5, -log(1 - sigmoid(z) + eps) for 0 <= t <= 0.5, absent for t < 0; score = mean sigmoid(z). */
int dasr_ragan(dasr_tensor a, dasr_tensor b, int32_t N, int32_t H, int32_t W, int32_t stage, int32_t n_glob, int32_t form, float ta, float tb,
               float coef, float gcoef, float eps, float* sums, float* part, float* loss_acc, float* score_a, float* score_b, float score_coef,
               dasr_tensor ga, dasr_tensor gb, void* stream);
/* Haar DWT level 1 as used by DASR_Model.wavelet_s (DASR_model.py:442-452): LL (C ch) and [LH|HL|HH] (3C ch), optional
 * norm (LL*0.5, Hc*0.5+0.5); and its adjoint (accumulating into gx).  H2, W2 = output size. */
int dasr_dwt_fwd(dasr_tensor x, int32_t N, int32_t C, int32_t H2, int32_t W2, int32_t norm, dasr_tensor ll, dasr_tensor hc, void* stream);
int dasr_dwt_bwd(dasr_tensor gll, dasr_tensor ghc, int32_t N, int32_t C, int32_t H2, int32_t W2, int32_t norm, dasr_tensor gx,
                 int32_t accumulate, void* stream);
/* depthwise k x k low-pass (GaussianFilter / AvgPool2d of FilterLow/FilterHigh, architecture.py:1177-1243), zero pad.
 * mode bit 1 set: normalise by the in-image fraction of the window (AvgPool2d(count_include_pad=False), model.py:69-74).
 * mode 0: out_low = low(x), out_high = a_h*(x - low(x)) + b_h.  mode 1 (adjoint): out_low (+)= low(x) + a_h*(x2 - low(x2))
 * with x = dL/dlow, x2 = dL/dhigh (either may be null). */
int dasr_lowpass(dasr_tensor x, dasr_tensor x2, const float* w, int32_t k, int32_t N, int32_t C, int32_t H, int32_t W,
                 int32_t mode, float a_h, float b_h, dasr_tensor out_low, dasr_tensor out_high, int32_t accumulate, void* stream);
/* nn.MaxPool2d(2,2) of the VGG19 feature stack, forward and backward (Ho, Wo = pooled size; Win = width of the INPUT: 2 Wo or 2 Wo + 1 --
 * an odd input drops its last row / column as nn.MaxPool2d's floor does; 0 = 2 Wo.  Round 3: before, odd inputs were mis-addressed); is_f32: 0 bf16, 1 f32, 2 f16 tensors; 3 / 4: split f16 / bf16 tensors (C channels in 2 * ceil(C/16) planes: hi planes, then lo planes);
 * backward only: 5 = split f16 activations x, plain f16 gradients */
int dasr_maxpool2(dasr_tensor x, int32_t is_f32, int32_t N, int32_t C, int32_t Ho, int32_t Wo, dasr_tensor y, int32_t Win, void* stream);
/* relu_mask: also zero the gradient where the pooled maximum is <= 0 (the ReLU' of the conv feeding the pool) */
int dasr_maxpool2_bwd(dasr_tensor x, dasr_tensor gy, int32_t is_f32, int32_t N, int32_t C, int32_t Ho, int32_t Wo, dasr_tensor gx,
                      int32_t relu_mask, int32_t Win, void* stream);
/* L1 between two blocked tensors (feature loss DASR_model.py:224-229; LL loss :220-222): loss_acc += coef*sum|a-b|,
 * ga = gcoef*sign(a-b).  is_f32 bit 1 set: squared form (MSE of the DSN VGG16 perceptual loss, loss.py:119-130). */
int dasr_l1_diff(dasr_tensor a, dasr_tensor b, int32_t is_f32, int32_t N, int32_t C, int32_t H, int32_t W, float coef, float gcoef,
                 float* loss_acc, dasr_tensor ga, void* stream);
/* per-channel affine on <=4 channels (VGG input normalisation architecture.py:1086-1087 and its adjoint); y_f32: 0 bf16, 1 f32, 2 f16 output, 3 split f16 (hi in plane 0, remainder in plane 1; no accumulate) */
int dasr_affine4(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, const float* scale4, const float* shift4, dasr_tensor y,
                 int32_t y_f32, int32_t accumulate, void* stream);
/* F.interpolate(bilinear, align_corners=False) of the domain-distance map (DASR_model.py:173-174), NCHW [N][1][h][w] */
int dasr_bilinear_up(const float* src, int32_t N, int32_t h, int32_t w, int32_t factor, float* dst, void* stream);

/* ---- LPIPS(alex) perceptual loss (csrc/lpips.hip) ----------------------------------------------------------------------
 * feature_criterion "LPIPS" of the shipped DASR configs: PerceptualLossLPIPS codes/SRN/models/modules/loss.py:66-72 -> PNetLin.forward
 * codes/PerceptualSimilarity/models/networks_basic.py:64-92 on the AlexNet slices of pretrained_networks.py:57-95.  The convs run on
 * dasr_conv (prec 3); these are the layers around them, all on blocked f32 tensors. */
/* mode 0: ScalingLayer (networks_basic.py:94-101, with the 2x-1 of models/util.py:36-38 folded into scale4/shift4) + 4x4 space-to-depth of
 * the 3-channel image (plane 0 of x) zero-padded by 2: y[c][Y][X][4*by+bx] = scale4[c]*x[c][4Y+by-2][4X+bx-2] + shift4[c], 3 planes of
 * (H+4)/4 x (W+4)/4 -- the 11x11/s4/p2 conv becomes 3x3/s1/p0 on 48 channels.  mode 1: adjoint, ACCUMULATED into channels 0..2 of x.
 * mode bits 4-6 (DSN --lpips_rot_flip, PerceptualLoss.forward codes/DSN/loss.py:155-168: torch.rot90 / torch.flip of both images in front of
 * LPIPS): the network sees T(x), T(x)[i][j] = x[u][v], (u, v) = (i, j) swapped if bit 4, u -> H-1-u if bit 5, v -> W-1-v if bit 6; mode 1 routes
 * the gradient back through the same map.  Transposing forms need H == W. */
int dasr_lpips_s2d(dasr_tensor x, int32_t N, int32_t H, int32_t W, const float* scale4, const float* shift4, dasr_tensor y, int32_t mode,
                   void* stream);
/* nn.MaxPool2d(3, 2) of torchvision alexnet.features[2] / [5] on H x W inputs (output (H-3)/2+1), and its backward: gather form, first
 * maximum in scan order wins (ATen); relu_mask: zero where x <= 0; accumulate: gx += (the head gradient of that layer is already there) */
int dasr_maxpool3s2(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor y, void* stream);
int dasr_maxpool3s2_bwd(dasr_tensor x, dasr_tensor gy, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor gx, int32_t relu_mask,
                        int32_t accumulate, void* stream);
/* one LPIPS layer: images n < N of f are compared with images n + pair_off; u = f0/(|f0|_2+eps), v likewise (normalize_tensor),
 * loss_acc += coef * sum_pixels sum_c lin[c] (u_c-v_c)^2;  g0 (optional) = gcoef * d(sum)/df0, zeroed where f0 <= 0 when relu_mask */
int dasr_lpips_head(dasr_tensor f, int64_t pair_off, int32_t N, int32_t C, int32_t H, int32_t W, const float* lin, float eps, float coef,
                    float gcoef, float* loss_acc, dasr_tensor g0, int32_t relu_mask, void* stream);

/* ---- DSN (codes/DSN) kernels ---------------------------------------------------------------------------------*/
/* domain-distance map for any discriminator conv table (receptive_cal.py:34-60, create_dataset_modified.py:14-24,119-126): D output value
 * (i, j) spread over its receptive-field window (jump / rf / start of the walk over the conv table), divided by the coverage count; d =
 * [N][1][n_h][n_w] D output, out = [N][1][H][W] map (channel 0 of 16-channel fp32 planes).  FSD uses the equivalent 17 x 17 box of dasr_lowpass. */
int dasr_ddm_spread(dasr_tensor d, int32_t N, int32_t n_h, int32_t n_w, int32_t H, int32_t W, int32_t jump, int32_t rf, float start,
                    dasr_tensor out, void* stream);
/* -log losses of codes/DSN/loss.py:11-41 on p = sigmoid(logit) (model.py:104-105): mode 0: -log(p+eps), mode 1:
 * -log(1-p+eps); loss_acc += coef*sum, score_acc += score_coef*sum(p), grad (+)= gcoef * d/dlogit */
int dasr_logloss(dasr_tensor x, int32_t N, int32_t H, int32_t W, int32_t mode, float eps, float coef, float gcoef, float* loss_acc,
                 float* score_acc, float score_coef, dasr_tensor grad, int32_t accumulate, void* stream);
/* backward of the generator's output sigmoid (model.py:55) */
int dasr_sigmoid_bwd(dasr_tensor y, dasr_tensor g, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor gz, void* stream);
/* y = sigmoid(x) on C (<= 4) channels of plane 0: the discriminator's output map at inference (codes/DSN/model.py:104-106,
 * consumed by create_dataset_modified.py:14-24) */
int dasr_sigmoid_fwd(dasr_tensor x, int32_t N, int32_t C, int32_t H, int32_t W, dasr_tensor y, void* stream);
/* gradient of nn.PReLU()'s single slope (model.py:29,215) from the layer output y and dL/dx; deterministic two-stage sum;
 * scratch256: 1024 floats (one partial per workgroup of the first stage; the name is historical) */
int dasr_prelu_grad(dasr_tensor y, dasr_tensor gx, int32_t N, int32_t C, int32_t H, int32_t W, const float* slope, float* scratch256,
                    float* dst, float scale, void* stream);
/* the same on f16 tensors (the DSN generator's 16-bit backward: y = f16 shadow of the layer output, gx = power-of-two pre-scaled f16 gradient;
 * the caller folds 1 / pre-scale into `scale`) */
int dasr_prelu_grad_f16(dasr_tensor y, dasr_tensor gx, int32_t N, int32_t C, int32_t H, int32_t W, const float* slope, float* scratch256,
                        float* dst, float scale, void* stream);
/* (ABI 20) the second stage alone for `count` slopes in one launch: workgroup k sums partial[k * stride .. + nblocks) in a fixed order and writes
 * *dsts[k] = scale * sum / (*slopes[k])^2.  The partials are those the data-gradient conv epilogues leave in dasr_conv_params::prelu_part (slope * dL/dh * h per
 * workgroup): the DSN generator's residual blocks no longer read h and dL/dz a second time (dasr_prelu_grad_f16 did). slopes / dsts: DEVICE arrays of device pointers. */
int dasr_prelu_final(const float* partial, int32_t nblocks, int64_t stride, int32_t count, const float* const* slopes, float* const* dsts, float scale, void* stream);
/* un-padded low-pass of the colour loss (FilterLow(padding=False), loss.py:52-56): mode 0 forward (H-k+1 x W-k+1 out),
 * mode 1 adjoint */
int dasr_lowpass_valid(dasr_tensor x, const float* w, int32_t k, int32_t N, int32_t C, int32_t H, int32_t W, int32_t mode,
                       dasr_tensor out, int32_t accumulate, void* stream);

/* ---- device-side input pipeline -----------------------------------------------------
 * Batch assembly of codes/SRN/data/LRHR_wavelet_unpairEq_fake_w_dataset.py:50-166 + data/util.py:116-128 on resident images:
 * sample k of the batch = size x size crop at (y0, x0) of image `src` (CHW fp32, RGB) -- first resized bilinearly to vH x vW when
 * that differs from H x W (cv2.resize(..., INTER_LINEAR) of the domain-distance map to the LR size) -- then horizontal flip,
 * vertical flip, transpose, in that order (util.augment).  dst is [n][C][size][size] fp32. */
typedef struct {
    const float* src;
    int32_t C, H, W;      /* source image */
    int32_t vH, vW;       /* size it is resized to before cropping (== H, W: no resize) */
    int32_t y0, x0;       /* crop origin */
    int32_t flags;        /* bit 0 hflip, bit 1 vflip, bit 2 transpose */
} dasr_crop_desc;
int dasr_gather_crops(const dasr_crop_desc* descs_dev, int32_t n, int32_t C, int32_t size, float* dst, void* stream);

/* ---- executor: run a recorded list of ops in one call (keeps the host out of the step) ----------*/
enum { DASR_OP_CONV = 1, DASR_OP_WGRAD = 2, DASR_OP_WGRAD_REDUCE = 3, DASR_OP_PACK = 4, DASR_OP_DOWNSUM = 5,
       DASR_OP_AXPBY = 6, DASR_OP_FILL = 7, DASR_OP_L1LOSS = 8, DASR_OP_NCHW2B = 9, DASR_OP_B2NCHW = 10,
       DASR_OP_INORM_FWD = 11, DASR_OP_INORM_BWD = 12, DASR_OP_BCE = 13, DASR_OP_DWT_FWD = 14, DASR_OP_DWT_BWD = 15,
       DASR_OP_LOWPASS = 16, DASR_OP_MAXPOOL = 17, DASR_OP_MAXPOOL_BWD = 18, DASR_OP_L1DIFF = 19, DASR_OP_AFFINE4 = 20,
       DASR_OP_BILINEAR = 21, DASR_OP_LOGLOSS = 22, DASR_OP_SIGMOID_BWD = 23, DASR_OP_PRELU_GRAD = 24, DASR_OP_LOWPASS_VALID = 25,
       DASR_OP_ADD_FLAT = 26, DASR_OP_SIGMOID_FWD = 27,
       /* scheduling ops: p[0] = event from dasr_event_create / a hipStream_t (NULL: back to the stream dasr_run_ops was called with) */
       DASR_OP_EVENT_RECORD = 28, DASR_OP_STREAM_WAIT = 29, DASR_OP_SET_STREAM = 30,
       DASR_OP_CVT_F16 = 31, DASR_OP_DOWNSUM_F16 = 32, DASR_OP_PIXSHUF = 33, DASR_OP_PIXUNSHUF = 34,
       DASR_OP_LPIPS_S2D = 35, DASR_OP_MAXPOOL3 = 36, DASR_OP_MAXPOOL3_BWD = 37, DASR_OP_LPIPS_HEAD = 38, DASR_OP_RAGAN = 39,
       DASR_OP_BNORM_FWD = 40, DASR_OP_BNORM_BWD = 41, DASR_OP_BNORM_RUNNING = 42, DASR_OP_DDM_SPREAD = 43,
       /* --wgan gradient penalty (round 4) */
       DASR_OP_INORM_JVP = 44, DASR_OP_INORM_SECOND = 45, DASR_OP_GRAD_PENALTY = 46, DASR_OP_FILL_SCALED = 47,
       DASR_OP_CONV_CHAIN = 48,  /* p[0] device layers, p[1] host layers, p[2] device dep_chunk, i[0] nlayers, p[3] device flags; l[0] device err word */
       DASR_OP_RDB_CHAIN = 49,   /* dasr_rdb_chain: p[0] device layers, p[1] host layers, i[0] nlayers, p[3] device flags; l[0] device err word */
       /* --wgan with BatchNorm discriminators (round 6).  JVP: t[0] x, t[1] t, i[0..3] N C H W, i[4] group, f[0] slope, p[0] gamma, p[1] beta, p[2] stats, t[2] out.
        * SECOND: t[0] x, t[1] t, t[2] ga, i[0..4] as above, f[0] slope, p[0..2] as above, t[3] out, i[5] accumulate, p[3] dgamma, f[1] pscale */
       DASR_OP_BNORM_JVP = 50, DASR_OP_BNORM_SECOND = 51,
       DASR_OP_PRELU_FINAL = 52   /* dasr_prelu_final: p[0] partial, i[0] nblocks, l[0] stride, i[1] count, p[1] slopes, p[2] dsts, f[0] scale */ };

typedef struct {
    int32_t op;  int32_t i[8];  float f[4];  int64_t l[4];  void* p[4];  dasr_tensor t[5];
    dasr_conv_params conv;
    double flops, bytes;   /* algorithmic work of the op (set by the plan builder; only read by the profiling session below) */
} dasr_op;

int dasr_run_ops(const dasr_op* ops, int32_t n, void* stream);
/* Enqueue `nlists` independent op lists, list i on streams[i], each from its own host thread (list 0 on the caller, a persistent pool for
 * the rest, bound to the caller's device); returns when every list has been enqueued.  For the sub-batch replica streams of a training step
 * (dasr_amd/models.py: three or four replicas are 5 000+ launches per step -- more than one enqueue thread feeds).  nlists <= 8.  On failure
 * dasr_last_failed_op() = op index | (list index << 24).  Replaces nothing in the reference (its step is one autograd graph on one stream,
 * codes/SRN/models/SR_model.py:77-85). */
int dasr_run_ops_mt(const dasr_op* const* lists, const int32_t* counts, void* const* streams, int32_t nlists);
/* events for the scheduling ops (hipEventDisableTiming); independent work of one list can be moved to a second stream this way
 * (the dense-block weight gradients are not on the data-gradient chain's critical path) */
void* dasr_event_create(void);
int dasr_event_destroy(void* ev);
int dasr_last_failed_op(void);   /* index of the op that made dasr_run_ops return non-zero */

/* ---- RCCL gradient exchange (csrc/rccl.hip) ------------------------------------------------------------------------------
 * Replaces the single-process nn.DataParallel of the reference (codes/SRN/models/networks.py:144-146,192-193): one process per GPU,
 * replicated weights, SUM all-reduce of the flat fp32 gradient buffer over xGMI (the 1/world factor is folded into dasr_wgrad_reduce).
 * Rank 0 creates a 128-byte id (dasr_rccl_unique_id) and hands it to the other ranks through any side channel; every rank then calls
 * dasr_rccl_init (collective).  dasr_allreduce / dasr_broadcast are in place, asynchronous on `stream`; return an ncclResult_t. */
int dasr_rccl_unique_id(void* id128);
int dasr_rccl_init(const void* id128, int32_t rank, int32_t world, void** comm_out);
int dasr_allreduce(void* comm, float* buf, int64_t count, void* stream);
int dasr_broadcast(void* comm, float* buf, int64_t count, int32_t root, void* stream);
int dasr_rccl_destroy(void* comm);

/* ---- profiling session (bench.py `roofline`) -----------------------------------------------------------
 * Between dasr_prof_begin and dasr_prof_end every kernel launch of the library (up to `capacity`) carries its own start/stop
 * events on its launch stream (hipExtLaunchKernel: the dispatch's begin/end timestamps, what rocprofv3 --kernel-trace prints).
 * dasr_prof_end synchronises the device and returns, per launch in issue order: duration in microseconds, the algorithmic
 * flops / bytes of the op it belongs to (dasr_op.flops / .bytes, attributed to the op's first launch), the op kind (bits 0-7; bits 8-15: the
 * plan builder's time-bucket tag dasr_op.i[7], which no kernel reads) and a
 * static string naming the kernel variant.  Returns the number of records written (<= max_out) or a negative error. */
int dasr_prof_begin(int32_t capacity);
int dasr_prof_end(int32_t max_out, float* us_out, double* flops_out, double* bytes_out, int32_t* op_out, const char** tag_out);
/* Restricts the NEXT profiling sessions to launches whose tag (kernel name / launcher signature) contains `substr` (NULL or "": every launch again).  A session over
 * one kernel family costs two events per matching launch and nothing else: bench.py times the dominant kernel this way over ordinary steps (round 6: a step in which
 * EVERY launch carries events runs ~6 % slower and over-reported the chained launches by 7-11 %).  DASR_EINVAL while a session is open. */
int dasr_prof_filter(const char* substr);

/* ---- diagnostics ----------------------------------------------------------------------------------*/
int dasr_abi_version(void);
/* Frees the scratch rows of the deterministic grid sums (one per (device, loss accumulator, stream), allocated on first use and otherwise kept for the life of the
 * process).  The caller guarantees that no launch of this library is in flight.  Returns 0. */
int dasr_red_release(void);
/* 1 if ds_read_b64_tr_b16 has the lane mapping the wgrad kernel assumes on this device, 0 if not, <0 on error.
 * Must be called once per process before dasr_wgrad (it also selects the wgrad gather mode). */
int dasr_probe_tr16(void* stream);
/* naive fp32 direct convolution (one thread per output), used by tests as an on-device cross-check */
int dasr_conv_naive(const dasr_conv_params* p, const float* w_ref /* [cout][cin][kh][kw] */, void* stream);

#ifdef __cplusplus
}
#endif
#endif
