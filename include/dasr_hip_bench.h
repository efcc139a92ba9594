/* dasr_hip_bench.h -- micro-benchmark probes of libdasr_bench.so (dasr_amd/csrc/bench_probes.hip).
 *
 * NOT part of the product library: libdasr_hip.so exports nothing declared here.  bench.py uses the MFMA-only probes to report the
 * dense-MFMA rate the box sustains at the clock its power state allows next to the spec peak; scripts/micro_*.py use the rest.
 * The wrong-result ablation instantiations of the dense conv kernel live in a third library (libdasr_hip_ablate.so =
 * the product sources compiled with -DDASR_BENCH, `python -m dasr_amd.build --ablate`), reachable through dasr_set_tuning(1, 100 + bits).
 */
#ifndef DASR_HIP_BENCH_H
#define DASR_HIP_BENCH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* MFMA-only micro-benchmark (every SIMD issuing back-to-back v_mfma_f32_32x32x16_bf16): the dense bf16 rate this box sustains at
 * the clock its power state allows, in TFLOP/s; `iters` MFMA quads per wave (e.g. 20000 ~ 1.5 ms).  Synchronises the stream. */
int dasr_probe_mfma_peak(int32_t iters, float* tflops_out, void* stream);
/* the same MFMA-only stream with operands that toggle: mode 0 bf16 / 1 f16 fragments from a per-lane random generator, 2 all-zero bf16
 * operands; the differences are the clock the power management allows under that switching activity (scripts/micro_mfma.py) */
int dasr_probe_mfma_data(int32_t iters, int32_t mode, float* tflops_out, void* stream);

/* Neighbour-flag synchronisation micro-benchmark (scripts/micro_sync.py): `blocks` co-resident workgroups rewrite a tile of
 * `tile_words` words per stage, publish a flag, wait for two ring neighbours (`nb_stride` 8: same XCD, 1: other XCDs) and read their tiles.
 * scope 0: no synchronisation (floor), 1: agent-scope release / acquire fences, 2: workgroup-scope fences + L2-served (sc1) flag and data
 * accesses.  Returns microseconds per stage, whether a wait timed out, and the number of stale neighbour reads.  Synchronises the stream. */
int dasr_probe_tile_sync(int32_t blocks, int32_t stages, int32_t nb_stride, int32_t scope, int32_t tile_words, float* us_per_stage,
                         int32_t* timed_out, int32_t* stale_reads, void* stream);

/* `blocks` workgroups (256 threads) that only hold their slots for `micros` microseconds; asynchronous on `stream`.  Tests use it as a stand-in for a
 * collective's kernels on the communication stream (the chained trunk launches of the product library need every workgroup slot of the device). */
int dasr_probe_spin(int32_t blocks, int32_t micros, void* stream);

/* Store-path probe (round 6, scripts/micro_store.py): `blocks` workgroups of 512 threads each write `kb` KiB (multiple of 8) of their own region of `buf` (blocks * kb KiB) with
 * 16-byte-per-lane stores, mode 0 plain / 1 sc1 (write-through), `reps` bursts separated by `gap` x s_sleep(16); cyc_out[block] = cycles spent in the bursts (first store
 * issued .. last store acknowledged).  Synchronises the stream. */
int dasr_probe_store(void* buf, int32_t blocks, int32_t kb, int32_t mode, int32_t reps, int32_t gap, unsigned long long* cyc_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
