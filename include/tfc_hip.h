/*
 * tfc_hip.h — C ABI of the MI355X (gfx950) entropy-coding + transform hot path.
 *
 * This is the drop-in boundary: every entry point replaces one CPU op kernel
 * (or one stock-TF composition) of tensorflow/compression; the reference
 * interface it stands in for is cited as file:line under /root/reference/
 * tensorflow_compression/.  Plain pointers and sizes only — no torch / TF types.
 *
 * Conventions
 *   - Pointers marked DEV are HIP device pointers, HOST are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls
 *     are asynchronous on that stream unless the comment says "synchronises".
 *   - Return value 0 = OK; non-zero = InvalidArgument-class failure whose text
 *     (same substrings the reference's OP_REQUIRES messages carry) is returned
 *     by tfc_last_error() on the calling thread.
 *   - A handle must not be used from two host threads at once (the reference
 *     handles are single-consumer too, cc/ops/range_coder_ops.cc:94-95).
 */
#ifndef TFC_HIP_H_
#define TFC_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tfc_tables tfc_tables;
typedef struct tfc_encoder tfc_encoder;
typedef struct tfc_decoder tfc_decoder;

/* Library identity; bumps when the ABI changes (compression_amd/_lib.py refuses a library whose
 * version is not the one it was written against).  2: round-3 ABI (per-handle modes, *_many entry
 * points, stream-ordered finalize, CU-masked streams). */
#define TFC_ABI_VERSION 2
int tfc_abi_version(void);
/* Text of the last failure on this thread ("" if none). */
const char* tfc_last_error(void);

/* Diagnostics for bench.py: when enabled, the library brackets its main kernels
 * ("enc_kernel", "dec_kernel", "gdn_forward", ...) with HIP events on the
 * launch stream; tfc_profile_query returns the accumulated time and launch
 * count of one kernel name (synchronises on the recorded events).  Enabling
 * resets the counters. */
void tfc_profile_enable(int on);
int tfc_profile_query(const char* kernel, double* total_ms, int64_t* launches);
/* Diagnostics for the tests: encode / decode launches of throughput-mode handles that went to the pipelined
 * kernels (csrc/range_pipe.h) since the library was loaded, and workgroups of the lane-per-stream kernels that
 * ran behind them as the fallback for a job they gave up on (synchronises the device).  Either may be null. */
int tfc_pipe_counters(int64_t* launches, int64_t* fallback_blocks);
/* A/B and test switch of the pipelined decoder (csrc/range_pipe.h): which table image its chain kernel runs on —
 * 0 by launch (the full image where it fits a CU and the launch's chain waves all find one, else the compact image),
 * 1 the full image (one bit per quotient value), 2 the compact image (every second bound at pair resolution: half the
 * bitmaps, ~10 % more cycles per row) — and how many chain waves share a workgroup's copy of it (0: by launch size).
 * Same symbols either way.  Returns the previous format; the environment (TFC_PIPE_FORMAT = full | pairs,
 * TFC_PIPE_WAVES) sets the initial values. */
int tfc_set_pipe_format(int format, int waves_per_workgroup);

/* Kernel family of a coder handle (no reference counterpart: the reference's ops shard streams over the
 * intra-op thread pool, range_coder_kernels.cc:212-267).  The bytes / symbols produced are identical.
 *   TFC_MODE_LATENCY     one code stream per 64-lane wave: shortest time for one call on an idle GPU
 *                        (elems x ~90 cycles encode, ~160 decode), 15-28 vector instructions per symbol;
 *   TFC_MODE_THROUGHPUT  one code stream per LANE: ~1 instruction per symbol and 64 streams, elems x
 *                        ~400 cycles per call — for callers that keep many independent calls in flight
 *                        on different HIP streams, or code thousands of streams per call;
 *   TFC_MODE_AUTO        by the stream count of the handle (throughput kernels from 4096 streams).
 * Tables the throughput kernels cannot hold (LDS image > 160 KB, rows with zero-width symbols) fall
 * back to the latency kernels.  The process-wide default applies to handles left at TFC_MODE_AUTO;
 * its initial value comes from the environment variable TFC_DEFAULT_MODE = latency | throughput. */
#define TFC_MODE_AUTO 0
#define TFC_MODE_LATENCY 1
#define TFC_MODE_THROUGHPUT 2
int tfc_set_default_mode(int mode);
int tfc_get_default_mode(void);

/* Process-wide hint: 1 = the caller keeps other kernels (the transforms of other batches) in flight beside the coder's,
 * 0 (default) = a coder call has the chip to itself.  Shared: handles of 512 streams and more are created with two
 * waves per SIMD (half the CUs; a convolution workgroup cannot use a CU that hosts a coder wave).  Takes effect for
 * handles created afterwards.  No reference counterpart (TensorFlow's executor owns such placement). */
int tfc_set_chip_shared(int shared);     /* -> the previous value */
/* Device memory the library keeps for reuse: every buffer a call or handle releases goes to per-size free lists instead
 * of back to the driver (hipFreeAsync behind a running kernel holds the calling thread until that kernel ends).
 * tfc_cache_bytes: bytes cached now; tfc_cache_trim: returns the blocks whose last use has completed to the driver
 * (-> bytes released).  TFC_CACHE_LIMIT_MB bounds the cache (default 65536).  No reference counterpart (TensorFlow's
 * allocator owns the ops' scratch memory). */
/* Elementwise passes of the model pipelines (csrc/elementwise.hip), one kernel each; dtype 0 = float32, 1 = bfloat16.
 * tfc_image_to_unit: y[i] = dtype(x[i]) / 255 for uint8 x (models/bls2017.py:164-170, bmshj2018.py:219-224).
 * tfc_unit_to_image: y[i] = saturate_cast<uint8>(round_half_even(dtype(x[i] * 255))) (bls2017.py:186-190).
 * tfc_index_prepare: out[i] = int32(min(max(indexes[i], 0), num_tables - 1)), cast toward zero
 * (continuous_indexed.py:272-296: `_normalize_indexes` and the int32 cast of `_flatten_indexes`, one index range). */
int tfc_image_to_unit(const void* x, void* y, int dtype, int64_t n, void* stream);
int tfc_unit_to_image(const void* x, int dtype, void* y, int64_t n, void* stream);
int tfc_index_prepare(const void* indexes, int dtype, int32_t* out, int64_t n, int num_tables, void* stream);
/* Spatial padding of an NHWC tensor [n, h, w, c] of 2- or 4-byte elements into y [n, top + h + bottom, left + w + right, c]:
 * reflect = 0 zeros, 1 mirror without repeating the edge (tf.pad "CONSTANT" / "REFLECT") — the pre-pad of
 * SignalConv2D's `same_reflect` / pre-padded `same_zeros` modes (python/layers/signal_conv.py:880-893). */
int tfc_pad2d(const void* x, void* y, int elem_bytes, int64_t n, int64_t h, int64_t w, int64_t c, int top, int bottom,
              int left, int right, int reflect, void* stream);
int tfc_cache_bytes(long long* bytes);
int tfc_cache_trim(long long* released);

/* ------------------------------------------------------------------------ */
/* CDF tables                                                               */
/* ------------------------------------------------------------------------ */

/* Validates `lookup` and uploads it in the device layout the coders use.
 * Replaces ScanCDF / IndexCDFVector / IndexCDFMatrix
 * (cc/kernels/range_coder_kernels.cc:101-164): rank 1 = ragged concatenation
 * of rows [+-precision, 0, ..., 1<<precision]; rank 2 = [rows, cols], rows
 * padded with 1<<precision.  Negative precision enables the Elias-gamma
 * escape for that row.  `lookup` is HOST (rank 1: cols ints; rank 2:
 * rows*cols ints).  Synchronises on `stream` for the upload. */
int tfc_tables_create(const int32_t* lookup, int rank, int64_t rows, int64_t cols,
                      void* stream, tfc_tables** out);
int64_t tfc_tables_count(const tfc_tables* t);
void tfc_tables_destroy(tfc_tables* t);

/* ------------------------------------------------------------------------ */
/* Multi-stream range encoder (one independent code stream per handle       */
/* element)                                                                 */
/* ------------------------------------------------------------------------ */

/* CreateRangeEncoder — cc/ops/range_coder_ops.cc:28-67,
 * cc/kernels/range_coder_kernels.cc:484-507.  `streams` = number of elements
 * of the handle shape. */
int tfc_encoder_create(const tfc_tables* tables, int64_t streams, void* stream,
                       tfc_encoder** out);

/* n handles of `streams` streams each with one allocation and one launch (per-handle driver calls and
 * small kernels are a measurable part of a step once the coding calls of a group share a launch);
 * out: HOST tfc_encoder*[n].  Each handle is destroyed on its own. */
int tfc_encoder_create_many(const tfc_tables* tables, int64_t streams, int n, void* stream,
                            tfc_encoder** out);

/* Selects the kernel family (TFC_MODE_*); only before the first encode call on the handle. */
int tfc_encoder_set_mode(tfc_encoder* e, int mode);
/* on = 1: encode calls never synchronise; an "index=… / value=… not in range" failure is recorded
 * on the device and returned by tfc_encoder_finalize / tfc_encoder_status instead of by the encode
 * call that met it (throughput-mode handles only; latency-mode calls validate before coding and
 * always report at once).  After such a failure the handle's streams are meaningless. */
int tfc_encoder_set_deferred_errors(tfc_encoder* e, int on);

/* EntropyEncodeChannel (index == NULL) / EntropyEncodeIndex —
 * cc/ops/range_coder_ops.cc:69-127, cc/kernels/range_coder_kernels.cc:191-272,
 * 290-322.  value/index: DEV int32 [streams, elems] row-major.  Appends to
 * every stream; may be called repeatedly on one handle.  Range errors
 * ("index=… not in range", "value=… not in range"): latency-mode handles run a
 * validation pass before anything is appended and synchronise once to read its
 * result (and the exact output bound); throughput-mode handles code in one
 * kernel and synchronise once afterwards to read the error word, unless
 * tfc_encoder_set_deferred_errors(e, 1). */
int tfc_encoder_encode(tfc_encoder* e, const int32_t* value, const int32_t* index,
                       int64_t elems, void* stream);

/* The same for n independent handles (same tables, same stream count, same elems; value[k] / index[k] of
 * handle k) in ONE launch where the handles use the throughput kernels (TFC_MODE_THROUGHPUT, or
 * TFC_MODE_AUTO with n * streams >= 4096): the hardware overlaps only a handful of kernels however many
 * HIP streams carry them, and a 512-stream call is 8 waves, so independent calls fill the chip only
 * as one grid.  Results are exactly those of n tfc_encoder_encode calls; other handles are coded by such
 * calls one after the other. */
int tfc_encoder_encode_many(int n, tfc_encoder* const* e, const int32_t* const* value,
                            const int32_t* const* index, int64_t elems, void* stream);

/* Fused quantise + encode, channel mode:
 *   sym = int32(rint(y - qoffset[c])) - cdf_offset[c],  c = j mod channels
 * i.e. ContinuousBatchedEntropyModel.compress's prologue
 * (python/entropy_models/continuous_batched.py:370-380) folded into the
 * coder's load.  y: DEV [streams, elems], dtype 0 = float32, 1 = bfloat16,
 * 2 = float16.  qoffset: DEV float32 [channels] or NULL; cdf_offset: DEV int32
 * [channels]. */
int tfc_encoder_encode_quantized(tfc_encoder* e, const void* y, int dtype,
                                 const float* qoffset, const int32_t* cdf_offset,
                                 int64_t channels, int64_t elems, void* stream);

/* Fused quantise + encode, index mode
 * (python/entropy_models/continuous_indexed.py:355-386):
 *   sym = int32(rint(y)) - cdf_offset[index]. */
int tfc_encoder_encode_quantized_indexed(tfc_encoder* e, const void* y, int dtype,
                                         const int32_t* index, const int32_t* cdf_offset,
                                         int64_t elems, void* stream);

/* ... for n independent handles (same tables, same geometry) as ONE coding launch: tfc_encoder_encode_quantized
 * x tfc_encoder_encode_many.  y: HOST array of n DEV pointers.  Handles of the throughput family quantise in
 * an elementwise pass of their own (HBM-bound) and code int32 symbols with the hand-scheduled blocks — the
 * fused call costs what int32 channel mode costs; other handles go one by one. */
int tfc_encoder_encode_quantized_many(int n, tfc_encoder* const* e, const void* const* y, int dtype,
                                      const float* qoffset, const int32_t* cdf_offset,
                                      int64_t channels, int64_t elems, void* stream);
/* ... and in index mode (tfc_encoder_encode_quantized_indexed for n handles): the main latents of several batches
 * of a hyperprior model (continuous_indexed.py:355-386) as one launch.  y, index: HOST arrays of n DEV pointers. */
int tfc_encoder_encode_quantized_indexed_many(int n, tfc_encoder* const* e, const void* const* y, int dtype,
                                              const int32_t* const* index, const int32_t* cdf_offset,
                                              int64_t elems, void* stream);

/* EntropyEncodeFinalize — cc/ops/range_coder_ops.cc:129-135,
 * cc/kernels/range_coder_kernels.cc:274-287 + cc/lib/range_coder.cc:266-307.
 * Flushes every stream, packs the streams back to back and returns the total
 * byte count.  Synchronises. */
int tfc_encoder_finalize(tfc_encoder* e, void* stream, int64_t* total_bytes);

/* The same without any host synchronisation: the packed blob is allocated at the slabs' capacity and
 * the total (= offsets[streams]) stays on the device; tfc_encoder_result is valid afterwards, a
 * decoder can be created on it at once (stream-ordered).  tfc_encoder_status synchronises, returns
 * deferred failures and the total byte count (total_bytes may be NULL). */
int tfc_encoder_finalize_device(tfc_encoder* e, void* stream);
/* ... of n handles: three launches in all where every handle holds one piece from the throughput kernels
 * (what tfc_encoder_encode_many leaves), otherwise handle by handle. */
int tfc_encoder_finalize_device_many(int n, tfc_encoder* const* e, void* stream);
int tfc_encoder_status(tfc_encoder* e, void* stream, int64_t* total_bytes);

/* After finalize: device views of the packed result (owned by the handle):
 * blob DEV uint8 [total_bytes], offsets DEV int64 [streams + 1]. */
int tfc_encoder_result(const tfc_encoder* e, const uint8_t** blob, const int64_t** offsets);
/* Bytes the device blob was allocated with (>= offsets[streams]): what a caller may map before it
 * knows the total (after tfc_encoder_finalize_device). */
int tfc_encoder_capacity(const tfc_encoder* e, int64_t* bytes);
/* After finalize: copies the result out.  dst_on_device selects hipMemcpy
 * direction.  Synchronises when copying to the host. */
int tfc_encoder_read(const tfc_encoder* e, uint8_t* blob_dst, int64_t* offsets_dst,
                     int dst_on_device, void* stream);
void tfc_encoder_destroy(tfc_encoder* e);

/* ------------------------------------------------------------------------ */
/* Multi-stream range decoder                                               */
/* ------------------------------------------------------------------------ */

/* CreateRangeDecoder — cc/ops/range_coder_ops.cc:137-153,
 * cc/kernels/range_coder_kernels.cc:597-619.  blob/offsets describe `streams`
 * byte strings (offsets int64 [streams+1]); src_on_device says where they
 * live.  Host input is copied to the device.  Device input is BORROWED: like
 * the reference, which ref-holds the tensor and reads it in place
 * (range_coder_kernels.cc:475-478; "caller must make sure `source` outlives",
 * cc/lib/range_coder.h:74-77), the buffers must stay valid and unchanged until
 * the decoder is destroyed. */
int tfc_decoder_create(const tfc_tables* tables, const uint8_t* blob, const int64_t* offsets,
                       int64_t streams, int src_on_device, void* stream, tfc_decoder** out);

/* n decoders on the device-resident strings of n finalized encoders (borrowed in place, like
 * tfc_decoder_create with src_on_device = 1): one allocation, one launch; out: HOST tfc_decoder*[n]. */
int tfc_decoder_create_many(const tfc_tables* tables, int n, tfc_encoder* const* from, void* stream,
                            tfc_decoder** out);

/* Selects the kernel family (TFC_MODE_*) of the following decode calls. */
int tfc_decoder_set_mode(tfc_decoder* d, int mode);

/* EntropyDecodeChannel (index == NULL) / EntropyDecodeIndex —
 * cc/ops/range_coder_ops.cc:155-237, cc/kernels/range_coder_kernels.cc:360-429,
 * 449-471.  index DEV int32 [streams, elems] or NULL; out DEV int32
 * [streams, elems].  Continues where the previous decode call stopped. */
int tfc_decoder_decode(tfc_decoder* d, const int32_t* index, int32_t* out, int64_t elems,
                       void* stream);

/* n independent handles in one launch (see tfc_encoder_encode_many); index may be NULL. */
int tfc_decoder_decode_many(int n, tfc_decoder* const* d, const int32_t* const* index,
                            int32_t* const* out, int64_t elems, void* stream);

/* Fused decode + dequantise (continuous_batched.py:416-422 /
 * continuous_indexed.py:411-417):
 *   y = float(sym + cdf_offset[c or index]) + (qoffset ? qoffset[c] : 0)
 * written as `dtype`.  Channel mode when index == NULL (then `channels` is the
 * table count), else index mode (qoffset must be NULL). */
int tfc_decoder_decode_dequantized(tfc_decoder* d, const int32_t* index, void* y, int dtype,
                                   const float* qoffset, const int32_t* cdf_offset,
                                   int64_t channels, int64_t elems, void* stream);

/* ... for n independent handles as ONE coding launch (channel mode; see tfc_encoder_encode_quantized_many).
 * y: HOST array of n DEV pointers. */
int tfc_decoder_decode_dequantized_many(int n, tfc_decoder* const* d, void* const* y, int dtype,
                                        const float* qoffset, const int32_t* cdf_offset,
                                        int64_t channels, int64_t elems, void* stream);
/* ... and in index mode (continuous_indexed.py:388-417 for n handles).  index, y: HOST arrays of n DEV pointers. */
int tfc_decoder_decode_dequantized_indexed_many(int n, tfc_decoder* const* d, const int32_t* const* index,
                                                void* const* y, int dtype, const int32_t* cdf_offset,
                                                int64_t elems, void* stream);

/* EntropyDecodeFinalize — cc/ops/range_coder_ops.cc:239-246,
 * cc/lib/range_coder.h:144-169.  ok: HOST uint8 [streams] (1 = the weak
 * end-of-stream check passed).  Also reports a deferred "index=… not in
 * range" failure of a previous decode call.  Synchronises. */
int tfc_decoder_finalize(tfc_decoder* d, uint8_t* ok, void* stream);
/* The same in two stream-ordered halves: the weak check into ok DEV uint8 [streams] without
 * synchronising, and the deferred index failure (synchronises). */
int tfc_decoder_finalize_device(tfc_decoder* d, uint8_t* ok, void* stream);
/* ... of n handles with the same stream count: ok DEV uint8 [n, streams]. */
int tfc_decoder_finalize_device_many(int n, tfc_decoder* const* d, uint8_t* ok, void* stream);
int tfc_decoder_status(tfc_decoder* d, void* stream);
void tfc_decoder_destroy(tfc_decoder* d);

/* ------------------------------------------------------------------------ */
/* Deprecated single-stream ops                                             */
/* ------------------------------------------------------------------------ */

/* RangeEncode — cc/ops/range_coding_ops.cc:30-90,
 * cc/kernels/range_coding_kernels.cc:175-275 (+ MergeAxes,
 * range_coding_kernels_util.cc:34-91).  data DEV int16 with shape
 * data_shape[nd]; cdf DEV int32 with shape cdf_shape[nd+1], broadcastable to
 * data_shape + [width].  Returns the encoded string in a malloc'ed HOST buffer
 * (*out, *out_len) which the caller frees with tfc_free().  Synchronises. */
int tfc_range_encode(const int16_t* data, const int64_t* data_shape, int nd,
                     const int32_t* cdf, const int64_t* cdf_shape, int nc,
                     int precision, int debug_level, void* stream,
                     uint8_t** out, int64_t* out_len);

/* RangeDecode — cc/ops/range_coding_ops.cc:92-124,
 * cc/kernels/range_coding_kernels.cc:277-379.  encoded HOST bytes; out DEV
 * int16 of shape out_shape[nd]. */
int tfc_range_decode(const uint8_t* encoded, int64_t encoded_len, const int64_t* out_shape,
                     int nd, const int32_t* cdf, const int64_t* cdf_shape, int nc,
                     int precision, int debug_level, void* stream, int16_t* out);

/* Deprecated UnboundedIndexRangeEncode / UnboundedIndexRangeDecode —
 * cc/kernels/unbounded_index_range_coding_kernels.cc:146-249, 259-367 (checks :54-143).  ONE stream for
 * the whole tensor: element i uses row index[i] of cdf DEV int32 [rows, width] (the first cdf_size[row]
 * entries are the table), is shifted by offset[row], and values outside [0, cdf_size - 2) are coded as the
 * row's last symbol followed by a variable-length code in `overflow_width`-bit digits.  data / index /
 * out DEV int32 with `total` elements; cdf_size, offset DEV int32 [rows].  debug_level 1 validates index,
 * cdf_size and the tables (same messages as the reference).  Encode returns a malloc'ed HOST buffer the
 * caller frees with tfc_free(); decode takes HOST bytes.  Both synchronise. */
int tfc_unbounded_index_range_encode(const int32_t* data, const int32_t* index, int64_t total,
                                     const int32_t* cdf, int64_t rows, int64_t width,
                                     const int32_t* cdf_size, const int32_t* offset, int precision,
                                     int overflow_width, int debug_level, void* stream, uint8_t** out,
                                     int64_t* out_len);
int tfc_unbounded_index_range_decode(const uint8_t* encoded, int64_t encoded_len, const int32_t* index,
                                     int64_t total, const int32_t* cdf, int64_t rows, int64_t width,
                                     const int32_t* cdf_size, const int32_t* offset, int precision,
                                     int overflow_width, int debug_level, int32_t* out, void* stream);

/* StochasticRound — cc/ops/quantization_ops.cc:21-44, cc/kernels/quantization_kernels.cc:47-96.
 * outputs[i] = floor(inputs[i] / step_size), plus 1 when the i-th draw of ONE xoshiro256+ generator
 * (seeded from `seed` with the C++ standard's seed_seq; 24-bit draws in [0, 1)) is below the fractional
 * part — the same draw for the same flat position as the reference's serial loop, so results are
 * identical for identical seeds.  inputs DEV float32 (dtype 0) / bfloat16 (1) / float16 (2), n elements;
 * seed HOST int32[seed_len]; seed_len 0 seeds from the clock (quantization_kernels.cc:75-81); outputs DEV
 * int32.  Asynchronous on `stream`. */
int tfc_stochastic_round(const void* inputs, int dtype, int64_t n, float step_size,
                         const int32_t* seed, int64_t seed_len, int32_t* outputs, void* stream);

void tfc_free(void* p);

/* ------------------------------------------------------------------------ */
/* PmfToQuantizedCdf                                                        */
/* ------------------------------------------------------------------------ */

/* cc/ops/pmf_to_cdf_ops.cc:28-57, cc/kernels/pmf_to_cdf_kernels.cc:58-208.
 * pmf DEV float32 [rows, n] -> cdf DEV int32 [rows, n+1]; every symbol >= 1,
 * cdf[:,0] = 0, cdf[:,n] = 1 << precision.  Which of several symbols with EQUAL penalty is adjusted
 * follows the order libstdc++'s std::sort leaves them in (csrc/sort_order.h reproduces its introsort),
 * i.e. the tables of the reference built with libstdc++; the reference itself disclaims portability of
 * that order across standard libraries (pmf_to_cdf_ops.cc:45-49). */
int tfc_pmf_to_quantized_cdf(const float* pmf, int64_t rows, int64_t n, int precision,
                             int32_t* cdf, void* stream);

/* A whole model's range-coding tables in one launch — python/entropy_models/continuous_base.py:217-296
 * (`_build_tables`) from the sampled PMFs on: row r of pmf DEV [rows, stride] holds lengths[r] (DEV int32, >= 1)
 * probabilities; the kernel appends overflow = max(1 - sum(pmf[:length]), 0) (continuous_base.py:277-279; float32, in
 * the fixed order csrc/pmf_to_cdf.hip states), runs PmfToQuantizedCdf (pmf_to_cdf_kernels.cc:159-208) on the
 * length + 1 values and writes [-precision, cdf[0 .. length + 1]] at out[offsets[r]] (offsets DEV int64: the caller's
 * prefix sums of length + 3; out DEV int32 [sum]).  max_length = the largest length (sizes the kernel's LDS). */
int tfc_build_tables(const float* pmf, int64_t rows, int64_t stride, const int32_t* lengths, const int64_t* offsets,
                     int64_t max_length, int precision, int32_t* out, void* stream);
/* The same with the overflow mass of every row given (overflow DEV float32 [rows]; null: as tfc_build_tables): the
 * reference forms max(1 - reduce_sum(p), 0) in the PRIOR's dtype and casts to float32 afterwards
 * (continuous_base.py:277-279), so a float64 / bfloat16 prior's caller sums in that arithmetic itself. */
int tfc_build_tables_overflow(const float* pmf, int64_t rows, int64_t stride, const int32_t* lengths,
                              const int64_t* offsets, int64_t max_length, int precision, const float* overflow,
                              int32_t* out, void* stream);
/* helpers.estimate_tails (python/distributions/helpers.py:29-104) for a deep factorized prior
 * (python/distributions/deep_factorized.py:166-246), whole iteration on the device: for every target t (DEV float32
 * [num_targets]) and channel c, the x where the channel's logits of the cumulative reach t — out DEV [num_targets,
 * channels]; iterations DEV int32 [num_targets] or null.  params DEV [channels, params_per_channel]: the
 * reparameterised MLP weights in the layout of tfc_factorized_bits_forward. */
int tfc_deep_factorized_tails(const float* params, int64_t channels, int64_t params_per_channel, int layers, int width,
                              const float* targets, int num_targets, float* out, int* iterations, void* stream);

/* ------------------------------------------------------------------------ */
/* GDN / IGDN                                                               */
/* ------------------------------------------------------------------------ */

/* GDN.call — python/layers/gdn.py:371-421 with channels_last layout:
 *   u = |x|^alpha (alpha_mode 1) or x^2 (alpha_mode 2); rectify => x = max(x,0) first
 *   n_i = beta_i + sum_j gamma[j,i] u_j
 *   y_i = x_i / n_i^eps  (inverse=0)   or   x_i * n_i^eps (inverse=1)
 * eps_mode 0: eps = 1; 1: eps = 0.5.  x,y DEV [pixels, channels] dtype (0 f32,
 * 1 bf16); beta DEV f32 [channels]; gamma DEV f32 [channels(in j), channels(out i)]. */
int tfc_gdn_forward(const void* x, void* y, int dtype, int64_t pixels, int64_t channels,
                    const float* beta, const float* gamma, int inverse, int rectify,
                    int alpha_mode, int eps_mode, void* stream);

/* Inference form: the kernels' LDS image of (beta, gamma) built ONCE (tfc_gdn_params_create launches the small
 * preparation kernel the plain entry point launches per call) and reused by every tfc_gdn_forward_prepared call
 * while the parameters do not change — python/layers/gdn.py holds beta / gamma as variables, constant outside
 * training.  channels / dtype as for tfc_gdn_forward. */
typedef struct tfc_gdn_params tfc_gdn_params;
int tfc_gdn_params_create(const float* beta, const float* gamma, int64_t channels, int dtype, void* stream,
                          tfc_gdn_params** out);
void tfc_gdn_params_destroy(tfc_gdn_params* p);
int tfc_gdn_forward_prepared(const tfc_gdn_params* params, const void* x, void* y, int64_t pixels,
                             int inverse, int rectify, int alpha_mode, int eps_mode, void* stream);

/* The same layer with general (e.g. learned) exponents — python/layers/gdn.py:345-369 creates alpha
 * (>= 1) and epsilon (>= 1e-6) as scalar parameters when the constructor gets None, and :386-387 /
 * :411-412 then take `inputs ** alpha` and `norm_pool ** epsilon` with tf.pow:
 *   u = x^alpha (rectify => x = max(x,0) first; a negative x gives NaN unless alpha is an integer, as tf.pow does)
 *   y_i = x_i / n_i^epsilon  or  x_i * n_i^epsilon.
 * alpha, epsilon: positive finite host scalars.  Forward only: the gradients of this variant (which include
 * d/dalpha and d/depsilon) are composed from device tensor ops by the host layer, not by this library. */
int tfc_gdn_forward_general(const void* x, void* y, int dtype, int64_t pixels, int64_t channels,
                            const float* beta, const float* gamma, int inverse, int rectify,
                            float alpha, float epsilon, void* stream);

/* Backward of tfc_gdn_forward (the reference relies on TF autodiff).  g = dL/dy.
 * Outputs: dx (dtype) and float32 accumulators dbeta [channels], dgamma
 * [channels, channels] which are ADDED to (caller zeroes them). */
int tfc_gdn_backward(const void* x, const void* g, void* dx, int dtype, int64_t pixels,
                     int64_t channels, const float* beta, const float* gamma, int inverse,
                     int rectify, int alpha_mode, int eps_mode, float* dbeta, float* dgamma,
                     void* stream);

/* ------------------------------------------------------------------------ */
/* SignalConv2D (same_zeros, explicit padding, NHWC, non-separable)         */
/* ------------------------------------------------------------------------ */

/* _correlate_down_explicit — python/layers/signal_conv.py:663-690:
 * cross-correlation with zero padding (k/2, (k-1)/2) and stride `stride`;
 * out = ceil(in / stride), first output aligned with the first input.
 * x DEV [N,H,W,Cin] (dtype: 0 f32, 1 bf16), w DEV float32 [kh,kw,Cin,Cout] (HWIO, the
 * layer's `kernel`), bias DEV f32 [Cout] or NULL, y DEV [N,ceil(H/s),ceil(W/s),Cout]
 * (same dtype as x).  activation: 0 none, 1 ReLU (fused).  Cin must be a
 * multiple of 16 or <= 4. */
int tfc_conv2d_down(const void* x, const void* w, const float* bias, void* y, int dtype,
                    int64_t n, int64_t h, int64_t wd, int64_t cin, int64_t cout,
                    int kh, int kw, int stride, int activation, void* stream);

/* _up_convolve_transpose_explicit — python/layers/signal_conv.py:778-847 with
 * extra_pad_end=True: zero-insertion upsampling by `stride` followed by a true
 * convolution centred at k/2, i.e. y[q*s + phi] = sum_i x[i] * w[phi + (q-i)*s + k/2];
 * out = in * stride.  w is the layer's own HWIO kernel [kh,kw,Cin,Cout]; the
 * library does the phase split.  stride 1 gives the flipped-kernel correlation
 * the reference uses for `corr=False` without upsampling (signal_conv.py:865-870). */
int tfc_conv2d_up(const void* x, const void* w, const float* bias, void* y, int dtype,
                  int64_t n, int64_t h, int64_t wd, int64_t cin, int64_t cout,
                  int kh, int kw, int stride, int activation, void* stream);

/* The layer with GDN / IGDN as its activation — SignalConv2D(activation=GDN(...)), python/layers/signal_conv.py:948-950
 * applying python/layers/gdn.py:371-421 to the convolution's output, as models/bls2017.py:61-91 and bmshj2018.py build
 * their transforms — in ONE kernel where the convolution kernel that takes the layer holds all output channels of its
 * pixels (the third-generation bfloat16 kernel: transposed 5x5 stride-2 layers and small stride-2 maps, Cout 128 / 192):
 * y = conv(x) + bias rounded to bfloat16, then y / (beta + gamma^T |y|) (inverse = 0) or y * (...) (inverse = 1), alpha =
 * epsilon = 1, no rectification; `gdn` = tfc_gdn_params_create of the layer's (beta, gamma) for bfloat16.  *fused = 1:
 * done; *fused = 0: `y` holds the convolution only (another kernel took the layer) and the caller applies
 * tfc_gdn_forward_prepared to it.  up = 0: tfc_conv2d_down's geometry, 1: tfc_conv2d_up's. */
int tfc_conv2d_gdn(const void* x, const void* w, const float* bias, void* y, int dtype,
                   int64_t n, int64_t h, int64_t wd, int64_t cin, int64_t cout, int kh,
                   int kw, int stride, int up, const tfc_gdn_params* gdn, int inverse, int* fused,
                   void* stream);

/* Inference: the layer's weights do not change between calls.  Every convolution kernel reads the float32 HWIO kernel
 * as fragments in its own order, packed by a small kernel in front of it (60-90 us, four to nine a model step).
 * tfc_conv2d_weights_key(key) names the VALUE of `w` for the NEXT tfc_conv2d_* call of the calling thread (key != 0,
 * chosen by the caller: one number per distinct weight tensor value, never reused for another): that call packs the
 * fragments once per (key, kernel, geometry) and later calls with the same key reuse them, from any thread or stream.
 * Without a key every call packs (training; the reference has no such notion: TF folds constants in its graph).
 * tfc_conv2d_drop_weights(key) releases them (drains the device first): when the weights changed or the layer dies. */
void tfc_conv2d_weights_key(uint64_t key);
int tfc_conv2d_drop_weights(uint64_t key);

/* Weight gradient of either direction (the reference relies on TF autodiff of
 * signal_conv.py:663-690 / 778-847).  G[t][ca][cb] = sum_{n,q} A[n, q*stride + t - k/2, ca] *
 * B[n, q, cb] with zeros outside A; q runs over B's grid.
 *   analysis  y = tfc_conv2d_down(x, w):  dw = G            with A = x  [n,ha,wa,ca=Cin],  B = dy [n,hb,wb,cb=Cout]
 *   synthesis y = tfc_conv2d_up(x, w):    dw = G transposed with A = dy [n,ha,wa,ca=Cout], B = x  [n,hb,wb,cb=Cin]
 * (transpose = 1 writes dw[t][cb][ca]).  dw DEV f32 [kh,kw,Cin,Cout] is ADDED to.  a, b DEV dtype
 * (0 f32, 1 bf16); channel counts <= 4 or multiples of 32 up to 256.  The INPUT gradient needs no
 * entry point of its own: dx of tfc_conv2d_down is tfc_conv2d_up(dy, w with its channel axes
 * swapped) cropped to the input size, and dx of tfc_conv2d_up is tfc_conv2d_down(dy, same). */
int tfc_conv2d_wgrad(const void* a, const void* b, float* dw, int dtype, int64_t n, int64_t ha,
                     int64_t wa, int64_t ca, int64_t hb, int64_t wb, int64_t cb, int kh, int kw,
                     int stride, int transpose, void* stream);

/* ------------------------------------------------------------------------ */
/* Training-time entropy bottleneck (deep factorized prior), fused          */
/* ------------------------------------------------------------------------ */

/* ContinuousBatchedEntropyModel.__call__(training=True) with a NoisyDeepFactorized prior and
 * expected_grads=False — python/entropy_models/continuous_batched.py:291-322,
 * python/ops/math_ops.py:157-216, python/distributions/uniform_noise.py:117-156,
 * python/distributions/deep_factorized.py:166-194:
 *   y_hat = y + noise;  log p = log(c(y_hat + .5) - c(y_hat - .5)),  c = sigmoid(logits(.));
 *   bits[u] = -sum over unit u of log p / ln 2.
 * y, noise (or NULL: y_hat = y), y_hat DEV [units, elems] dtype (0 f32, 1 bf16), channels
 * innermost (elems % channels == 0, channels <= 512).  The per-channel MLP has `layers` layers
 * 1 -> width -> ... -> 1; params DEV f32 [channels, P] holds the REPARAMETERISED values
 * (softplus(matrix), bias, tanh(factor)) per channel: layer 0 m[W] b[W] a[W]; middle layers
 * m[W][W] (row = output) b[W] a[W]; last layer m[W] b[1].  log_prob DEV f32 [units, elems] or
 * NULL; bits DEV f32 [units].  Built for (layers, width) = (3,3), (4,3), (3,5). */
int tfc_factorized_bits_forward(const void* y, const void* noise, void* y_hat, int dtype,
                                int64_t units, int64_t elems, int64_t channels, const float* params,
                                int layers, int width, float* log_prob, float* bits, void* stream);

/* Gradients of the above: gbits DEV f32 [units] = dL/dbits; dy DEV [units, elems] dtype =
 * dL/dy_hat through the likelihood (the caller adds the straight path); dparams DEV f32
 * [channels, P] is ADDED to (gradients w.r.t. the reparameterised values). */
int tfc_factorized_bits_backward(const void* y_hat, int dtype, int64_t units, int64_t elems,
                                 int64_t channels, const float* params, int layers, int width,
                                 const float* gbits, void* dy, float* dparams, void* stream);

/* The same with expected gradients — python/ops/math_ops.py:157-216, perturb_and_apply(expected_grads=True),
 * as continuous_batched.py:291-322 calls it when the entropy model was built with expected_grads=True:
 * dy = gbits * (log p(y + .5) - log p(y - .5)) / -ln 2 at the UNPERTURBED input y DEV [units, elems] dtype
 * (the expectation of the derivative over the uniform noise); dparams through log p(y_hat) as above. */
int tfc_factorized_bits_backward_expected(const void* y, const void* y_hat, int dtype, int64_t units,
                                          int64_t elems, int64_t channels, const float* params, int layers,
                                          int width, const float* gbits, void* dy, float* dparams, void* stream);

/* Training-time call of the indexed entropy model with a NoisyNormal prior, fused —
 * python/entropy_models/continuous_indexed.py:313-353 (`__call__(training=True)`: perturb_and_apply of
 * `_log_prob`), python/distributions/uniform_noise.py:117-156 over a Normal base:
 *   y_hat = y + noise (noise NULL: y itself);  log p = log(Phi((y_hat + .5) / scale) - Phi((y_hat - .5) / scale));
 *   bits[unit] = -sum log p / ln 2.
 * y, noise, y_hat DEV [units, elems] dtype (0 f32, 1 bf16); scale DEV f32 [units, elems] (> 0; the location is
 * the caller's: it shifts y); bits DEV f32 [units]. */
int tfc_noisy_normal_bits_forward(const void* y, const void* noise, const float* scale, void* y_hat, int dtype,
                                  int64_t units, int64_t elems, float* bits, void* stream);

/* Gradients of the above: dy DEV [units, elems] dtype = dL/dy_hat through the likelihood (the caller adds the
 * straight path), dscale DEV f32 [units, elems] = dL/dscale (overwritten).  y_in non-NULL selects expected
 * gradients (math_ops.py:157-216): dy = gbits * (log p(y_in + .5) - log p(y_in - .5)) / -ln 2 at the
 * unperturbed input. */
int tfc_noisy_normal_bits_backward(const void* y_in, const void* y_hat, const float* scale, int dtype,
                                   int64_t units, int64_t elems, const float* gbits, void* dy, float* dscale,
                                   void* stream);

/* The four calls above with the Laplace-mixture tail the entropy models take as `laplace_tail_mass`
 * (python/entropy_models/continuous_base.py:298-334, `_log_prob`): with m = laplace_tail_mass in (0, 1) and
 * Q = NoisyLaplace(0, 1),
 *   probs = (1 - m) p(y_hat) + m Q(y_hat);   log p := probs < 1e-10 ? log m + log Q(y_hat) : log probs,
 * in the likelihood, in its gradients (only the branch the reference's tf.where selects carries one) and in the
 * finite difference of expected gradients.  Q is evaluated in closed form (csrc/laplace_tail.h): the
 * reference's float32 difference of Laplace cumulatives loses all its digits beyond |y_hat| ~ 15, exactly where
 * this branch matters.  The backward calls take the unperturbed input y (y_in) or NULL like
 * tfc_noisy_normal_bits_backward: non-NULL selects expected gradients.  The Laplace component sits at 0: the
 * NoisyNormal calls are for a prior whose location the caller has subtracted from y. */
int tfc_factorized_bits_forward_tail(const void* y, const void* noise, void* y_hat, int dtype, int64_t units,
                                     int64_t elems, int64_t channels, const float* params, int layers, int width,
                                     float laplace_tail_mass, float* log_prob, float* bits, void* stream);
int tfc_factorized_bits_backward_tail(const void* y, const void* y_hat, int dtype, int64_t units, int64_t elems,
                                      int64_t channels, const float* params, int layers, int width,
                                      float laplace_tail_mass, const float* gbits, void* dy, float* dparams,
                                      void* stream);
int tfc_noisy_normal_bits_forward_tail(const void* y, const void* noise, const float* scale, void* y_hat, int dtype,
                                       int64_t units, int64_t elems, float laplace_tail_mass, float* bits,
                                       void* stream);
int tfc_noisy_normal_bits_backward_tail(const void* y_in, const void* y_hat, const float* scale, int dtype,
                                        int64_t units, int64_t elems, float laplace_tail_mass, const float* gbits,
                                        void* dy, float* dscale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TFC_HIP_H_ */
