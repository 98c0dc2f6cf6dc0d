/*
 * kapre_hip.h -- C ABI of libkapre_hip.so: Kapre's time-frequency hot path as hand-written
 * gfx950 (MI355X / CDNA4) HIP kernels.
 *
 * This is the drop-in boundary.  The reference (keunwoochoi/kapre 0.4.0) has no FFI of its own:
 * each Keras layer's call() hands its tensors to TensorFlow ops.  Every entry point below
 * replaces the group of tf.* calls of ONE reference call() (file:line cited per function), so a
 * Kapre maintainer can bind it from the layer with a ctypes stub.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch / HIP types in signatures
 *    (kpr_stream_t is the raw hipStream_t value, 0 = the default stream).
 *  - Every data pointer is a DEVICE pointer owned by the caller unless the name ends in _host.
 *    The library never allocates caller-visible memory; scratch comes from the caller
 *    (sizes from kpr_*_workspace_bytes).  The library keeps only immutable per-process caches
 *    (twiddle / DFT tables keyed by device and transform size).
 *  - Every call only ENQUEUES work on `stream` and never synchronises (first use of a new
 *    n_fft uploads a table with a blocking copy before enqueueing).
 *  - Return value: 0 = ok, negative = error (KPR_E_*); text via kpr_last_error() (thread local).
 *  - Data formats follow Kapre: waveforms are (batch, time, ch) "channels_last" or
 *    (batch, ch, time) "channels_first"; spectrograms are (batch, frame, freq, ch)
 *    "channels_last" or (batch, ch, frame, freq) "channels_first".
 *  - float32 / complex64 (interleaved re,im) is Kapre's floatx and the tuned path; the *_f64 / *_c128 entry
 *    points serve layers built with dtype='float64'.
 */
#ifndef KAPRE_HIP_H
#define KAPRE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KPR_VERSION 120 /* 0.1.20 (round 6): stand-alone ApplyFilterbank on banks with a band plan = k_fb_pw (banded row sums; a row
                           * with a NaN / Inf bin returns the DENSE product's result); forward transforms with win_length > n_fft
                           * (cropped frames) on every FFT family, float64 included; option "fb_variant"; kpr_mel_workspace_bytes
                           * covers banks of more than 256 filters;
                           * 111 -> + kpr_device_status / KPR_E_DEVICE: a bounded wait that runs out inside a kernel is reported
                           * (round 5); 110 -> kernels k_mel_fused / k_stft2 removed: kpr_set_option("mel_variant", 1) and
                           * ("stft_variant", 2) are rejected (round 5);
                           * 101 -> + kpr_last_launches, banded mel plan in the packed filterbank (round 4);
                           * 100 -> backward entry points, kpr_filterbank_forget, kpr_debug_sclk_mhz (round 3) */

typedef void* kpr_stream_t;

enum { KPR_CHANNELS_FIRST = 0, KPR_CHANNELS_LAST = 1 };

/* what the STFT kernel writes */
enum {
    KPR_OUT_COMPLEX = 0,   /* complex64, STFT.call                      time_frequency.py:146-187 */
    KPR_OUT_MAGNITUDE = 1, /* float32 |X|, STFT.call + Magnitude.call   time_frequency.py:351-359 */
    KPR_OUT_PHASE = 2      /* float32 angle(X), STFT.call + Phase.call  time_frequency.py:394-402 */
};

enum {
    KPR_OK = 0,
    KPR_E_BADARG = -1,      /* invalid argument (null pointer, non-positive size, bad enum) */
    KPR_E_UNSUPPORTED = -2, /* configuration not implemented */
    KPR_E_HIP = -3,         /* a HIP runtime call failed */
    KPR_E_WORKSPACE = -4,   /* workspace missing or too small */
    KPR_E_DEVICE = -5       /* a kernel of an EARLIER call gave up one of its bounded waits (see kpr_device_status) */
};

/* decibel parameters of backend.magnitude_to_decibel (backend.py:126-194); enabled = 0 skips it */
typedef struct {
    int32_t enabled;
    float ref_value;
    float amin;
    float dynamic_range;
} kpr_db_params;

/* geometry of one STFT configuration (STFT.__init__, time_frequency.py:101-144) */
typedef struct {
    int64_t batch;
    int32_t channels;
    int64_t time;        /* samples per channel */
    int32_t n_fft;
    int32_t win_length;  /* frame_length of tf.signal.stft / inverse_stft.  Forward transforms with win_length > n_fft: frames are cut and
                          * counted with win_length, windowed, and cropped to their first n_fft samples (rfft semantics); the inverse
                          * zero-extends its n_fft-point frames to win_length (time_frequency.py:174-182, :307-314)              */
    int32_t hop_length;
    int32_t pad_begin;   /* left zero pad of n_fft - hop samples (time_frequency.py:169-172) */
    int32_t pad_end;     /* tf.signal.stft(pad_end=...)                                      */
    int32_t in_layout;   /* waveform layout  KPR_CHANNELS_*                                  */
    int32_t out_layout;  /* spectrogram layout KPR_CHANNELS_*                                */
} kpr_stft_geom;

int kpr_version(void);
const char* kpr_last_error(void);
/* Kernels the calling thread's most recent forward call (any kpr_*_f32 / _f64 / _c64 / _c128 entry point) launched, in order,
 * with the template arguments that pick the instance, e.g. "k_stats_init + k_mel_pw<1024,w16> + k_db_clamp",
 * "k_istft_pw_il<512,s4>", "k_stft<512,magnitude,cl>" (thread local; diagnostics: bench.py reports it as roofline.kernel so
 * that the label is what the dispatch actually chose, tests/test_fuzz_gate.py asserts which instance a launch size reached). */
const char* kpr_last_launches(void);

/* Device status.  The kernels whose waves hand work to each other through LDS flags (k_mel_ws, k_istft_ws[_mr], k_istft_pw) bound
 * every such wait (~0.2 s): a protocol error must not hang the GPU.  A wait that runs out leaves wrong values in that launch's
 * output AND raises a bit in a word of mapped host memory:
 *   - every later forward call of the process fails with KPR_E_DEVICE (checked on entry, no synchronisation: a volatile host
 *     read) until kpr_device_status() has read the word;
 *   - kpr_device_status(&flags) returns the bits raised so far and clears them: 0 / flags = 0 when healthy, KPR_E_DEVICE otherwise
 *     (bit 0 k_mel_ws, 1 k_istft_ws consumer, 2 k_istft_ws producer, 3 k_istft_pw, 4 stale band plan -- below --, 5 the output
 *     slot ring of k_mel_pw's PAIR form, 31 the self-test).  It does not synchronise: call it after the stream of the launches in question has been waited for.
 *     flags_out may be NULL.
 * Bit 4: k_mel_pw found that the packed filterbank at fb_packed no longer carries the band plan the library had cached for that
 * address (another blob was written there without kpr_filterbank_forget): that launch computed nothing; reading the status also
 * drops the cache, so the call can simply be repeated.
 * kpr_debug_spin_timeout launches a one-wave kernel whose wait cannot end (limit 64 polls): the reporting chain under test. */
int kpr_device_status(unsigned* flags_out);
int kpr_debug_spin_timeout(kpr_stream_t stream);

/* Process-wide tuning switches (thread safe; take effect for calls issued afterwards).  The library
 * never reads the process environment: what a call does depends on its arguments and these only.
 *   "mel_variant"  0 = automatic (default: the per-wave kernel k_mel_pw for n_fft 256 / 512 / 1024 / 2048 whenever the packed
 *                  filterbank carries a band plan -- mel and other triangular banks; for other matrices (log-frequency
 *                  banks, dense ones) the MFMA kernels: k_mel_ts at n_fft 256 / 512, for interleaved stereo and long runs at
 *                  n_fft 1024, k_mel_ws for the rest of n_fft 1024 / 2048; k_mel_mr for the 18 sizes with a mixed-radix /
 *                  two-pass plan: 96 ... 1000) | 2 = k_mel_ws with the filterbank streamed from L2 per tile (instead of
 *                  register-resident slices) | 3 = k_mel_ws wherever it applies, STFT + filterbank as two launches for
 *                  n_fft 256 and the mixed-radix sizes | 4 = the tile-synchronous kernel k_mel_ts wherever it applies |
 *                  5 / 6 / 7 = k_mel_pw with 8 / 4 / 16 waves per workgroup | 8 = its PAIR form (interleaved waveforms with
 *                  an even channel count at n_fft 1024 / 2048: two channel-frames per fetch; automatic for launches that
 *                  fill the chip) wherever it applies (A/B runs, tests).  1 (the round-1 ring kernel, removed) is rejected
 *   "stft_variant" 0 = automatic (default, channels_first complex / magnitude output and the interleaved-pair layouts:
 *                  k_stft3 -- sixteen-wave workgroups drawing frame groups from an LDS counter -- from 8 groups per CU up,
 *                  k_stft below) | 1 = k_stft | 3 = k_stft3 wherever it applies.  2 (k_stft2, removed) is rejected
 *   "istft_path"   0 = automatic (default: k_istft_pw -- overlap-add in registers, sixteen complete waves per CU -- for
 *                  n_fft 512 / 1024 / 2048 with hop = n_fft / 8, / 4 or / 2 and launches that fill the chip (channels_last with a
 *                  power-of-two channel count included, hop = n_fft / 4 or / 2); else the ring
 *                  kernel, the barrier kernel for launches of up to 3072 frames) | 1 = no wave-specialised ring kernel |
 *                  2 = irFFT + overlap-add as two kernels | 3 = the ring kernel whenever its preconditions hold |
 *                  4 = k_istft_pw whenever its preconditions hold.  Paths 1, 2, 3 produce bit-identical waveforms (same
 *                  frames, ascending order); k_istft_pw sums the R - 1 hop blocks at each boundary between two of its
 *                  frame runs as (earlier frames) + (later frames): deterministic, at most two roundings away (tests)
 *   "mixed_radix"  1 = mixed-radix FFTs for n_fft = 2^a 3^b 5^c plans (default) | 0 = Bluestein instead
 *   "db_chunks"    0 = automatic (default) | n = blocks per batch item of the decibel passes
 *   "db_slots"     0 = automatic (default) | n = statistics slots per batch item in the fused decibel kernels (rounded
 *                  down to a power of two, at most 32; 1 = the single slot of rounds 1-2): small batches spread the
 *                  workgroups' closing max / min atomics over several words per item (bit-identical results)
 *   "mel_cl_stage" 1 = (default) the PAIR form of k_mel_pw with a channels_last output of four or more channels collects the
 *                  n_filt x C block of every (item, frame) in LDS and writes it as one contiguous run | 0 = 8-byte (c, c + 1) stores
 *                  per filter (A/B runs, tests; bit-identical results)
 *   "fb_variant"   0 = automatic (default: stand-alone ApplyFilterbank with a band plan in the packed filterbank -- mel / triangular
 *                  banks, n_freq - 1 a multiple of four up to 1024 -- runs k_fb_pw, banded row sums, on contiguous rows and on rows of
 *                  two interleaved channels) | 1 = the MFMA kernels of rounds 2-5 for every bank (A/B runs, tests)
 *   "verbose"      1 = print launch plans to stderr
 * Unknown name or out-of-range value: KPR_E_BADARG. */
int kpr_set_option(const char* name, int value);
int kpr_get_option(const char* name, int* value);

/* Diagnostics, not needed by a binder.  kpr_debug_stamps: device buffer of 12*32 + 1 int64 that the waves
 * of one workgroup fill with s_memtime stamps (NULL = off, the default); kpr_debug_calib_read8: a kernel
 * with exactly known HBM traffic (reads n_float2 * 8 bytes) for calibrating the rocprofv3 counters. */
int kpr_debug_stamps(void* dev_buf);
int kpr_debug_calib_read8(const void* x, int64_t n_float2, float* out, kpr_stream_t stream);
/* shader clock in MHz measured under a dense packed-f32 vector load (s_memtime ticks per 100 MHz s_memrealtime tick,
 * mean over all waves); blocking; bench.py prints it as sclk_mhz */
int kpr_debug_sclk_mhz(float* out_mhz_host);

/* 1 when n_fft (256, 512, 1024, 2048) runs directly on the LDS Stockham FFT kernels.  n_fft =
 * 2^a 5^b in {160, 200, 320, 400, 640, 800, 1000} and the sizes with a factor 3 in {96, 120, 192, 240,
 * 360, 384, 480, 600, 720, 768, 960} run mixed-radix FFTs, the other even sizes up to 1024 Bluestein's
 * algorithm on top of the Stockham FFT, 4096 and 8192 two / four 1024-point sub-FFTs per frame (forward),
 * the rest a DFT-as-GEMM path. */
int kpr_fft_fast_path(int n_fft);

/* Which forward / inverse FFT family a transform size runs (the float32 STFT / InverseSTFT entry points; fused mel kernels
 * exist for 256 ... 2048 and for the 18 mixed-radix / two-pass sizes).  <0 on bad args. */
enum {
    KPR_FFT_DFT_GEMM = 0,    /* a prime factor above 64: DFT as a GEMM, O(n_fft^2) per frame                     */
    KPR_FFT_POW2 = 1,        /* 256, 512, 1024, 2048: Stockham FFT, 16 points per lane                            */
    KPR_FFT_MIXED_RADIX = 2, /* 2^a 5^b: 160, 200, 320, 400, 640, 800, 1000                                       */
    KPR_FFT_TWO_PASS = 3,    /* sizes with a factor 3: 96, 120, 192, 240, 360, 384, 480, 600, 720, 768, 960       */
    KPR_FFT_BLUESTEIN = 4,   /* the other even sizes up to 1024: chirp-z on the Stockham FFT                      */
    KPR_FFT_SUB_FFT = 5,     /* 4096, 8192: two / four 1024-point sub-FFTs per frame                              */
    KPR_FFT_GENERIC = 6      /* everything else whose prime factors are <= 64: run-time mixed radix in LDS        */
};
/* (win_length > n_fft is classified as win_length = n_fft since KPR_VERSION 120: the forward transform crops its frames) */
int kpr_fft_plan(int n_fft, int win_length);

/* number of frames tf.signal.stft produces for this geometry (after the optional pad_begin);
 * mirrors the reference tests' helpers tests/test_time_frequency.py:32-39.  <0 on bad args. */
int64_t kpr_num_frames(const kpr_stft_geom* g);

/* ---------------------------------------------------------------------------------------------
 * STFT  (replaces tf.transpose + tf.pad + tf.signal.stft + tf.transpose, time_frequency.py:164-185;
 * with mode != KPR_OUT_COMPLEX also the tf.abs / tf.math.angle of the following layer).
 *   x       : float32 waveform in g->in_layout
 *   window  : float32[win_length] analysis window (backend.get_window_fn(name)(win_length)); win_length > n_fft: all win_length
 *             values are passed, the first n_fft are used
 *   out     : complex64 / float32 spectrogram in g->out_layout, n_frames x (n_fft/2+1) per channel
 *   workspace: size from kpr_stft_workspace_bytes (0 for every size the FFT kernels cover)
 */
int64_t kpr_stft_workspace_bytes(const kpr_stft_geom* g, int mode);
int kpr_stft_f32(const float* x, const kpr_stft_geom* g, const float* window, void* out, int mode,
                 void* workspace, int64_t workspace_bytes, kpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused melspectrogram: STFT -> Magnitude -> ApplyFilterbank [-> MagnitudeToDecibel]
 * (the Sequential built by composed.get_melspectrogram_layer, composed.py:138-261; also serves
 * get_log_frequency_spectrogram_layer, composed.py:264-385, with a log filterbank).
 *   fb      : float32 filterbank, row-major (n_freq = n_fft/2+1, n_filt) exactly as
 *             backend.filterbank_mel returns it (backend.py:231)
 *   fb_packed : DEVICE copy of the blob kpr_filterbank_pack builds on the host (same fb, same kranges),
 *             uploaded by the caller; the fused single-kernel path needs it.  The blob starts with a header
 *             naming the matrix it was packed from; on the first call with a given (pointer, n_freq, n_filt,
 *             kranges) the header is read back (one 32-byte copy on the call's stream, waited for -- not under stream
 *             capture) and a blob packed for another matrix shape or other kranges is refused with KPR_E_BADARG.
 *             NULL = two-kernel path (STFT, then |.| x fb GEMM), which needs the larger workspace
 *             kpr_mel_workspace_bytes_unpacked.
 *   fb_kranges_host : optional HOST int32[2*ceil(n_filt/16)] from kpr_filterbank_kranges
 *             (rows outside [lo,hi) of a 16-filter tile are exactly zero and are skipped);
 *             NULL = treat the matrix as dense
 *   out     : float32 (batch, frame, n_filt, ch) or (batch, ch, frame, n_filt)
 *   workspace: kpr_mel_workspace_bytes (dB item statistics; two-kernel path scratch for every n_fft without a fused
 *             kernel and for more than 1024 filters, where fb_packed is ignored); with fb_packed == NULL
 *             kpr_mel_workspace_bytes_unpacked
 *   errors  : malformed fb_kranges_host -> KPR_E_BADARG (only a filterbank too wide for the packed schedule falls
 *             back to the dense product silently)
 */
int64_t kpr_mel_workspace_bytes(const kpr_stft_geom* g, int n_filt, const kpr_db_params* db);
int64_t kpr_mel_workspace_bytes_unpacked(const kpr_stft_geom* g, int n_filt);
int kpr_mel_f32(const float* x, const kpr_stft_geom* g, const float* window, const float* fb,
                const float* fb_packed, int n_filt, const int32_t* fb_kranges_host,
                const kpr_db_params* db, float* out, void* workspace, int64_t workspace_bytes,
                kpr_stream_t stream);

/* Packed (MFMA-fragment order) copy of a filterbank for kpr_mel_f32: size in floats, and the
 * HOST-side packer.  Layout: 64 header words (uint32: 'KPFB' magic, n_freq, n_filt, tiles, chunks, hash of
 * the kranges, zeros), then for 16-filter tile t, chunk c (32 rows), half g, lane l,
 * s = 0..3:
 * out[64 + ((chunk0(t)+c)*2+g)*256 + l*4 + s] = fb[lo(t) + 32c + 16g + 4s + (l>>4)][16t + (l&15)]
 * (0 outside the matrix), lo(t) (rounded down to a multiple of 8) / chunk counts derived from fb_kranges_host
 * (NULL = dense).
 * At most 1024 filters and 32000 rows (KPR_E_UNSUPPORTED beyond: callers then pass fb_packed = NULL and the
 * product runs as a dense GEMM). */
int64_t kpr_filterbank_pack_floats(int n_freq, int n_filt, const int32_t* fb_kranges_host);
int kpr_filterbank_pack(const float* fb_host, int n_freq, int n_filt,
                        const int32_t* fb_kranges_host, float* out_host);

/* The header check of a packed blob is cached per (device address, matrix shape, kranges).  A caller that frees a blob
 * tells the library so: a different buffer that later lands on the same address is then read and checked again instead of
 * being taken for the old one.  fb_packed = NULL forgets every blob.  Host-side bookkeeping only, never fails. */
int kpr_filterbank_forget(const float* fb_packed);

/* Scan a HOST copy of a (n_freq, n_filt) filterbank and write, per tile of 16 filters, the
 * half-open row range [lo, hi) (lo rounded down, hi rounded up to multiples of 4) outside of
 * which every entry of the tile is exactly 0.0f.  out_host has 2*ceil(n_filt/16) entries. */
int kpr_filterbank_kranges(const float* fb_host, int n_freq, int n_filt, int32_t* out_host);

/* ---------------------------------------------------------------------------------------------
 * Stand-alone layers (used when the user composes layers by hand instead of the fused helper)
 */

/* Magnitude.call (tf.abs, time_frequency.py:359) / Phase.call (tf.math.angle, :402) on n complex */
int kpr_abs_c64(const void* x, int64_t n, float* out, kpr_stream_t stream);
int kpr_angle_c64(const void* x, int64_t n, float* out, kpr_stream_t stream);

/* ApplyFilterbank.call (tf.tensordot + tf.transpose, time_frequency.py:535-548).
 *   x  : float32 (batch, frame, n_freq, ch) [layout = LAST] or (batch, ch, frame, n_freq) [FIRST]
 *   out: same layout with n_freq replaced by n_filt */
int kpr_apply_filterbank_f32(const float* x, int64_t batch, int channels, int64_t frames,
                             int n_freq, int layout, const float* fb, int n_filt,
                             const int32_t* fb_kranges_host, float* out, kpr_stream_t stream);

/* Same operation, fast path: with fb_packed (kpr_filterbank_pack of the same fb / kranges, on the
 * DEVICE)
 *   - banks with a band plan (at most two non-zeros per bin, in neighbouring filters: mel / triangular banks, n_freq - 1 a
 *     multiple of four up to 1024) on contiguous rows (channels_first, or one channel) and on rows of two interleaved channels
 *     (channels_last stereo): k_fb_pw, banded row sums straight from global memory (round 6).  A row that contains a NaN / Inf bin returns what the reference's dense tensordot returns -- every filter NaN or
 *     +-Inf -- (the row is recomputed against fb, which therefore must be the matrix fb_packed was packed from);
 *   - other wide banded matrices (log-frequency banks, interleaved rows): the fused kernel's MFMA consumers fed by loader waves;
 *     a non-finite bin poisons the 16-filter tiles whose row range contains it, not the other filters of the row;
 *   - everything else falls through to kpr_apply_filterbank_f32 (dense: a non-finite bin poisons the whole row, unless
 *     fb_kranges_host lets the GEMM skip the tiles that do not contain it).
 * fb_packed may be NULL. */
int kpr_apply_filterbank_packed_f32(const float* x, int64_t batch, int channels, int64_t frames,
                                    int n_freq, int layout, const float* fb, const float* fb_packed,
                                    int n_filt, const int32_t* fb_kranges_host, float* out,
                                    kpr_stream_t stream);

/* MagnitudeToDecibel.call -> backend.magnitude_to_decibel (backend.py:126-194).
 * x is viewed as n_items rows of item_size elements (Kapre: item = one batch element, all other
 * axes flattened; a rank-1 input is ONE item).  In-place (out == x) is allowed.
 * workspace: kpr_db_workspace_bytes(n_items). */
int64_t kpr_db_workspace_bytes(int64_t n_items);
int kpr_mag_to_db_f32(const float* x, int64_t n_items, int64_t item_size, const kpr_db_params* db,
                      float* out, void* workspace, int64_t workspace_bytes, kpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * InverseSTFT.call (tf.transpose + tf.signal.inverse_stft + tf.transpose, time_frequency.py:304-317)
 *   spec         : complex64 spectrogram, g->out_layout, n_frames x (n_fft/2+1) per channel
 *   synth_window : float32[win_length] = tf.signal.inverse_stft_window_fn(hop, fwd)(win_length)
 *   out          : float32 waveform in g->in_layout, length (n_frames-1)*hop + win_length
 * g->time, pad_begin and pad_end are ignored; n_frames is given explicitly.
 */
int64_t kpr_istft_workspace_bytes(const kpr_stft_geom* g, int64_t n_frames);
int kpr_istft_f32(const void* spec, const kpr_stft_geom* g, int64_t n_frames,
                  const float* synth_window, float* out, void* workspace, int64_t workspace_bytes,
                  kpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * float64 / complex128 variants.  Kapre computes in the dtype of the Keras layer ("complex64 if x is float32,
 * complex128 if x is float64", time_frequency.py:155); a layer built with dtype='float64' runs these.  Same
 * arguments and layouts as the float32 entry points above with double / complex128 (interleaved re,im) data;
 * any n_fft whose frame fits in LDS (even n_fft <= 10240, odd <= 5120), any win_length.  Plain size-generic kernels: float64
 * is not the hot path.
 */
int kpr_stft_f64(const double* x, const kpr_stft_geom* g, const double* window, void* out, int mode,
                 kpr_stream_t stream);
int64_t kpr_istft_f64_workspace_bytes(const kpr_stft_geom* g, int64_t n_frames);
int kpr_istft_f64(const void* spec, const kpr_stft_geom* g, int64_t n_frames, const double* synth_window,
                  double* out, void* workspace, int64_t workspace_bytes, kpr_stream_t stream);
int kpr_abs_c128(const void* x, int64_t n, double* out, kpr_stream_t stream);
int kpr_angle_c128(const void* x, int64_t n, double* out, kpr_stream_t stream);
int kpr_apply_filterbank_f64(const double* x, int64_t batch, int channels, int64_t frames, int n_freq, int layout,
                             const double* fb, int n_filt, double* out, kpr_stream_t stream);
/* backend.magnitude_to_decibel (backend.py:126-194) in float64; in-place (out == x) is allowed */
int kpr_mag_to_db_f64(const double* x, int64_t n_items, int64_t item_size, double ref_value, double amin,
                      double dynamic_range, double* out, kpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Backward passes (vector-Jacobian products).  The reference's layers are TensorFlow graphs and therefore
 * differentiable: a Kapre front end sits inside model.fit / tf.GradientTape (time_frequency.py:146-187, :289-319,
 * :351-359, :402, :535-548; backend.py:186-192).  Complex cotangents follow TensorFlow's convention
 * (dL/dRe + i dL/dIm).  The LINEAR layers need no kernels of their own -- their adjoints are forward launches:
 *   STFT^T          (spectrogram cotangent G -> waveform cotangent):  kpr_spec_edge_scale_c64(G, s_edge = 1,
 *                   s_mid = 0.5), then kpr_istft_f32 with synth_window = n_fft * analysis window; sample t of the
 *                   waveform is sample t + pad_left of that output (zero where no frame reaches);
 *   InverseSTFT^T   (waveform cotangent -> spectrogram cotangent):  kpr_stft_f32 (no padding) with window =
 *                   2 / n_fft * synthesis window, then kpr_spec_edge_scale_c64(s_edge = 0.5, s_mid = 1) in place;
 *   ApplyFilterbank^T:  kpr_apply_filterbank_f32 with the transposed matrix (n_freq <-> n_filt).
 * kapre_amd/autograd.py composes exactly these calls.  The entry points below cover the non-linear layers.
 */

/* Magnitude (tf.abs on complex: g * x / |x|, 0 at x == 0) and Phase (tf.math.angle: g * (-im + i re) / |x|^2).
 * x, gx: n complex (interleaved re, im); g: n real */
int kpr_abs_c64_bwd(const void* x, const float* g, int64_t n, void* gx, kpr_stream_t stream);
int kpr_angle_c64_bwd(const void* x, const float* g, int64_t n, void* gx, kpr_stream_t stream);
int kpr_abs_c128_bwd(const void* x, const double* g, int64_t n, void* gx, kpr_stream_t stream);
int kpr_angle_c128_bwd(const void* x, const double* g, int64_t n, void* gx, kpr_stream_t stream);

/* out = in * (bin is DC, or Nyquist of an even n_fft ? s_edge : s_mid) on a complex spectrogram whose frequency axis
 * has n_freq = n_fft / 2 + 1 bins and `inner` elements per bin step (channels for channels_last data, 1 for
 * channels_first); n = complex element count.  In place (out == in) is allowed. */
int kpr_spec_edge_scale_c64(const void* in, int64_t n, int n_freq, int inner, int n_fft, float s_edge, float s_mid,
                            void* out, kpr_stream_t stream);
int kpr_spec_edge_scale_c128(const void* in, int64_t n, int n_freq, int inner, int n_fft, double s_edge, double s_mid,
                             void* out, kpr_stream_t stream);

/* backend.magnitude_to_decibel (backend.py:186-192) backward, from the layer's INPUT x and the output cotangent gy:
 * gx = 10 / (ln 10 * x) * [x >= amin] * (gy [l >= max_l - dynamic_range] + [l == max_l] / ties * sum of the gy
 * below the floor), l = the decibel value before the floor -- the floor moves with the item's maximum, as in
 * TensorFlow's gradient of tf.maximum(l, reduce_max(l) - dynamic_range).  Items as in kpr_mag_to_db_f32. */
int kpr_mag_to_db_bwd_f32(const float* x, const float* gy, int64_t n_items, int64_t item_size,
                          const kpr_db_params* db, float* gx, kpr_stream_t stream);
int kpr_mag_to_db_bwd_f64(const double* x, const double* gy, int64_t n_items, int64_t item_size, double ref_value,
                          double amin, double dynamic_range, double* gx, kpr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Consumers / neighbours of the path (kapre/signal.py, time_frequency.py:563-644)
 */

/* frames tf.signal.frame produces: 1 + (time - frame_length) / hop (0 when time < frame_length),
 * ceil(time / hop) with pad_end.  <0 on bad args. */
int64_t kpr_frame_count(int64_t time, int frame_length, int hop_length, int pad_end);

/* Frame.call (tf.signal.frame, signal.py:86-104).
 *   x  : float32 (batch, time, ch) [layout = LAST] or (batch, ch, time) [FIRST]
 *   out: (batch, n_frames, frame_length, ch) [LAST] or (batch, ch, n_frames, frame_length) [FIRST];
 *        samples beyond the end of the signal (pad_end) are pad_value */
int kpr_frame_f32(const float* x, int64_t batch, int channels, int64_t time, int layout,
                  int frame_length, int hop_length, int pad_end, float pad_value, float* out,
                  kpr_stream_t stream);

/* Energy.call (signal.py:187-213): scale * sum over the frame of x^2, frames as above,
 * scale = ref_duration / (frame_length / sample_rate).
 *   out: (batch, n_frames, ch) [LAST] or (batch, ch, n_frames) [FIRST] */
int kpr_energy_f32(const float* x, int64_t batch, int channels, int64_t time, int layout,
                   int frame_length, int hop_length, int pad_end, float pad_value, float scale,
                   float* out, kpr_stream_t stream);

/* tf.pad modes Delta accepts (time_frequency.py:596-600) */
enum { KPR_PAD_CONSTANT = 0, KPR_PAD_SYMMETRIC = 1, KPR_PAD_REFLECT = 2 };

/* Delta.call (tf.pad + K.conv2d with the kernel [-n .. n] / (2 sum i^2), time_frequency.py:614-635).
 *   x, out: float32 (batch, time, freq, ch) [LAST] or (batch, ch, time, freq) [FIRST]
 *   win_length: odd, >= 3.  In-place (out == x) is NOT allowed. */
int kpr_delta_f32(const float* x, int64_t batch, int channels, int64_t frames, int n_freq, int layout,
                  int win_length, int pad_mode, float* out, kpr_stream_t stream);

/* Backward passes of the three layers above (cotangent of the output -> cotangent of the input; shapes and layouts
 * as in the forward entry points).  Gather form, deterministic.
 *   Frame^T : gx[t] = sum of g[f][t - f hop] over the frames that cover sample t (padding receives nothing)
 *   Energy^T: gx[t] = 2 scale x[t] * sum of g[f] over the frames that cover t
 *   Delta^T : the transposed correlation including the mirror images the padding mode folds back */
int kpr_frame_bwd_f32(const float* g, int64_t batch, int channels, int64_t time, int layout, int frame_length,
                      int hop_length, int pad_end, float* gx, kpr_stream_t stream);
int kpr_energy_bwd_f32(const float* x, const float* g, int64_t batch, int channels, int64_t time, int layout,
                       int frame_length, int hop_length, int pad_end, float scale, float* gx, kpr_stream_t stream);
int kpr_delta_bwd_f32(const float* g, int64_t batch, int channels, int64_t frames, int n_freq, int layout,
                      int win_length, int pad_mode, float* gx, kpr_stream_t stream);

/* LogmelToMFCC.call (tf.signal.mfccs_from_log_mel_spectrograms, signal.py:418-436) has no entry
 * point of its own: it is kpr_apply_filterbank_f32 with the (n_mels, n_mfccs) DCT-II matrix
 * M[n][k] = 2 cos(pi (2n+1) k / (2 n_mels)) / sqrt(2 n_mels) and fb_kranges_host = NULL. */

#ifdef __cplusplus
}
#endif
#endif /* KAPRE_HIP_H */
