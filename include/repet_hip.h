/*
 * repet_hip.h -- C ABI of librepet_hip.so, the MI355X (gfx950) REPET separation engine.
 *
 * The reference (zafarrafii/REPET-Python, repet.py) has no FFI seam: its boundary is the Python call
 *   repet.<original|extended|adaptive|sim|simonline>(audio_signal[N,C], sampling_frequency)
 *     -> background_signal[N,C] float64                      (repet.py:67,205,422,571,712)
 * This header is what a ctypes binding of that call binds instead. Plain C types
 * only; the caller owns every buffer; the library keeps no caller pointer after a call returns.
 *
 * Every derived size (window length, period range in frames, cutoff bin, buffer length ...) crosses
 * the ABI as an INTEGER already evaluated by the caller with the reference's own expressions
 * (Python round()/np.round are half-to-even, C round() is not) -- the C side never converts
 * seconds to frames.
 *
 * Return value: 0 on success, negative repet_status otherwise; text via repet_last_error().
 * Thread-safety: a repet_ctx serialises its own calls on one HIP stream; different contexts may be
 * used from different host threads concurrently.
 */
#ifndef REPET_HIP_H
#define REPET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REPET_ABI_VERSION 4

typedef enum repet_status {
    REPET_OK = 0,
    REPET_ERR_BAD_ARG = -1,   /* malformed argument: the Python shim raises ValueError            */
    REPET_ERR_TOO_SHORT = -2, /* clip too short for the algorithm (repet.py:1263, :802): ValueError */
    REPET_ERR_HIP = -3,       /* HIP runtime error: RuntimeError                                   */
    REPET_ERR_OOM = -4,       /* device allocation failed                                          */
    REPET_ERR_LIMIT = -5      /* size outside what the kernels support (documented in DESIGN.md)   */
} repet_status;

/* Which public function of the reference to run. */
typedef enum repet_algo {
    REPET_ORIGINAL = 0,  /* repet.py:67-202  */
    REPET_EXTENDED = 1,  /* repet.py:205-419 */
    REPET_ADAPTIVE = 2,  /* repet.py:422-568 */
    REPET_SIM = 3,       /* repet.py:571-709 */
    REPET_SIMONLINE = 4  /* repet.py:712-911 */
} repet_algo;

typedef enum repet_dtype { REPET_F32 = 0, REPET_F64 = 1, REPET_I16 = 2 } repet_dtype;

/* The reference's nine module globals (repet.py:42-63), already converted to frames/bins/samples. */
typedef struct repet_params {
    int32_t window_length;       /* W = 2^ceil(log2(0.04 fs))                       repet.py:130 */
    int32_t step_length;         /* H = W/2                                         repet.py:132 */
    int32_t period_lo;           /* np.round(period_range[0]*fs/H)                  repet.py:165 */
    int32_t period_hi;           /* np.round(period_range[1]*fs/H)                  repet.py:165 */
    int32_t cutoff_bins;         /* round(cutoff_frequency*W/fs)                    repet.py:173 */
    int32_t filter_order;        /* adaptive median order                           repet.py:54  */
    int32_t seg_len_frames;      /* int(round(segment_length*fs/H))   (adaptive)    repet.py:519 */
    int32_t seg_step_frames;     /* int(round(segment_step*fs/H))     (adaptive)    repet.py:520 */
    int32_t sim_distance_frames; /* int(round(similarity_distance*fs/H))            repet.py:670 */
    int32_t sim_number;          /* similarity_number                               repet.py:60  */
    int32_t buffer_frames;       /* round(buffer_length*fs/H)         (simonline)   repet.py:787 */
    int32_t flags;               /* REPET_FLAG_* (0: the defaults); was reserved0 before ABI 3 */
    int64_t seg_len_samples;     /* round(segment_length*fs)          (extended)    repet.py:266 */
    int64_t seg_step_samples;    /* round(segment_step*fs)            (extended)    repet.py:267 */
    double sim_threshold;        /* similarity_threshold                            repet.py:58  */
} repet_params;

/* repet_params.flags -- samples that are not finite. repet.py never looks at its input (:125, :1220): a NaN sample makes the
 * frames that hold it NaN; `sim` / `simonline` confine the damage to those frames; the period family turns the beat spectrum
 * of the clip / segment / windows NaN (period = period_range[0] + 1 there) and, through np.median's NaN rule, marks the same
 * position of EVERY period. Since ABI 4 that is the DEFAULT here too (flags = 0, a fresh context): repet_run / repet_run_batch
 * (both transports) and the context calls let the samples through and all five variants return what the reference returns for
 * NaN samples: NaN positions equal, every other sample within the usual bar, periods and similar-frame lists equal (tested
 * against the oracle, which equals the reference bit for bit on such input). An INFINITE sample is treated as a NaN sample:
 * for `sim` / `simonline` that is the reference's result; for the period family the reference's own result depends on where
 * pocketfft's butterflies meet inf - inf, and the engine's NaN samples are a superset of the reference's.
 * REPET_FLAG_REFUSE_NONFINITE (bit 1) is the option: host arrays with such samples are refused (REPET_ERR_BAD_ARG, "contains
 * NaN or infinite samples" -- the default before ABI 4). Bit 0 (REPET_FLAG_STRICT_REFERENCE, the ABI-3 opt-in) is still
 * accepted and means the default. Any other bit is an argument error. Context calls: repet_ctx_set_strict_reference(ctx, 0)
 * turns the refusal on for that context's uploads. */
#define REPET_FLAG_STRICT_REFERENCE 1
#define REPET_FLAG_REFUSE_NONFINITE 2

/* The nine module-level parameters of the reference (repet.py:42-63), for hosts that do not keep them as Python
 * globals, and the derivation of repet_params from them exactly as the reference's public functions do it
 * (Python round / np.round are half-to-even; so is this). repet_default_settings writes the reference's defaults:
 * 100 Hz, [1, 10] s, 10 s / 5 s segments, order 5, threshold 0, distance 1 s, 100 frames, 10-s buffer. */
typedef struct repet_settings {
    double cutoff_frequency;      /* Hz  */
    double period_range[2];       /* s   */
    double segment_length;        /* s   */
    double segment_step;          /* s   */
    double similarity_threshold;  /* [0, 1] */
    double similarity_distance;   /* s   */
    double buffer_length;         /* s   */
    int32_t filter_order;
    int32_t similarity_number;
} repet_settings;
void repet_default_settings(repet_settings* s);
int repet_derive_params(const repet_settings* s /* NULL: the defaults */, double sampling_frequency, repet_params* out);

/* Per-stage device time of the last repet_ctx_execute, from HIP events on the context's stream. */
#define REPET_MAX_STAGES 16
typedef struct repet_timing {
    int32_t n_stages;
    int32_t reserved0;
    float total_ms;                       /* first event to last event                     */
    float stage_ms[REPET_MAX_STAGES];     /* one entry per stage, in launch order          */
    char stage_name[REPET_MAX_STAGES][24];
    double stage_bytes[REPET_MAX_STAGES]; /* algorithmic HBM bytes of the stage */
    double stage_flops[REPET_MAX_STAGES]; /* algorithmic flops of the stage                 */
} repet_timing;

typedef struct repet_ctx repet_ctx;

/* ---- library --------------------------------------------------------------------------------- */
int repet_abi_version(void);
int repet_device_count(void);
/* (ABI 3) The CPUs of the NUMA node `device` hangs off, among those this process may run on (from the device's PCI address in
 * sysfs; *n_cpus = 0: unknown, or the machine has one node). On a two-socket host the drop-in call -- whose cost is host
 * threads narrowing the caller's array into pinned memory and two PCIe copies -- takes 3.9 ms when the calling process runs
 * on the GPU's node and 4.3 .. 5.9 ms elsewhere (180-s stereo clip): a host binds itself with
 * sched_setaffinity to these CPUs (repet.bind_host_to_device). The library pins only its OWN worker threads
 * (REPET_HOST_NUMA=0: not even those). */
int repet_device_host_cpus(int device, int32_t* cpus, int32_t capacity, int32_t* n_cpus);
const char* repet_last_error(void); /* thread-local text of the last failure on this thread */

/* ---- context: one per (host thread, device); owns a stream, workspaces, twiddle tables ------- */
int repet_ctx_create(int device, repet_ctx** out);
int repet_ctx_destroy(repet_ctx* ctx);

/* Device-resident path (what bench.py times): upload once, execute any number of times, download.
 * upload  : audio[n_samples*n_channels] in NumPy C order (audio[n*C + c]), converted to fp32 on device
 * execute : runs the whole algorithm on the resident clip; blocks until the stream is idle
 * download: background_signal as float64 [n_samples][n_channels]                                  */
int repet_ctx_upload(repet_ctx* ctx, const void* audio, int dtype, int64_t n_samples, int32_t n_channels);
/* n_clips equal-shape clips back to back, audio[n_clips][n_samples][n_channels] (a Python loop over
 * repet.simonline(clip, fs), BASELINE.json configs[4]). execute then separates every clip: simonline runs each
 * stage ONCE over all clips (one launch per stage instead of one per clip and stage), the other variants work
 * through the resident clips one after the other. download / download_foreground return
 * [n_clips][n_samples][n_channels]; the integer intermediates are those of the last clip. */
int repet_ctx_upload_batch(repet_ctx* ctx, const void* audio, int dtype, int64_t n_samples, int32_t n_channels,
                           int32_t n_clips);
int repet_ctx_execute(repet_ctx* ctx, int algo, const repet_params* p, repet_timing* timing /* nullable */);
int repet_ctx_download(repet_ctx* ctx, double* out);

/* Device-resident ingest / egress (multi-GPU: a waveform received over RCCL/xGMI lands in device memory and goes
 * straight into the engine, no host bounce -- SURVEY 8e; the reference has no counterpart, its arrays live in host RAM):
 * fp32 interleaved [n_clips][n_samples][n_channels], device (or peer-accessible) memory, complete on return (the
 * producer of dev_audio must have finished; dev_out may be handed to a collective right away). */
int repet_ctx_upload_device(repet_ctx* ctx, const float* dev_audio, int64_t n_samples, int32_t n_channels, int32_t n_clips);
/* (ABI 2) The same with the fp32 REMAINDERS of a float64 waveform beside its fp32 samples: dev_audio_lo[i] = (float)(x[i] -
 * (double)(float)x[i]) (nullable: none). repet.py computes in float64 throughout (:149, :1223, :1318-1326); the engine takes
 * the few float64 decisions of its peak picking from sample + remainder = 48 bits of the caller's double, and a waveform
 * that reaches a device over RCCL must carry them to be separated exactly as the same waveform passed to repet_run is. */
int repet_ctx_upload_device_split(repet_ctx* ctx, const float* dev_audio, const float* dev_audio_lo, int64_t n_samples,
                                  int32_t n_channels, int32_t n_clips);
int repet_ctx_download_device(repet_ctx* ctx, float* dev_out);
/* (ABI 3) Borrowed views for a host that keeps the exchange on the device (one process per GPU, torch.distributed over
 * RCCL: repet/parallel.py). The pointers stay valid until the next upload of this context (a larger clip may move the
 * buffers); what a run writes into the result is ordered on the context's stream, which repet_ctx_stream hands out so that
 * the host can enqueue its sends, receives and adds BEHIND the run instead of waiting for it on the CPU.
 * repet_ctx_result_view : the fp32 result [n_clips][n_samples][n_channels] of the last run (what repet_ctx_download widens)
 * repet_ctx_input_view  : the resident fp32 samples and, when a float64 upload left any, their fp32 remainders (else NULL)
 * repet_ctx_download_from: n_values fp32 values from ANY device buffer of this context's device widened into a host float64
 *                         array through the context's pinned ring (the gather side of a scatter: results received from peers) */
int repet_ctx_set_strict_reference(repet_ctx* ctx, int on);   /* (ABI 3; on by default since ABI 4) on = 0: REPET_FLAG_REFUSE_NONFINITE for this context's uploads */
int repet_ctx_stream(repet_ctx* ctx, void** hip_stream);
int repet_ctx_result_view(repet_ctx* ctx, float** dev_out, int64_t* n_values /* nullable */);
int repet_ctx_input_view(repet_ctx* ctx, float** dev_audio, float** dev_audio_lo /* nullable */, int64_t* n_values /* nullable */);
int repet_ctx_download_from(repet_ctx* ctx, const float* dev_src, int64_t n_values, double* out);
/* Declare the resident samples to be [sample0, sample0 + n_samples) of a clip of n_total samples, for
 * repet_ctx_execute_extended_range: a rank of a multi-GPU `extended` then holds (and returns) only the samples its own
 * segments cover -- (count + 1) segment steps instead of the whole clip (repet.py:306-414 touches nothing else).
 * Cleared by every upload; (0, 0) clears it. */
int repet_ctx_set_window(repet_ctx* ctx, int64_t n_total, int64_t sample0);

/* ---- WAVE files either side of the path (wavread / wavwrite, repet.py:914-946; README.md:62-72) -------------------
 * repet_wav_parse      : header of a RIFF/WAVE file image (PCM 8/16/24/32-bit, IEEE float 32/64-bit, plain or
 *                        WAVE_FORMAT_EXTENSIBLE); REPET_ERR_BAD_ARG with a message for anything else.
 * repet_ctx_upload_wav : the file's samples become the resident clip: the RAW bytes cross PCIe (2-3 bytes per sample, not
 *                        8), are decoded on the device and normalised exactly as wavread does -- divided by
 *                        2^(8*itemsize-1) of the array scipy.io.wavfile.read returns (repet.py:929; 24-bit PCM counts
 *                        as int32, float files are divided as well).
 * repet_ctx_result_wav : image of the file wavwrite(background | audio - background, fs, file) writes for a float64
 *                        (dtype REPET_F64) or float32 (REPET_F32) array: IEEE-float WAVE, fmt + fact + data chunks.
 *                        which: 1 background, 2 foreground. capacity >= 58 + n_samples*n_channels*itemsize. */
typedef struct repet_wav_info {
    int32_t format;              /* 1 PCM, 3 IEEE float (the sub-format of an extensible header) */
    int32_t n_channels;
    int32_t sampling_frequency;
    int32_t bits_per_sample;
    int32_t bytes_per_sample;    /* container width: block_align / n_channels */
    int32_t reserved0;
    int64_t data_offset;         /* first byte of the samples in the file image */
    int64_t n_samples;           /* per channel */
} repet_wav_info;
int repet_wav_parse(const void* file_bytes, int64_t n_bytes, repet_wav_info* info);
int repet_ctx_upload_wav(repet_ctx* ctx, const void* file_bytes, int64_t n_bytes, repet_wav_info* info_out);
int repet_ctx_result_wav(repet_ctx* ctx, int which, int dtype, void* file_out, int64_t capacity, int64_t* n_written);
/* sampling frequency repet_ctx_result_wav writes into the header when the clip did not come from repet_ctx_upload_wav */
int repet_ctx_set_sampling_frequency(repet_ctx* ctx, int32_t sampling_frequency);

/* Pinned host buffers from a recycling pool, for results: repet_ctx_download / repet_run write into any memory, but a
 * buffer that is already faulted in and pinned takes the copy at link speed (a fresh malloc / np.empty of a 3-minute
 * clip page-faults 31 000 times on first touch). repet_host_free returns the buffer to the pool; NULL when the pool
 * declines (use ordinary memory then). The Python module wraps these as the NumPy arrays it returns
 * (the reference returns a fresh array per call, repet.py:176,302,540,683,829). */
void* repet_host_alloc(size_t bytes);
void repet_host_free(void* ptr);
/* Non-blocking form of execute: enqueues the whole run on the context's stream and returns; contexts have
 * their own streams, so runs of different contexts overlap on the device. Errors detected at enqueue time are
 * returned here, device-side failures by repet_ctx_synchronize (or the next blocking call). */
int repet_ctx_execute_async(repet_ctx* ctx, int algo, const repet_params* p);
int repet_ctx_synchronize(repet_ctx* ctx);
/* Per-stage device times of runs that are enqueued back to back (bench.py's timed region): between begin and end every
 * repet_ctx_execute_async of this context (the first n_steps of them) records its own block of HIP events on the
 * context's stream; end waits for the stream and returns the MEAN stage times over the runs made (names, bytes and
 * flops as repet_ctx_execute's timing gives them). No host synchronisation happens between the runs. */
int repet_ctx_timing_series_begin(repet_ctx* ctx, int32_t n_steps);
int repet_ctx_timing_series_end(repet_ctx* ctx, repet_timing* mean, int32_t* n_steps /* nullable */);

/* The steps either side of the path in every README example (README.md:64-98), kept on the device:
 * foreground_signal = audio_signal - background_signal (README.md:69) of the last run, float64 [n][C];
 * and the magnitude spectrogram abs(_stft(mean over channels of the signal)[0:F]) (README.md:79-81) of the
 * mixture (which = 0), the background (1) or the foreground (2) as spec[T][F] fp32, T = repet_frame_count(n, W, W/2, 1). */
int repet_ctx_download_foreground(repet_ctx* ctx, double* out);
int repet_ctx_spectrogram(repet_ctx* ctx, int which, int32_t window_length, float* spec_out, int64_t n_frames);

/* extended only (repet.py:205-419): the segment plan of an n_samples clip (repet.py:271-281), and a run
 * restricted to segments [first, first+n_segments). Contributions of the other segments stay zero, and
 * the cross-fade is linear in the segments, so the outputs of disjoint ranges (e.g. one range per GPU)
 * add up to the full result. */
int64_t repet_extended_segment_count(int64_t n_samples, const repet_params* p);
int repet_ctx_execute_extended_range(repet_ctx* ctx, const repet_params* p, int64_t first, int64_t n_segments,
                                     repet_timing* timing /* nullable */);
/* (ABI 3) The same enqueued on the context's stream without waiting for it (see repet_ctx_execute_async). */
int repet_ctx_execute_extended_range_async(repet_ctx* ctx, const repet_params* p, int64_t first, int64_t n_segments);

/* ---- one-shot drop-in: replaces repet.<algo>(audio_signal, fs) (repet.py:67,205,422,571,712) -- */
/* Measurement aid (bench.py's roofline): the median of a list of at most list_bound values is a compare-exchange
 * selection network evaluated per lane (np.median, repet.py:1535); *network_size = wires of the compiled network
 * (0: lists longer than 128 use bisection), *instructions = min/max instructions per evaluation. A NEGATIVE list_bound asks
 * about the bit-sliced selection (mask_bits.hip) for lists of at most -list_bound entries: *network_size = 0, *instructions =
 * boolean wave instructions per frame (all its cells) and code plane. */
int repet_median_network_info(int32_t list_bound, int32_t* network_size, int32_t* instructions);

/* repet_run keeps one context per calling thread and device (stream, tables, grow-only workspaces -- for `sim` the
 * T x T similarity matrix) so that repeated calls reuse them; it is destroyed when the thread exits. This releases the
 * calling thread's contexts now (e.g. after a one-off long clip). */
int repet_release_thread_ctx(void);
int repet_run(int algo, const void* audio, int dtype, int64_t n_samples, int32_t n_channels,
              const repet_params* p, double* out, int device, repet_timing* timing /* nullable */);

/* Batch of independent clips dealt round-robin (longest first) over n_devices GPUs of this process: one host thread,
 * context and stream per device; every device uploads its own clips from the caller's arrays and downloads its own
 * results (each over its own PCIe link). REPET_LOGICAL_DEVICES=n (test switch) lets n_devices exceed the visible GPUs:
 * logical device d runs on physical device d % visible. */
int repet_run_batch(int algo, int32_t n_clips, const void* const* audio, int dtype,
                    const int64_t* n_samples, const int32_t* n_channels, const repet_params* p,
                    double* const* out, int32_t n_devices);
/* The same with the waveforms travelling over xGMI: the clips enter through device 0 and are worked through in
 * rounds of one clip per device -- a round goes to its devices as ONE group of ncclSend / ncclRecv (fp32, interleaved; a
 * float64 clip as two planes, samples and remainders; ncclCommInitAll over devices 0 .. n_devices-1, cached; librccl opened
 * on first use), is separated from the received device buffers while the root stages the next round, and its results
 * return in a group of their own.
 * REPET_RCCL_SELF=1 with n_devices == 1 (test switch): every clip is sent by device 0 to itself inside the group.
 * repet_last_batch_info: what the calling thread's last batch call did -- out[0] transport (0 host, 1 RCCL), out[1] clips that
 * went through send / receive, out[2] clips whose remainder plane was resident when they were separated, out[3] RCCL groups
 * completed. */
int repet_last_batch_info(int64_t out[4]);
/* The resident clip as the engine holds it: the fp32 samples and, for a float64 upload, their fp32 remainders
 * (x - (double)(float)x; zeros and *has_remainders = 0 when none was needed). [n_clips][n_samples][n_channels] floats each. */
int repet_ctx_download_input(repet_ctx* ctx, float* samples_out, float* remainders_out, int32_t* has_remainders);
/* (ABI 4) Clips one after another through ONE device with `depth` (1 .. 8) of them in flight: every clip in flight has a
 * context of its own (stream, pinned ring, workspaces), so clip k + 1 is narrowed and uploaded and clip k - 1 copied back and
 * widened while clip k is separated. Results are what repet_run gives for each clip, bit for bit. The contexts stay cached
 * for the next call; repet_release_thread_ctx frees them. (What a root rank does with its own share of a scattered batch,
 * and the one-GPU form of repet.run_batch.) */
int repet_run_stream(int algo, int32_t n_clips, const void* const* audio, int dtype,
                     const int64_t* n_samples, const int32_t* n_channels, const repet_params* p,
                     double* const* out, int device, int32_t depth);
int repet_run_batch_rccl(int algo, int32_t n_clips, const void* const* audio, int dtype,
                         const int64_t* n_samples, const int32_t* n_channels, const repet_params* p,
                         double* const* out, int32_t n_devices);

/* ---- stage-level exports (parity tests; layouts follow the reference helper they replace) ---- */

/* repet.py:1021-1028 (centred=1) / repet.py:781 (centred=0): number of frames for n samples. */
int64_t repet_frame_count(int64_t n_samples, int32_t window_length, int32_t step_length, int32_t centred);

/* _stft, repet.py:1001-1060. x[n] one channel -> spec[T][F] interleaved (re,im) fp32, F = W/2+1
 * (frame-major; the reference's (W,T) array is its transpose plus the mirrored bins). */
int repet_stft(repet_ctx* ctx, const float* x, int64_t n, const float* window, int32_t window_length,
               int32_t step_length, int32_t centred, float* spec_out, int64_t n_frames);

/* _istft, repet.py:1063-1105. spec[T][F] (re,im) -> y[(T-1)*H] (centred: trimmed W-H each end and
 * divided by sum(window[0:W:H])). */
int repet_istft(repet_ctx* ctx, const float* spec, int64_t n_frames, const float* window,
                int32_t window_length, int32_t step_length, float* y_out, int64_t n_out);

/* _selfsimilaritymatrix, repet.py:1209-1225. v[T][F] (frame-major magnitudes) -> s[T][T]. */
int repet_selfsim(repet_ctx* ctx, const float* v, int64_t n_frames, int32_t n_freq, float* s_out);
/* The same with the SEGMENT RECORDS the peak picking of `sim` takes its candidates from (repet.py:1294-1345 scans every element;
 * a strict maximum of a window of +-d >= 31 elements is the maximum of its aligned 32-element segment): for every row and every
 * run of columns [32 u, 32 u + 32) the largest value (NaN counted as +inf), the largest of the run's OTHER elements (-inf for
 * a run of one), and the offset of the largest inside the run (the lowest among equals). max_out / second_out / at_out:
 * [n_frames][ceil(n_frames / 32)]. */
int repet_selfsim_records(repet_ctx* ctx, const float* v, int64_t n_frames, int32_t n_freq, float* s_out, float* max_out,
                          float* second_out, int32_t* at_out);

/* _similaritymatrix, repet.py:1228-1246 (the online variant's frame-vs-buffer similarity):
 * a[TA][F], b[TB][F] frame-major magnitudes -> s[TA][TB] cosine similarities. */
int repet_similarity(repet_ctx* ctx, const float* a, int64_t n_a, const float* b, int64_t n_b, int32_t n_freq,
                     float* s_out);

/* _acorr, repet.py:1108-1139: x[n_rows][n_cols] -> unbiased autocorrelation of every column, ac[lag][col]. */
int repet_acorr(repet_ctx* ctx, const float* x, int32_t n_rows, int32_t n_cols, float* ac_out);

/* _beatspectrum, repet.py:1142-1158 (input already squared by the caller, as in the reference):
 * p[T][F] -> beat[n_lags], n_lags <= T. */
int repet_beat_spectrum(repet_ctx* ctx, const float* p, int64_t n_frames, int32_t n_freq,
                        float* beat_out, int32_t n_lags);

/* _beatspectrogram, repet.py:1161-1206: p[T][F] -> beat[T][seg_len] (frame-major). */
int repet_beat_spectrogram(repet_ctx* ctx, const float* p, int64_t n_frames, int32_t n_freq,
                           int32_t seg_len, int32_t seg_step, float* beat_out);

/* _periods, repet.py:1249-1291: beat[n_cols][n_lags] -> period[n_cols]. */
int repet_periods(repet_ctx* ctx, const float* beat, int32_t n_cols, int32_t n_lags, int32_t period_lo,
                  int32_t period_hi, int32_t* period_out);

/* _localmaxima over every row of m[n_rows][n_cols], repet.py:1294-1383: idx[n_rows][number] (-1 padded,
 * sorted by value descending) and count[n_rows]. */
int repet_local_maxima(repet_ctx* ctx, const float* m, int32_t n_rows, int32_t n_cols, float min_value,
                       int32_t min_distance, int32_t number, int32_t* idx_out, int32_t* count_out);

/* _mask / _adaptivemask / _simmask, repet.py:1386-1545: v[T][F] -> mask[T][F] (no high-pass override). */
int repet_mask_period(repet_ctx* ctx, const float* v, int64_t n_frames, int32_t n_freq, int32_t period,
                      float* mask_out);
int repet_mask_adaptive(repet_ctx* ctx, const float* v, int64_t n_frames, int32_t n_freq,
                        const int32_t* periods, int32_t filter_order, float* mask_out);
int repet_mask_sim(repet_ctx* ctx, const float* v, int64_t n_frames, int32_t n_freq, const int32_t* idx,
                   const int32_t* count, int32_t number, float* mask_out);

/* repet_mask_sim through the rank transform below (what `sim` does on clips of more than 1 024 frames): path 1 = the packed
 * 16-bit selection network on the rank codes, path 2 = the bit-sliced selection on the same codes; both give the bits of
 * repet_mask_sim. n_freq - 1 a multiple of 128 (path 2: a power of two <= 2048), lists of at most 128 entries.
 * median_codes_out (nullable, path 2): out[n_frames][n_freq - 1] as repet_ctx_last_median_codes describes them. */
int repet_mask_sim_ranked(repet_ctx* ctx, const float* v, int64_t n_frames, int32_t n_freq, const int32_t* idx,
                          const int32_t* count, int32_t number, int32_t path, float* mask_out, uint32_t* median_codes_out);

/* Rank transform behind the median of `sim` (np.median over the similar frames is a selection, repet.py:1535: it only
 * needs the ORDER of a bin's magnitudes over the clip). v[T][F] -> codes_out[T][n] (0x0400 + number of frames whose
 * magnitude in that bin is strictly smaller) and sorted_out[n][T] (every bin's magnitudes in ascending order), for the
 * first n = F rounded down to a multiple of 128 bins; 1024 < n_frames <= 30720. */
int repet_rank_columns(repet_ctx* ctx, const float* v, int64_t n_frames, int32_t n_freq, uint16_t* codes_out,
                       float* sorted_out);

/* Integer intermediates of the last repet_ctx_execute (for index-set / period parity checks).
 * periods: original -> 1 value; extended -> one per segment; adaptive -> one per frame.
 * sim indices: idx[T][number] (-1 padded) + count[T]; simonline: rows for frames B-1..T-1, FRAME numbers
 * (of a batch context: n_rows = rows of one clip gives the first clip's, rows * n_clips all clips' back to back). */
int repet_ctx_last_periods(repet_ctx* ctx, int32_t* out, int32_t capacity, int32_t* n_written);
int repet_ctx_last_sim_indices(repet_ctx* ctx, int32_t* idx_out, int32_t* count_out, int32_t n_rows,
                               int32_t number);
int repet_ctx_last_frame_count(repet_ctx* ctx, int64_t* n_frames);
/* sim: which form of np.median (repet.py:1535) the last run took -- 0 the selection network on the float magnitudes,
 * 1 the packed 16-bit network on the rank codes (rank.hip), 2 the bit-sliced selection on the same codes (mask_bits.hip).
 * All three give the same bits; REPET_MEDIAN=f32|rank forces the first two. */
int repet_ctx_last_median_path(repet_ctx* ctx, int32_t* path);
/* After a run on path 2: what the selection left per cell, out[channel][frame][bin] for the first n_bins bins (n_bins <=
 * n_freq - 1: the Nyquist bin is not ranked): bits 0-14 the rank code (number of strictly smaller magnitudes of the bin over
 * the clip, rank.hip, without its 0x0400 base) of the lower median of the frame's similar frames, bit 15 set where that is
 * below the frame's own code, bits 16-30 the code of the upper median (equal to the lower one for an odd list). */
int repet_ctx_last_median_codes(repet_ctx* ctx, uint32_t* out, int64_t n_frames, int32_t n_bins);
/* sim / simonline: counters of the near-tie refinement of the last run's peak picking (_localmaxima,
 * repet.py:1294-1345): out[0] rows with a decision inside the fp32 tolerance, out[1] near-tied elements
 * re-decided from float64 similarities, out[2] decisions that changed, out[3] flat rows left to fp32. */
int repet_ctx_last_refine_stats(repet_ctx* ctx, int64_t out[4]);
/* The second level of the same peak picking: the reference decides on float64 similarities of complex128 spectra
 * (repet.py:149, :1220-1223, :1318-1326). Rows whose float64 re-decision on the fp32 spectra met a comparison the fp32
 * spectra cannot settle (closer than 2.5e-7), and flat rows, are decided again from float64 spectra computed from the
 * waveform. out[0] rows decided again, out[1] elements given float64 spectra, out[2] rows whose list changed,
 * out[3] largest |fp32-spectra value - float64-spectra value| met, in units of 1e-12, out[4] float64 unit rows computed,
 * out[5] 1 when the resident clip carries the fp32 remainders of a float64 upload (48-bit samples), out[6] rows taken up
 * again from the first pass's records (the fast path), out[7] of those, rows handed on to the general path. */
int repet_ctx_last_exact_stats(repet_ctx* ctx, int64_t out[8]);

/* ---- streaming online REPET-SIM (the reference's simonline needs the whole signal, repet.py:712-911) ------
 * open  : state for one stream of n_channels (any number, as repet.py:812 loops) with the parameters of derive_params(fs);
 * push  : feed n_samples more samples (NumPy C order); every frame that is now complete is processed and the
 *         newly final background samples (whole hops of W/2) are written to out (float64, interleaved,
 *         `capacity` samples per channel; n_samples + window_length always suffices), their count to *n_written;
 * finish: end of stream -- the zero-padded last frame (repet.py:781,813) and the tail; afterwards the total
 *         written equals the total pushed. REPET_ERR_TOO_SHORT if fewer samples were pushed than the reference's
 *         warm-up needs (repet.py:795-810).
 * The concatenated output equals repet.simonline(whole signal) bit for bit (same kernels, same order). */
typedef struct repet_online repet_online;
int repet_online_open(int device, int32_t n_channels, const repet_params* p, repet_online** out);
int repet_online_push(repet_online* h, const void* audio, int dtype, int64_t n_samples, double* out, int64_t capacity,
                      int64_t* n_written);
int repet_online_finish(repet_online* h, double* out, int64_t capacity, int64_t* n_written);
int repet_online_close(repet_online* h);

/* (ABI 4) Self-test of the host conversions a staged upload / download runs (float64 -> fp32 samples + fp32 remainders,
 * fp32 -> float64; non-temporal AVX-512 / AVX2 lines where the CPU has them) against scalar loops on n values with NaN,
 * infinities, denormals, PCM-exact runs and every misalignment. No GPU needed. Returns the number of values that differ
 * (0 = pass). */
int64_t repet_host_conversion_selftest(int64_t n, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif /* REPET_HIP_H */
