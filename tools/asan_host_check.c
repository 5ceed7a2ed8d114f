/* Host-side sanitizer run (make -C repet-python_amd/csrc asan; SURVEY.md section 5 "race detection / sanitizers").
 *
 * Linked against build_diag/librepet_hip_asan.so -- the library with its HOST code under AddressSanitizer and
 * UndefinedBehaviorSanitizer (device code untouched: -fno-gpu-sanitize) -- it drives every entry point of
 * include/repet_hip.h that does its work on the host: settings -> sizes, frame and segment counts, the network table,
 * the RIFF/WAVE header parser (well-formed files of every supported kind, then 200 000 truncated / mutated images),
 * and the argument checks of the context calls. Without a GPU the context calls must fail cleanly; with one, a 40-s
 * clip goes through upload -> execute -> download so that the orchestrator's host buffers are covered as well.
 * Any sanitizer report aborts the program (halt_on_error); a wrong answer returns non-zero. */
#include "repet_hip.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int failures = 0;
static int gpu_used = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                      \
        }                                                                    \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}

static void put16(uint8_t* p, unsigned v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put32(uint8_t* p, uint32_t v) { put16(p, v & 0xffff); put16(p + 2, v >> 16); }

/* RIFF/WAVE image: optional LIST chunk of odd length in front of fmt, plain or extensible fmt, data */
static size_t make_wav(uint8_t* buf, int format, int channels, int bits, int container_bytes, int frames, int extensible,
                       int with_list) {
    uint8_t* p = buf;
    memcpy(p, "RIFF", 4); p += 8;
    memcpy(p, "WAVE", 4); p += 4;
    if (with_list) {
        memcpy(p, "LIST", 4); put32(p + 4, 5); memcpy(p + 8, "INFOx", 5); p[13] = 0; p += 14;      /* odd size + pad byte */
    }
    const int fmt_size = extensible ? 40 : 16;
    memcpy(p, "fmt ", 4); put32(p + 4, (uint32_t)fmt_size); p += 8;
    put16(p, extensible ? 0xFFFEu : (unsigned)format);
    put16(p + 2, (unsigned)channels);
    put32(p + 4, 44100);
    put32(p + 8, (uint32_t)(44100 * channels * container_bytes));
    put16(p + 12, (unsigned)(channels * container_bytes));
    put16(p + 14, (unsigned)bits);
    if (extensible) {
        static const uint8_t tail[14] = {0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71};
        put16(p + 16, 22); put16(p + 18, (unsigned)bits); put32(p + 20, 0);
        put16(p + 24, (unsigned)format); memcpy(p + 26, tail, 14);
    }
    p += fmt_size;
    const uint32_t data_bytes = (uint32_t)(frames * channels * container_bytes);
    memcpy(p, "data", 4); put32(p + 4, data_bytes); p += 8;
    for (uint32_t i = 0; i < data_bytes; ++i) p[i] = (uint8_t)rnd();
    p += data_bytes;
    put32(buf + 4, (uint32_t)(p - buf - 8));
    return (size_t)(p - buf);
}

static void check_sizes(void) {
    repet_settings s;
    repet_default_settings(&s);
    repet_params p;
    CHECK(repet_abi_version() > 0);
    static const double rates[] = {8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 192000, 1, 7, 1e6};
    for (size_t i = 0; i < sizeof rates / sizeof rates[0]; ++i) {
        memset(&p, 0, sizeof p);
        const int rc = repet_derive_params(NULL, rates[i], &p);
        CHECK(rc == REPET_OK || repet_last_error()[0] != 0);
        if (rc == REPET_OK) {
            CHECK(p.window_length > 0 && (p.window_length & (p.window_length - 1)) == 0);      /* a power of two */
            CHECK(p.step_length > 0 && p.step_length <= p.window_length);
        }
        CHECK(repet_derive_params(&s, rates[i], &p) == rc);
    }
    CHECK(repet_derive_params(NULL, 0.0, &p) != REPET_OK);
    CHECK(repet_derive_params(NULL, -44100.0, &p) != REPET_OK);
    CHECK(repet_derive_params(NULL, NAN, &p) != REPET_OK);
    CHECK(repet_derive_params(NULL, 44100.0, NULL) != REPET_OK);
    CHECK(repet_derive_params(NULL, 44100.0, &p) == REPET_OK);
    for (int64_t n = -3; n < 70000; n += (n < 5000 ? 1 : 997)) {
        const int64_t t = repet_frame_count(n, p.window_length, p.step_length, 1);
        const int64_t u = repet_frame_count(n, p.window_length, p.step_length, 0);
        CHECK(n <= 0 || t >= 1);                                    /* centred: at least one frame (repet.py:1021-1028) */
        CHECK(n < p.window_length || u >= 1);                       /* online framing needs a whole window (repet.py:781) */
        (void)repet_extended_segment_count(n * 50, &p);
    }
    CHECK(repet_frame_count(1000, 0, 0, 1) <= 0 || repet_last_error() != NULL);
    (void)repet_extended_segment_count(1 << 20, NULL);
    for (int bound = -2; bound < 400; ++bound) {
        int32_t size = -1, instr = -1;
        const int rc = repet_median_network_info(bound, &size, &instr);
        if (rc == REPET_OK && size > 0) CHECK(size >= bound && instr > 0);
    }
    (void)repet_median_network_info(100, NULL, NULL);
}

static void check_wav_parser(void) {
    static uint8_t buf[1 << 16];
    repet_wav_info info;
    struct { int format, bits, bytes; } kinds[] = {{1, 8, 1}, {1, 16, 2}, {1, 24, 3}, {1, 24, 4}, {1, 32, 4}, {3, 32, 4}, {3, 64, 8}};
    for (size_t k = 0; k < sizeof kinds / sizeof kinds[0]; ++k)
        for (int channels = 1; channels <= 3; ++channels)
            for (int ext = 0; ext < 2; ++ext)
                for (int list = 0; list < 2; ++list) {
                    const int frames = 100 + (int)(rnd() % 400);
                    const size_t n = make_wav(buf, kinds[k].format, channels, kinds[k].bits, kinds[k].bytes, frames, ext, list);
                    memset(&info, 0, sizeof info);
                    const int rc = repet_wav_parse(buf, (int64_t)n, &info);
                    CHECK(rc == REPET_OK);
                    if (rc == REPET_OK) {
                        CHECK(info.format == kinds[k].format && info.n_channels == channels);
                        CHECK(info.bits_per_sample == kinds[k].bits && info.bytes_per_sample == kinds[k].bytes);
                        CHECK(info.n_samples == frames && info.sampling_frequency == 44100);
                        CHECK(info.data_offset > 0 && info.data_offset + (int64_t)frames * channels * kinds[k].bytes <= (int64_t)n);
                    }
                    /* every truncation of the image: an error or a shorter clip, never a read past the end. The copy
                     * sits at the END of an exactly-sized heap block so that one byte too far is a report. */
                    for (size_t cut = 0; cut < n; cut += (cut < 128 ? 1 : 37)) {
                        uint8_t* exact = (uint8_t*)malloc(cut ? cut : 1);
                        memcpy(exact, buf, cut);
                        if (repet_wav_parse(exact, (int64_t)cut, &info) == REPET_OK)
                            CHECK(info.data_offset + info.n_samples * info.n_channels * info.bytes_per_sample <= (int64_t)cut);
                        free(exact);
                    }
                }
    /* mutated images */
    for (int it = 0; it < 200000; ++it) {
        const int k = (int)(rnd() % (sizeof kinds / sizeof kinds[0]));
        const size_t n = make_wav(buf, kinds[k].format, 1 + (int)(rnd() % 2), kinds[k].bits, kinds[k].bytes, 8 + (int)(rnd() % 8),
                                  (int)(rnd() & 1), (int)(rnd() & 1));
        const int flips = 1 + (int)(rnd() % 4);
        for (int f = 0; f < flips; ++f) {
            const size_t at = rnd() % (n < 96 ? n : 96);            /* the headers are where the parser decides */
            buf[at] = (rnd() & 1) ? (uint8_t)rnd() : (uint8_t)(buf[at] ^ (1u << (rnd() % 8)));
        }
        const size_t len = (rnd() & 3) ? n : rnd() % (n + 1);
        uint8_t* exact = (uint8_t*)malloc(len ? len : 1);
        memcpy(exact, buf, len);
        if (repet_wav_parse(exact, (int64_t)len, &info) == REPET_OK) {
            CHECK(info.n_channels > 0 && info.bytes_per_sample > 0 && info.n_samples >= 0 && info.data_offset >= 0);
            CHECK(info.data_offset + info.n_samples * info.n_channels * info.bytes_per_sample <= (int64_t)len);
        }
        free(exact);
    }
    CHECK(repet_wav_parse(NULL, 100, &info) != REPET_OK);
    CHECK(repet_wav_parse(buf, 0, &info) != REPET_OK);
    CHECK(repet_wav_parse(buf, -5, &info) != REPET_OK);
    CHECK(repet_wav_parse(buf, 44, NULL) != REPET_OK);
}

static void check_context_calls(void) {
    repet_params p;
    CHECK(repet_derive_params(NULL, 16000.0, &p) == REPET_OK);
    double out[8];
    float fout[8];
    int64_t n64 = 0;
    int32_t n32 = 0;
    /* null handles */
    CHECK(repet_ctx_upload(NULL, out, REPET_F64, 4, 1) != REPET_OK);
    CHECK(repet_ctx_execute(NULL, REPET_SIM, &p, NULL) != REPET_OK);
    CHECK(repet_ctx_download(NULL, out) != REPET_OK);
    CHECK(repet_ctx_download_device(NULL, fout) != REPET_OK);
    CHECK(repet_ctx_last_frame_count(NULL, &n64) != REPET_OK);
    CHECK(repet_ctx_last_periods(NULL, &n32, 1, &n32) != REPET_OK);
    CHECK(repet_ctx_destroy(NULL) == REPET_OK);
    CHECK(repet_online_close(NULL) == REPET_OK || repet_last_error()[0] != 0);
    CHECK(repet_ctx_create(0, NULL) != REPET_OK);
    CHECK(repet_ctx_last_median_path(NULL, &n32) != REPET_OK);
    CHECK(repet_ctx_last_median_codes(NULL, NULL, 1, 1) != REPET_OK);
    CHECK(repet_ctx_download_input(NULL, fout, fout, &n32) != REPET_OK);
    CHECK(repet_mask_sim_ranked(NULL, fout, 2000, 1025, &n32, &n32, 100, 2, fout, NULL) != REPET_OK);

    const int devices = repet_device_count();
    repet_ctx* ctx = NULL;
    const int rc = repet_ctx_create(0, &ctx);
    if (devices <= 0) {
        CHECK(rc != REPET_OK && ctx == NULL && repet_last_error()[0] != 0);
        CHECK(repet_run(REPET_SIM, out, REPET_F64, 8, 1, &p, out, 0, NULL) != REPET_OK);
        printf("no GPU: context calls fail cleanly (%s)\n", repet_last_error());
        return;
    }
    CHECK(rc == REPET_OK && ctx != NULL);
    if (rc != REPET_OK) return;
    /* a 40-s two-channel clip through every variant: the orchestrator's host-side buffers under the sanitizers */
    const int64_t n = 40 * 16000;
    double* x = (double*)malloc((size_t)n * 2 * sizeof(double));
    double* y = (double*)malloc((size_t)n * 2 * sizeof(double));
    for (int64_t i = 0; i < 2 * n; ++i) x[i] = sin(0.01 * (double)(i / 2) * (1 + (i & 1))) * 0.3 + ((double)(rnd() % 2001) - 1000.0) * 1e-6;
    CHECK(repet_ctx_upload(ctx, x, REPET_F64, n, 2) == REPET_OK);
    (void)repet_ctx_upload(ctx, x, REPET_F64, 0, 2);                  /* an empty clip: accepted or refused, not a crash */
    CHECK(repet_ctx_upload(ctx, x, 99, n, 2) != REPET_OK);
    CHECK(repet_ctx_upload(ctx, x, REPET_F64, n, 0) != REPET_OK);
    CHECK(repet_ctx_upload(ctx, x, REPET_F64, n, 2) == REPET_OK);
    static const int algos[] = {REPET_ORIGINAL, REPET_EXTENDED, REPET_ADAPTIVE, REPET_SIM, REPET_SIMONLINE};
    for (size_t a = 0; a < sizeof algos / sizeof algos[0]; ++a) {
        const int rc_run = repet_ctx_execute(ctx, algos[a], &p, NULL);
        if (rc_run != REPET_OK) fprintf(stderr, "variant %d: %s\n", algos[a], repet_last_error());
        CHECK(rc_run == REPET_OK);
        CHECK(repet_ctx_download(ctx, y) == REPET_OK);
        double energy = 0;
        for (int64_t i = 0; i < 2 * n; ++i) energy += y[i] * y[i];
        CHECK(isfinite(energy) && energy > 0);
    }
    /* sim once more: 1 251 frames take the rank transform and the bit-sliced median; its per-cell words come back */
    CHECK(repet_ctx_execute(ctx, REPET_SIM, &p, NULL) == REPET_OK);
    CHECK(repet_ctx_last_median_path(ctx, &n32) == REPET_OK && n32 == 2);
    CHECK(repet_ctx_last_frame_count(ctx, &n64) == REPET_OK && n64 > 1024);
    {
        const int32_t bins = p.window_length / 2;
        uint32_t* codes = (uint32_t*)malloc((size_t)2 * (size_t)n64 * (size_t)bins * sizeof(uint32_t));
        CHECK(repet_ctx_last_median_codes(ctx, codes, n64, bins) == REPET_OK);
        CHECK(repet_ctx_last_median_codes(ctx, codes, n64 + 1, bins) != REPET_OK);
        CHECK(repet_ctx_last_median_codes(ctx, codes, n64, 100000) != REPET_OK);
        uint32_t top = 0;
        for (size_t i = 0; i < (size_t)2 * (size_t)n64 * (size_t)bins; ++i) top = (codes[i] & 0x7fffu) > top ? (codes[i] & 0x7fffu) : top;
        CHECK(top > 0 && top < (uint32_t)n64);          /* ranks of a column of n64 frames */
        free(codes);
    }
    CHECK(repet_mask_sim_ranked(ctx, fout, 100, 1025, &n32, &n32, 100, 2, fout, NULL) != REPET_OK);       /* too few frames */
    CHECK(repet_mask_sim_ranked(ctx, fout, 2000, 1025, &n32, &n32, 100, 3, fout, NULL) != REPET_OK);      /* no such path */
    CHECK(repet_ctx_execute(ctx, 77, &p, NULL) != REPET_OK);
    CHECK(repet_ctx_execute(ctx, REPET_SIM, NULL, NULL) != REPET_OK);
    CHECK(repet_ctx_last_frame_count(ctx, &n64) == REPET_OK && n64 > 0);
    free(x);
    free(y);
    CHECK(repet_ctx_destroy(ctx) == REPET_OK);
    printf("GPU present: five variants on a 40-s clip through the sanitized host code\n");
    gpu_used = 1;
}

int main(void) {
    check_sizes();
    check_wav_parser();
    check_context_calls();
    /* the host conversions (non-temporal lines, PCM-exact blocks) against scalar loops, every misalignment: under ASan / UBSan */
    CHECK(repet_host_conversion_selftest(70001, 9u) == 0);
    CHECK(repet_host_conversion_selftest(4095, 10u) == 0);
    CHECK(repet_host_conversion_selftest(0, 1u) == -1);
    if (failures) {
        fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    printf("asan_host_check: ok\n");
    fflush(stdout);
    fflush(stderr);
    /* Leave without the process-exit finalizers once a GPU was used: libamdhip64's own __cxa_finalize frees HSA memory
       through the sanitizer's device allocator after that allocator has marked the device runtime unloaded (ASan CHECK
       "dev_runtime_unloaded_" in sanitizer_allocator_device.h, two runs of three on the pool). Everything this program
       checks -- the library's host code -- has run and been reported by then; the library's own objects were destroyed
       by repet_ctx_destroy above. */
    if (gpu_used) _exit(0);
    return 0;
}
