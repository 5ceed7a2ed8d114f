/* A host that is not Python: separates a raw float64 clip with the C ABI alone.
 *
 *   gcc -I include examples/c_client.c -o c_client -L repet-python_amd/lib -lrepet_hip -Wl,-rpath,$PWD/repet-python_amd/lib
 *   ./c_client <algo 0..4> <fs> <channels> <in.f64> <out.f64>
 *
 * in.f64 / out.f64: interleaved float64 samples, NumPy C order (number_samples, number_channels) -- what
 * repet.<algo>(audio_signal, fs) takes and returns (repet.py:67,205,422,571,712). Parameters are the
 * reference's defaults (repet_default_settings); the GPU test compares the output with the Python drop-in. */
#include <stdio.h>
#include <stdlib.h>

#include "repet_hip.h"

int main(int argc, char** argv) {
    if (argc != 6) {
        fprintf(stderr, "usage: %s algo fs channels in.f64 out.f64\n", argv[0]);
        return 2;
    }
    const int algo = atoi(argv[1]);
    const double fs = atof(argv[2]);
    const int channels = atoi(argv[3]);
    FILE* in = fopen(argv[4], "rb");
    if (!in) { perror(argv[4]); return 2; }
    fseek(in, 0, SEEK_END);
    const long bytes = ftell(in);
    fseek(in, 0, SEEK_SET);
    const int64_t n_samples = bytes / (long)sizeof(double) / channels;
    double* audio = (double*)malloc((size_t)bytes);
    double* background = (double*)malloc((size_t)bytes);
    if (!audio || !background || fread(audio, 1, (size_t)bytes, in) != (size_t)bytes) { fprintf(stderr, "read failed\n"); return 2; }
    fclose(in);

    if (repet_abi_version() != REPET_ABI_VERSION || repet_device_count() < 1) {
        fprintf(stderr, "no usable librepet_hip / HIP device\n");
        return 3;
    }
    repet_settings settings;
    repet_params params;
    repet_default_settings(&settings);
    if (repet_derive_params(&settings, fs, &params) != REPET_OK) { fprintf(stderr, "%s\n", repet_last_error()); return 4; }
    repet_timing timing;
    const int rc = repet_run(algo, audio, REPET_F64, n_samples, channels, &params, background, 0, &timing);
    if (rc != REPET_OK) {
        fprintf(stderr, "repet_run failed (%d): %s\n", rc, repet_last_error());
        return 4;
    }
    printf("separated %lld samples x %d channels in %.3f ms of device time (%d stages)\n", (long long)n_samples, channels,
           timing.total_ms, timing.n_stages);
    FILE* out = fopen(argv[5], "wb");
    if (!out || fwrite(background, 1, (size_t)bytes, out) != (size_t)bytes) { perror(argv[5]); return 2; }
    fclose(out);
    free(audio);
    free(background);
    return 0;
}
