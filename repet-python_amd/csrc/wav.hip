// WAVE files either side of the path: wavread / wavwrite of the reference (repet.py:914-946) are
// scipy.io.wavfile plus one division; here a file image is parsed on the host, its raw PCM bytes travel through the
// pinned ring as they are (2 or 3 bytes per sample instead of 8) and are decoded AND normalised on the device, and a
// result leaves as a complete file image. Conventions reproduced (they are scipy's and the reference's, checked in
// tests/test_wav.py against scipy.io.wavfile itself):
//   * 8-bit PCM is unsigned, 16/32-bit signed, 24-bit PCM reads as int32 with the sample in the TOP three bytes;
//   * wavread divides by 2^(8 * itemsize - 1) of the array SciPy returned (repet.py:929) -- 2^7 for uint8 (so 8-bit
//     files come out in [0, 2)), 2^15, 2^31 for 24- and 32-bit PCM, and also 2^31 / 2^63 for FLOAT files (a quirk: a
//     float32 file at full scale reads as 4.7e-10; reproduced, not fixed);
//   * wavwrite of a float64 array writes IEEE-float format with an 18-byte fmt chunk and a fact chunk.
#include "../../include/repet_hip.h"
#include "common.h"

#include <cstring>
#include <string>

namespace repet {

namespace {
inline uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
}  // namespace

// 0 = ok; otherwise a message (static storage)
const char* wav_parse(const void* file, int64_t n_bytes, repet_wav_info* info) {
    const unsigned char* b = static_cast<const unsigned char*>(file);
    if (!b || !info || n_bytes < 12) return "not a WAVE file (too short)";
    if (std::memcmp(b, "RIFF", 4) != 0 || std::memcmp(b + 8, "WAVE", 4) != 0)
        return std::memcmp(b, "RIFX", 4) == 0 || std::memcmp(b, "RF64", 4) == 0 ? "RIFX / RF64 files are not handled here"
                                                                                : "not a WAVE file (no RIFF....WAVE header)";
    std::memset(info, 0, sizeof(*info));
    bool have_fmt = false;
    int block_align = 0;
    int64_t pos = 12;
    while (pos + 8 <= n_bytes) {
        const unsigned char* ck = b + pos;
        const int64_t size = rd32(ck + 4);
        const int64_t body = pos + 8;
        if (std::memcmp(ck, "fmt ", 4) == 0) {
            if (size < 16 || body + 16 > n_bytes) return "truncated fmt chunk";
            int tag = rd16(b + body);
            info->n_channels = rd16(b + body + 2);
            info->sampling_frequency = (int32_t)rd32(b + body + 4);
            block_align = rd16(b + body + 12);
            info->bits_per_sample = rd16(b + body + 14);
            if (tag == 0xFFFE) {                     // WAVE_FORMAT_EXTENSIBLE: the real tag opens the sub-format GUID
                if (size < 40 || body + 26 > n_bytes) return "truncated extensible fmt chunk";
                tag = rd16(b + body + 24);
            }
            info->format = tag;
            have_fmt = true;
        } else if (std::memcmp(ck, "data", 4) == 0) {
            if (!have_fmt) return "data chunk before fmt chunk";
            if (info->n_channels < 1 || block_align < 1) return "bad channel count or block alignment";
            info->bytes_per_sample = block_align / info->n_channels;
            if (info->bytes_per_sample * info->n_channels != block_align) return "block alignment is not a multiple of the channel count";
            const bool pcm = info->format == 1, flt = info->format == 3;
            const int w = info->bytes_per_sample;
            if (!(pcm && (w == 1 || w == 2 || w == 3 || w == 4)) && !(flt && (w == 4 || w == 8)))
                return "unsupported sample format (PCM 8/16/24/32-bit and IEEE float 32/64-bit are handled)";
            int64_t avail = n_bytes - body;
            int64_t bytes = size < avail ? size : avail;      // a size of 0xFFFFFFFF / a truncated file: what is there
            info->data_offset = body;
            info->n_samples = bytes / block_align;
            return nullptr;
        }
        pos = body + size + (size & 1);              // chunks are word-aligned
    }
    return have_fmt ? "no data chunk" : "no fmt chunk";
}

// element width SciPy's array has for this file: what wavread's normalisation 2^(8 * itemsize - 1) sees (repet.py:929)
int wav_itemsize(const repet_wav_info& w) { return w.bytes_per_sample == 3 ? 4 : w.bytes_per_sample; }

// ---- device decode + normalisation ----------------------------------------------------------------------------------
__global__ void decode_pcm_kernel(const unsigned char* __restrict__ raw, int format, int width, float* __restrict__ dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const unsigned char* p = raw + i * width;
        float v;
        if (format == 3) {
            if (width == 4) { float f; memcpy(&f, p, 4); v = f * 4.656612873077393e-10f; }                    // / 2^31 (the quirk)
            else { double d; memcpy(&d, p, 8); v = (float)(d * 1.0842021724855044e-19); }                    // / 2^63
        } else if (width == 1) v = (float)p[0] * 0.0078125f;                                                  // uint8 / 2^7
        else if (width == 2) v = (float)(short)(p[0] | (p[1] << 8)) * 3.0517578125e-05f;                      // / 2^15
        else if (width == 3) v = (float)((int)((unsigned)p[0] << 8 | (unsigned)p[1] << 16 | (unsigned)p[2] << 24) >> 8) * 1.1920928955078125e-07f;   // (s24 << 8) / 2^31
        else v = (float)(int)((unsigned)p[0] | (unsigned)p[1] << 8 | (unsigned)p[2] << 16 | (unsigned)p[3] << 24) * 4.656612873077393e-10f;          // / 2^31
        dst[i] = v;
    }
}

hipError_t launch_decode_pcm(const void* raw, int format, int width, float* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int64_t blocks = ceil_div(n, 256);
    hipLaunchKernelGGL(decode_pcm_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s,
                       static_cast<const unsigned char*>(raw), format, width, dst, n);
    return hipGetLastError();
}

// ---- file image of a float result, as scipy.io.wavfile.write lays it out -------------------------------------------
int64_t wav_float_header(unsigned char* out, int sampling_frequency, int n_channels, int64_t n_samples, int item_bytes) {
    const int64_t data_bytes = n_samples * n_channels * item_bytes;
    auto wr16 = [](unsigned char* p, unsigned v) { p[0] = v & 255; p[1] = (v >> 8) & 255; };
    auto wr32 = [](unsigned char* p, uint64_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = (v >> 24) & 255; };
    unsigned char* p = out;
    std::memcpy(p, "RIFF", 4); wr32(p + 4, (uint64_t)(4 + 8 + 18 + 8 + 4 + 8 + data_bytes)); std::memcpy(p + 8, "WAVE", 4); p += 12;
    std::memcpy(p, "fmt ", 4); wr32(p + 4, 18); p += 8;
    wr16(p, 3); wr16(p + 2, (unsigned)n_channels); wr32(p + 4, (uint64_t)sampling_frequency);
    wr32(p + 8, (uint64_t)sampling_frequency * item_bytes * n_channels); wr16(p + 12, (unsigned)(n_channels * item_bytes));
    wr16(p + 14, (unsigned)(item_bytes * 8)); wr16(p + 16, 0); p += 18;
    std::memcpy(p, "fact", 4); wr32(p + 4, 4); wr32(p + 8, (uint64_t)n_samples); p += 12;
    std::memcpy(p, "data", 4); wr32(p + 4, (uint64_t)data_bytes); p += 8;
    return p - out;                                   // 58
}

}  // namespace repet
