// K4 / K4b, wavefront-per-row form: peak picking of REPET-SIM (_localmaxima / _indices, repet.py:1294-1383) with ONE
// WAVEFRONT per row and no workgroup barrier anywhere.
//
// The workgroup-per-row kernel (peaks.hip) spends its time waiting: nine barriers per row, four workgroups per CU (its
// row lives in 31 KB of LDS), every phase a chain of memory / LDS latencies -- 70 000 cycles per row for about 4 000
// cycles of arithmetic. Here a wave walks its row in chunks of 1 024 elements (16 per lane, with a halo of d on both
// sides), so that sixteen rows are in flight per CU and nothing ever waits for another wave:
//   * a chunk is loaded with coalesced 16-byte loads, turned lane-contiguous through a padded wave-private LDS buffer;
//   * the window maxima M_w[i] = max(v[i .. i+w-1]), w = 2^floor(log2 d), are doubled up IN REGISTERS: a partner that
//     sits in the next lane's registers comes in through a DPP wave shift (v_max_f32_dpp ... wave_shl:1), no LDS;
//   * M_w goes back to the LDS buffer once, and the strict test reads its four windows from there as in peaks.hip:
//         left = max(M_w[i-d], M_w[i-w]),  right = max(M_w[i+1], M_w[i+d-w+1]);
//   * survivors and near-ties are compacted with wave ballots into wave-private lists (counters live in SGPRs, no
//     atomics); rivals of a near-tie are looked for in the row itself (cache-resident), the float64 re-decision and
//     the ranking are the ones of peaks.hip, run by the wave alone.
// Same decisions, element for element, as peaks.hip (tests compare both against the oracle); used for windows of up to
// 63 elements (every default configuration: d = 31, 43, 47) and candidate lists of up to 640 entries, otherwise the
// launcher falls back to the workgroup kernel.
#include "peaks.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace repet {

// Diagnostics (make stamps, tools/peak_stamps.py). REPET_PEAK_SPANS: (start, end) of every row's wave on the chip-wide 100 MHz
// clock -- two scalar reads and one store per row, the same registers as the shipped kernel. REPET_PEAK_STAMPS: cycles per
// phase of a few rows (costs registers: the stamped kernel holds three waves per SIMD where the shipped one holds four).
#if defined(REPET_PEAK_STAMPS) || defined(REPET_PEAK_SPANS)
__device__ unsigned long long g_wave_stamps[8 * 8];
__device__ unsigned long long g_wave_span[2 * 16384];
__device__ unsigned int g_wave_phase[10 * 16384];           // REPET_PEAK_STAMPS: cycles per phase of EVERY row
#endif
#ifdef REPET_PEAK_STAMPS
#define WSTAMP_DECL unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#define WSTAMP(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_[k] += now_ - last_; last_ = now_; }
#define WSTAMP_OUT if (lane == 0 && (r % 997) == 5 && r / 997 < 8) { for (int k_ = 0; k_ < 8; ++k_) g_wave_stamps[(r / 997) * 8 + k_] = acc_[k_]; } \
    if (lane == 0 && r < 16384) { for (int k_ = 0; k_ < 8; ++k_) g_wave_phase[10 * r + k_] = (unsigned)acc_[k_]; g_wave_phase[10 * r + 8] = (unsigned)n_amb; g_wave_phase[10 * r + 9] = (unsigned)n_peak; }
#elif defined(REPET_PEAK_SPANS)
#define WSTAMP_DECL const unsigned long long span0_ = __builtin_amdgcn_s_memrealtime();
#define WSTAMP(k)
#define WSTAMP_OUT if (lane == 0 && r < 16384) { g_wave_span[2 * r] = span0_; g_wave_span[2 * r + 1] = __builtin_amdgcn_s_memrealtime(); }
#else
#define WSTAMP_DECL
#define WSTAMP(k)
#define WSTAMP_OUT
#endif

namespace {

constexpr int kChunkElems = 1024;                 // elements loaded per chunk: 64 lanes x 16
constexpr int kChunkGroups = kChunkElems / 4;     // float4 groups
constexpr int kBufGroups = kChunkGroups + kChunkGroups / 4 + 4;   // one pad group per 4 (a lane's 4 groups sit 5 apart) + slack
constexpr int kMaxWaveCap = 640;

__device__ __forceinline__ int phys4(int g) { return g + (g >> 2); }

// value of lane + 1 (lane 63: `old`)
__device__ __forceinline__ float from_next_lane(float v, float old) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// slot of this lane among the lanes with `flag`, counted from `base` (wave-uniform); returns the new base in *next
__device__ __forceinline__ int ballot_slot(bool flag, int base, int lane, int* next) {
    const unsigned long long ballot = __ballot(flag);
    *next = base + __popcll(ballot);
    return base + __popcll(ballot & ((1ull << lane) - 1ull));
}

constexpr int kCandCap = 192;         // flagged elements buffered between the cheap sweep and the full decision

struct WaveLds {
    float4* buf;            // kBufGroups float4: raw chunk, then its window maxima; later the candidate ranks
    float* pval; int* pidx; // candidates (cap)
    double* amb_exact; double* riv_exact;
    int* amb_idx; int* riv_idx; float* amb_val;
    short* riv_owner; short* riv_ref; short* unl_list;
    unsigned char* amb_ok; unsigned char* amb_lose;
    float* cutv;
    // elements the sweep could not rule out: (index, value, max of its two windows). Overlays amb_exact .. unl_list,
    // which are only used after the sweep.
    int* cand_i; float* cand_v; float* cand_m;
};

__host__ __device__ inline size_t wave_lds_bytes(int cap, int buf_groups = kBufGroups) {
    return (size_t)buf_groups * 16 + (size_t)cap * 8 + kAmbCap * 8 + kRivalCap * 8 + kAmbCap * 4 + kRivalCap * 4 + kAmbCap * 4 +
           3 * kRivalCap * 2 + 2 * kAmbCap + 16;
}

__device__ __forceinline__ WaveLds carve(unsigned char* base, int cap, int buf_groups = kBufGroups) {
    WaveLds w;
    unsigned char* p = base;
    w.buf = reinterpret_cast<float4*>(p); p += (size_t)buf_groups * 16;
    unsigned char* post = p;                                  // 2 496 bytes used only after the sweep
    w.amb_exact = reinterpret_cast<double*>(p); p += kAmbCap * 8;
    w.riv_exact = reinterpret_cast<double*>(p); p += kRivalCap * 8;
    w.riv_idx = reinterpret_cast<int*>(p); p += kRivalCap * 4;
    w.riv_owner = reinterpret_cast<short*>(p); p += kRivalCap * 2;
    w.riv_ref = reinterpret_cast<short*>(p); p += kRivalCap * 2;
    w.unl_list = reinterpret_cast<short*>(p); p += kRivalCap * 2;
    static_assert(kAmbCap * 8 + kRivalCap * 8 + kRivalCap * 4 + 3 * kRivalCap * 2 >= kCandCap * 12, "candidate buffer must fit its overlay");
    w.cand_i = reinterpret_cast<int*>(post);
    w.cand_v = reinterpret_cast<float*>(post + kCandCap * 4);
    w.cand_m = reinterpret_cast<float*>(post + kCandCap * 8);
    w.pval = reinterpret_cast<float*>(p); p += (size_t)cap * 4;
    w.pidx = reinterpret_cast<int*>(p); p += (size_t)cap * 4;
    w.amb_idx = reinterpret_cast<int*>(p); p += kAmbCap * 4;
    w.amb_val = reinterpret_cast<float*>(p); p += kAmbCap * 4;
    w.cutv = reinterpret_cast<float*>(p); p += 16;
    w.amb_ok = p; p += kAmbCap;
    w.amb_lose = p;
    return w;
}

// LDS traffic of one wave is ordered by the hardware; this only stops the compiler from moving accesses across
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace

// The band of candidates around the top-`number` cut, ranked by the float64 values in L.amb_exact (higher index first on
// ties): members that stay are written to the list. Returns false when the cut itself -- the lowest value kept against
// the highest one dropped -- is closer than delta2; L.amb_lose then marks the members that may belong on the other side.
template <bool LEVEL2, class OutIndex>
__device__ __forceinline__ bool rank_cut_band(const PeakArgs& a, const WaveLds& L, int lane, int n_band, int n_above, int* out,
                                              OutIndex out_index, bool count_stats) {
    int changed = 0;
    double low_kept = INFINITY, high_dropped = -INFINITY;
    for (int k = lane; k < n_band; k += 64) {
        const double e = L.amb_exact[k];
        const int i = L.amb_idx[k];
        int crank = 0;
        for (int t = 0; t < n_band; ++t) crank += (L.amb_exact[t] > e) || (L.amb_exact[t] == e && L.amb_idx[t] > i);
        const bool keep = n_above + crank < a.number;
        if (keep) out[n_above + crank] = out_index(i);
        if (keep) low_kept = fmin(low_kept, e); else high_dropped = fmax(high_dropped, e);
        changed += (keep != (L.amb_ok[k] != 0));
        L.amb_lose[k] = keep ? 1 : 0;                          // (kept, for the marking below)
    }
    for (int sh = 32; sh > 0; sh >>= 1) {
        low_kept = fmin(low_kept, __shfl_xor(low_kept, sh));
        high_dropped = fmax(high_dropped, __shfl_xor(high_dropped, sh));
    }
    if (count_stats) {
        if (lane == 0) stat_add(a.stats, 1, (unsigned)n_band);
        if (changed) stat_add(a.stats, 2, (unsigned)changed);
    }
    const bool settled = !(low_kept - high_dropped < a.delta2) || !(a.redo_list || LEVEL2);
    wave_sync();
    if (!settled)
        for (int k = lane; k < n_band; k += 64) {
            const double e = L.amb_exact[k];
            const bool kept_now = L.amb_lose[k] != 0;
            L.amb_lose[k] = ((kept_now && e < high_dropped + a.delta2) || (!kept_now && e > low_kept - a.delta2)) ? 1 : 0;
        }
    wave_sync();
    return settled;
}


// where the lite kernel finds the float64 unit rows (written by launch_unit_rows_f64 earlier on the stream)
struct LiteSource {
    const double* u64; const unsigned int* u64_gen; int64_t u64_clip_stride, gen_clip_stride; int FS; unsigned int gen;
    // (lite kernel) the record's fp32-safe candidates, copied into the wave's lists only when a verdict changes
    const float* rec_pval; const int* rec_pidx; int rec_np;
};

// (lite kernel) the marked members of the band get their level-2 values (one pipelined list), then the band is ranked again
template <class OutIndex, class ElemFrame>
__device__ __forceinline__ void finish_band_level2(const PeakArgs& a, const WaveLds& L, int lane, int n_band, int n_above, int* out,
                                                   OutIndex out_index, ElemFrame elem_frame, const LiteSource& ls, int clip,
                                                   int64_t self_frame, bool* missing) {
    short* items = L.unl_list;
    int n2 = 0;
    for (int k0 = 0; k0 < n_band; k0 += 64) {
        const bool f = k0 + lane < n_band && L.amb_lose[k0 + lane];
        int next;
        const int slot = ballot_slot(f, n2, lane, &next);
        if (f && slot < kRivalCap) items[slot] = (short)(k0 + lane);
        n2 = next;
    }
    wave_sync();
    double worst = 0.0;
    const bool have = level2_similarity_list(
        ls.u64 + (int64_t)clip * ls.u64_clip_stride, ls.u64_gen + (int64_t)clip * ls.gen_clip_stride, ls.gen, self_frame, ls.FS, lane, n2,
        [&](int it) -> int64_t { return elem_frame(L.amb_idx[items[it]]); },
        [&](int it, double e2) {
            const int k = items[it];
            worst = fmax(worst, fabs(e2 - L.amb_exact[k]));
            wave_sync();
            if (lane == 0) L.amb_exact[k] = e2;
        });
    wave_sync();
    if (!have) { *missing = true; return; }
    if (lane == 0 && a.stats) {
        stat_add(a.stats, 6, (unsigned)n2);
        if (worst > 0.0) stat_max(a.stats, 8, (unsigned)fmin(worst * 1e12, 4.0e9));
    }
    for (int k = lane; k < n_band; k += 64) {
        const double e = L.amb_exact[k];
        const int i = L.amb_idx[k];
        int crank = 0;
        for (int t = 0; t < n_band; ++t) crank += (L.amb_exact[t] > e) || (L.amb_exact[t] == e && L.amb_idx[t] > i);
        if (n_above + crank < a.number) out[n_above + crank] = out_index(i);
    }
}

// ---- second level, fast path (the general path is peaks_exact.hip) --------------------------------------
// A row whose float64 verdicts on the fp32 spectra (level 1) are closer than delta2 leaves a RECORD of the wave's lists
// where the close verdict was met, and the frames behind the elements involved go on a queue; a lean kernel computes their
// float64 unit rows (launch_unit_rows_f64, peaks_exact.hip), and local_maxima_lite_kernel takes the row up again from the
// record with those values (level 2) -- nothing of the row is scanned twice.
//   type 1: met in the near-tie verdicts: the lists as they are after the level-1 values (candidates safe in fp32, near-tied
//           elements, their rivals, all level-1 values) -- and, when the cut was close as well, its band; the lite kernel
//           re-takes the verdicts and, if one changes, ranking and cut (else only the recorded band, if any);
//   type 2: met only at the top-`number` cut: the ranked candidates and the band's level-1 values; the lite kernel
//           re-ranks the band.
struct LiteHeader { int type, n_peak, n_near, n_rival, n_unl, n_band, n_above, np; };
__host__ __device__ inline size_t lite_record_bytes(int cap) {
    return 32 + (size_t)kAmbCap * (4 + 8) * 2 + (size_t)kRivalCap * (2 + 2 + 2 + 2 + 4 + 8) + (size_t)cap * 12;
}
struct LiteRecord {
    LiteHeader* h; int* amb_idx; double* amb_exact; int* band_idx; double* band_exact; short* riv_owner; short* riv_ref; short* unl_list; int* riv_idx;
    double* riv_exact; float* pval; int* pidx; int* prank;
};
__device__ __forceinline__ LiteRecord carve_record(unsigned char* p, int cap) {
    LiteRecord q;
    q.h = reinterpret_cast<LiteHeader*>(p); p += 32;
    q.amb_exact = reinterpret_cast<double*>(p); p += kAmbCap * 8;
    q.band_exact = reinterpret_cast<double*>(p); p += kAmbCap * 8;
    q.riv_exact = reinterpret_cast<double*>(p); p += kRivalCap * 8;
    q.amb_idx = reinterpret_cast<int*>(p); p += kAmbCap * 4;
    q.band_idx = reinterpret_cast<int*>(p); p += kAmbCap * 4;
    q.riv_idx = reinterpret_cast<int*>(p); p += kRivalCap * 4;
    q.riv_owner = reinterpret_cast<short*>(p); p += kRivalCap * 2;
    q.riv_ref = reinterpret_cast<short*>(p); p += kRivalCap * 2;
    q.unl_list = reinterpret_cast<short*>(p); p += kRivalCap * 4;          // (+ padding to a 4-byte boundary)
    q.pval = reinterpret_cast<float*>(p); p += (size_t)cap * 4;
    q.pidx = reinterpret_cast<int*>(p); p += (size_t)cap * 4;
    q.prank = reinterpret_cast<int*>(p);
    return q;
}

// Everything of a row after the sweep: near-tie verdicts, ranking, top-`number` cut, the list. LEVEL2 = false: the first
// pass (records and queue entries are written where a verdict is closer than delta2). LEVEL2 = true: the lite kernel, whose
// lists and level-1 values come from a type-1 record (`rec`) and whose close verdicts are re-taken from float64 spectra.
template <bool LEVEL2>
__device__ __forceinline__ void wave_finish_row(const PeakArgs& a, const WaveLds& L, int lane, int64_t r, int64_t j, int clip, float dlt,
                                                int n_peak, int n_amb, int n_riv, int n_unl, const LiteSource* ls, bool* missing, bool* unchanged) {
    const int n = a.n;
    auto elem_frame = [&](int i) -> int64_t {                 // unit row of the frame behind element i
        if (a.mode == 0) return i;
        int l = (int)(j - i) % n;
        if (l < 0) l += n;
        return j - l - a.shift;
    };
    auto elem_row = [&](int i) -> const float* { return a.unit + elem_frame(i) * (int64_t)a.unit_pitch; };
    auto out_index = [&](int i) -> int { return a.mode == 0 ? i : (int)elem_frame(i); };
    const int64_t self_frame = j - a.shift;
    int lite_slot = -1;                                       // (first pass) this row's record

    if (dlt > 0.0f && n_amb > 0) {
        // Near-tie refinement (see peaks.hip): float64 similarities of the same fp32 spectra decide.
        const int n_near = n_amb, n_rival = n_riv, n_items = n_near + n_unl;
        if constexpr (!LEVEL2) {
            const int len4 = a.unit_pitch >> 2;
            const float* self_row = a.unit + self_frame * (int64_t)a.unit_pitch;
            auto item_row = [&](int it) -> const float* {
                return elem_row(it < n_near ? L.amb_idx[it] : L.riv_idx[L.unl_list[it - n_near]]);
            };
            auto keep = [&](int it, double e) {
                if (lane == 0) {
                    if (it < n_near) L.amb_exact[it] = e; else L.riv_exact[L.unl_list[it - n_near]] = e;
                }
            };
            if (a.unit_norm && len4 <= 320) {
                auto item_norm = [&](int it) -> double {
                    return a.unit_norm[elem_frame(it < n_near ? L.amb_idx[it] : L.riv_idx[L.unl_list[it - n_near]])];
                };
                exact_similarity_list_normed(self_row, a.unit_norm[self_frame], len4, lane, n_items, item_row, item_norm, keep);
            } else
                exact_similarity_list(self_row, len4, lane, n_items, item_row, keep);
            wave_sync();
        }
        // best[s]: the largest rival value of near-tied element s (similarities of magnitude spectra are >= 0: their bit
        // patterns order like unsigned integers; a NaN rival's pattern is above every number, and NaN - x is never "close")
        unsigned long long* best = reinterpret_cast<unsigned long long*>(L.buf);      // the window maxima are no longer needed
        unsigned char* amb_n2 = reinterpret_cast<unsigned char*>(L.buf) + kAmbCap * 8;   // elements / rivals that need level 2
        unsigned char* riv_n2 = amb_n2 + kAmbCap;
        auto verdicts = [&]() {                                // amb_lose and best from the current values
            for (int k = lane; k < n_near; k += 64) { best[k] = 0ull; L.amb_lose[k] = 0; }
            wave_sync();
            for (int e = lane; e < n_rival; e += 64) {
                const int s = L.riv_owner[e], ref = L.riv_ref[e];
                const double er = ref >= 0 ? L.amb_exact[ref] : L.riv_exact[e];
                if (!(L.amb_exact[s] > er)) L.amb_lose[s] = 1;
                atomicMax(&best[s], (unsigned long long)__double_as_longlong(er));
            }
            wave_sync();
        };
        verdicts();
        if constexpr (LEVEL2)                                  // the level-1 verdicts, to see whether level 2 changes any
            for (int k = lane; k < n_near; k += 64) L.amb_ok[k] = (!L.amb_lose[k] && L.amb_exact[k] >= a.min_value64) ? 1 : 0;
        // A verdict the fp32 spectra cannot settle: the element against its best rival, or against the threshold, closer
        // than delta2. It is re-taken from float64 spectra of the element and of every rival that may be the best one.
        bool close = false;
        if (a.redo_list || LEVEL2) {
            for (int k = lane; k < n_near; k += 64) {
                const double ek = L.amb_exact[k];
                const bool versus_rival = fabs(ek - __longlong_as_double((long long)best[k])) < a.delta2;
                amb_n2[k] = (versus_rival || fabs(ek - a.min_value64) < a.delta2) ? (versus_rival ? 3 : 1) : 0;
                close = close || amb_n2[k] != 0;
            }
            close = __any(close);
        }
        if (close) {
            wave_sync();
            for (int e = lane; e < n_rival; e += 64) {
                const int s = L.riv_owner[e], ref = L.riv_ref[e];
                const double er = ref >= 0 ? L.amb_exact[ref] : L.riv_exact[e];
                const bool need = (amb_n2[s] & 2) && er > __longlong_as_double((long long)best[s]) - a.delta2;
                riv_n2[e] = (need && ref < 0) ? 1 : 0;
                if (need && ref >= 0 && !amb_n2[ref]) amb_n2[ref] = 1;      // (racing writers all store 1)
            }
            wave_sync();
            if constexpr (!LEVEL2) {
                // the frames go on the queue, the lists into the row's record
                if (lane == 0) enqueue_frame_for_exact(a, clip, self_frame);
                for (int k = lane; k < n_near; k += 64) if (amb_n2[k]) enqueue_frame_for_exact(a, clip, elem_frame(L.amb_idx[k]));
                for (int e = lane; e < n_rival; e += 64) if (riv_n2[e]) enqueue_frame_for_exact(a, clip, elem_frame(L.riv_idx[e]));
                int slot = -1;
                if (lane == 0) { slot = claim_lite_slot(a, r, clip); if (slot == -2) flag_row_for_exact(a, r, clip); }
                slot = __shfl(slot, 0);
                if (slot >= 0) {
                    lite_slot = slot;
                    const LiteRecord q = carve_record(a.records + (size_t)slot * a.record_bytes, a.peak_cap);
                    if (lane == 0) *q.h = LiteHeader{1, n_peak, n_near, n_rival, n_unl, 0, 0, 0};
                    for (int k = lane; k < n_near; k += 64) { q.amb_idx[k] = L.amb_idx[k]; q.amb_exact[k] = L.amb_exact[k]; }
                    for (int e = lane; e < n_rival; e += 64) {
                        q.riv_owner[e] = L.riv_owner[e]; q.riv_ref[e] = L.riv_ref[e]; q.riv_idx[e] = L.riv_idx[e]; q.riv_exact[e] = L.riv_exact[e];
                    }
                    const int np0 = n_peak < a.peak_cap ? n_peak : a.peak_cap;
                    for (int p = lane; p < np0; p += 64) { q.pval[p] = L.pval[p]; q.pidx[p] = L.pidx[p]; }
                }
            } else {
                // level-2 values in place of the level-1 ones (all of the row's in one pipelined list), then the verdicts again
                short* items = L.unl_list;                     // (free here: the record carries no unlisted-rival map)
                int n2 = 0;
                for (int k0 = 0; k0 < n_near; k0 += 64) {
                    const bool f = k0 + lane < n_near && amb_n2[k0 + lane];
                    int next;
                    const int slot = ballot_slot(f, n2, lane, &next);
                    if (f && slot < kRivalCap) items[slot] = (short)(k0 + lane);
                    n2 = next;
                }
                for (int e0 = 0; e0 < n_rival; e0 += 64) {
                    const bool f = e0 + lane < n_rival && riv_n2[e0 + lane];
                    int next;
                    const int slot = ballot_slot(f, n2, lane, &next);
                    if (f && slot < kRivalCap) items[slot] = (short)(kAmbCap + e0 + lane);
                    n2 = next;
                }
                wave_sync();
                if (n2 > kRivalCap) { *missing = true; return; }       // (more than the list holds: the general path)
                double worst = 0.0;
                const bool have = level2_similarity_list(
                    ls->u64 + (int64_t)clip * ls->u64_clip_stride, ls->u64_gen + (int64_t)clip * ls->gen_clip_stride, ls->gen, self_frame,
                    ls->FS, lane, n2,
                    [&](int it) -> int64_t { const int c = items[it]; return elem_frame(c < kAmbCap ? L.amb_idx[c] : L.riv_idx[c - kAmbCap]); },
                    [&](int it, double e2) {
                        const int c = items[it];
                        double* at = c < kAmbCap ? &L.amb_exact[c] : &L.riv_exact[c - kAmbCap];
                        worst = fmax(worst, fabs(e2 - *at));
                        wave_sync();
                        if (lane == 0) *at = e2;
                    });
                wave_sync();
                if (!have) { *missing = true; return; }
                if (lane == 0 && a.stats) {
                    stat_add(a.stats, 6, (unsigned)n2);
                    if (worst > 0.0) stat_max(a.stats, 8, (unsigned)fmin(worst * 1e12, 4.0e9));
                }
                verdicts();
                bool differs = false;
                for (int k = lane; k < n_near; k += 64)
                    differs = differs || (((!L.amb_lose[k] && L.amb_exact[k] >= a.min_value64) ? 1 : 0) != L.amb_ok[k]);
                if (!__any(differs)) { *unchanged = true; return; }         // the first pass's list stands (up to its cut band)
                if (lane == 0 && a.stats) stat_add(a.stats, 7, 1u);
                for (int p = lane; p < ls->rec_np; p += 64) { L.pval[p] = ls->rec_pval[p]; L.pidx[p] = ls->rec_pidx[p]; }
                wave_sync();
            }
        }
        int changed = 0;
        for (int k0 = 0; k0 < n_near; k0 += 64) {
            const int k = k0 + lane;
            bool win = false;
            if (k < n_near) {
                const double ek = L.amb_exact[k];
                win = !L.amb_lose[k] && ek >= a.min_value64;
                if constexpr (!LEVEL2) changed += (win != (L.amb_ok[k] != 0));
            }
            int next;
            const int slot = ballot_slot(win, n_peak, lane, &next);
            if (win && slot < a.peak_cap) { L.pval[slot] = (float)L.amb_exact[k]; L.pidx[slot] = L.amb_idx[k]; }
            n_peak = next;
        }
        if (a.stats && !LEVEL2) {
            if (lane == 0) { stat_add(a.stats, 0, 1u); stat_add(a.stats, 1, (unsigned)n_near); }
            if (changed) stat_add(a.stats, 2, (unsigned)changed);
        }
        wave_sync();
    }

    // rank by counting: value descending, higher index first on ties (np.argsort(...)[::-1])
    int np_ = n_peak;
    if (np_ > a.peak_cap) np_ = a.peak_cap;
    const int kept = np_ < a.number ? np_ : a.number;
    for (int k = np_ + lane; k < ((np_ + 3) & ~3); k += 64) L.pval[k] = -INFINITY;       // pad to a float4 boundary
    wave_sync();
    int* out = a.idx + r * (int64_t)a.idx_pitch;
    const float4* pv4 = reinterpret_cast<const float4*>(L.pval);
    const bool cut_check = dlt > 0.0f && np_ > a.number;
    int* prank = reinterpret_cast<int*>(L.buf);               // the window maxima are no longer needed (cap <= 640 ints fit)
    // Two candidates per lane and pass over the list; equal VALUES are only counted here -- every candidate meets itself
    // in the list, so a tie branch inside the loop was taken by some lane in nearly every iteration (26 000 of a row's
    // 125 000 cycles). A candidate with a real tie (count > 1: rare) settles it by index afterwards.
    for (int pb = 0; pb < np_; pb += 128) {
        const int pa = pb + lane, pc = pb + 64 + lane;
        const bool has_a = pa < np_, has_c = pc < np_;
        const float va = has_a ? L.pval[pa] : INFINITY, vc = has_c ? L.pval[pc] : INFINITY;
        int gt_a = 0, eq_a = 0, gt_c = 0, eq_c = 0;
#pragma unroll 4
        for (int q4 = 0; 4 * q4 < np_; ++q4) {
            const float4 u = pv4[q4];
            gt_a += (u.x > va) + (u.y > va) + (u.z > va) + (u.w > va);
            eq_a += (u.x == va) + (u.y == va) + (u.z == va) + (u.w == va);
            gt_c += (u.x > vc) + (u.y > vc) + (u.z > vc) + (u.w > vc);
            eq_c += (u.x == vc) + (u.y == vc) + (u.z == vc) + (u.w == vc);
        }
        auto settle = [&](bool has, int p, float v, int gt, int eq) {
            if (!has) return;
            const int i = L.pidx[p];
            int rank = gt;
            if (eq > 1)                                       // higher index first on ties
                for (int q = 0; q < np_; ++q) rank += (L.pval[q] == v && L.pidx[q] > i);
            if (rank < a.number) out[rank] = out_index(i);
            if (cut_check) prank[p] = rank;
        };
        settle(has_a, pa, va, gt_a, eq_a);
        settle(has_c, pc, vc, gt_c, eq_c);
    }
    if (cut_check) {
        // Top-`number` cut with more candidates than slots (see peaks.hip): candidates within delta of the boundary
        // are re-ranked by float64 similarity.
        wave_sync();
        for (int p = lane; p < np_; p += 64) {
            if (prank[p] == a.number - 1) L.cutv[0] = L.pval[p];
            if (prank[p] == a.number) L.cutv[1] = L.pval[p];
        }
        wave_sync();
        const float c_in = L.cutv[0], c_out = L.cutv[1];
        if (c_in - c_out <= dlt) {
            const float lo = c_out - dlt, hi = c_in + dlt;
            int n_band = 0, n_above = 0;
            for (int p0 = 0; p0 < np_; p0 += 64) {
                const int p = p0 + lane;
                const float v = p < np_ ? L.pval[p] : -INFINITY;
                const bool above = p < np_ && v > hi, band = p < np_ && !above && v >= lo;
                n_above += __popcll(__ballot(above));
                int next;
                const int slot = ballot_slot(band, n_band, lane, &next);
                if (band && slot < kAmbCap) { L.amb_idx[slot] = L.pidx[p]; L.amb_ok[slot] = prank[p] < a.number; }
                n_band = next;
            }
            wave_sync();
            if (n_band <= kAmbCap) {
                const int len4 = a.unit_pitch >> 2;
                const float* self_row = a.unit + self_frame * (int64_t)a.unit_pitch;
                exact_similarity_list(self_row, len4, lane, n_band, [&](int it) { return elem_row(L.amb_idx[it]); },
                                      [&](int it, double e) { if (lane == 0) L.amb_exact[it] = e; });
                wave_sync();
                const bool settled = rank_cut_band<LEVEL2>(a, L, lane, n_band, n_above, out, out_index, a.stats && !LEVEL2);
                if (!settled) {
                    // the cut is closer than delta2: the members that may belong on the other side (amb_lose marks them)
                    if constexpr (!LEVEL2) {
                        if (lane == 0) enqueue_frame_for_exact(a, clip, self_frame);
                        for (int k = lane; k < n_band; k += 64) if (L.amb_lose[k]) enqueue_frame_for_exact(a, clip, elem_frame(L.amb_idx[k]));
                        if (lite_slot >= 0) {
                            const LiteRecord q = carve_record(a.records + (size_t)lite_slot * a.record_bytes, a.peak_cap);
                            if (lane == 0) { q.h->n_band = n_band; q.h->n_above = n_above; q.h->np = np_; }
                            for (int k = lane; k < n_band; k += 64) { q.band_idx[k] = L.amb_idx[k]; q.band_exact[k] = L.amb_exact[k]; }
                        } else {
                            int slot = -1;
                            if (lane == 0) { slot = claim_lite_slot(a, r, clip); if (slot == -2) flag_row_for_exact(a, r, clip); }
                            slot = __shfl(slot, 0);
                            if (slot >= 0) {
                                const LiteRecord q = carve_record(a.records + (size_t)slot * a.record_bytes, a.peak_cap);
                                if (lane == 0) *q.h = LiteHeader{2, 0, 0, 0, 0, n_band, n_above, np_};
                                for (int k = lane; k < n_band; k += 64) { q.band_idx[k] = L.amb_idx[k]; q.band_exact[k] = L.amb_exact[k]; }
                            }
                        }
                    } else {
                        finish_band_level2(a, L, lane, n_band, n_above, out, out_index, elem_frame, *ls, clip, self_frame, missing);
                        if (*missing) return;
                    }
                }
            } else if (lane == 0) {
                if (a.stats && !LEVEL2) stat_add(a.stats, 3, 1u);
                flag_row_for_exact(a, r, clip);                // a flat cut: the general second level
            }
        }
    }
    for (int k = kept + lane; k < a.number; k += 64) out[k] = -1;
    if (lane == 0) a.count[r] = kept;
}

// RD = d & 3: the two window reads at run-time offsets (-d and d-w+1) then have compile-time float4 remainders
// SEG (round 4, mode 0 only): the candidates come from the row's SEGMENT RECORDS (PeakArgs::seg) instead of a sweep over its
// n elements. A strict maximum of a window of +-d >= 31 elements is the maximum of its own aligned 32-element segment (the
// whole segment lies inside its window), so a row of n elements has n / 32 possible candidates, and a candidate's window
// maximum is made of its segment's second value, the maxima of the segments that lie whole inside the window, and at most
// two segments cut by the window's edges -- whose maximum counts as it is when its position lies inside the window, and
// whose raw elements are looked at only when the position lies outside AND the second value could still matter. The
// tolerance band of the near-tie refinement widens "maximum" to "within delta of the maximum": a segment whose two largest
// values are within delta of each other has its 32 elements read. Everything behind the candidates (decide(), rivals,
// float64 verdicts, ranking, second level) is the sweep kernel's own code: same decisions element for element.
template <int RD, bool SEG>
// One wavefront per WORKGROUP: rows take 30 .. 140 us (the refinement of near-ties varies), and a workgroup's LDS and
// registers are only handed back when its last wave is done (s_memrealtime spans of every row: tools/peak_stamps.py).
#ifdef REPET_PEAK_WAVES      // experiment (round 6): at most that many waves of this kernel per SIMD -- the rest of the SIMD's registers and the
                             // CU's LDS are left to the column sort that runs beside it
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(REPET_PEAK_WAVES, REPET_PEAK_WAVES))) void local_maxima_wave_kernel(PeakArgs a, int64_t n_rows, int lds_per_wave) {
#else
__global__ __launch_bounds__(64) void local_maxima_wave_kernel(PeakArgs a, int64_t n_rows, int lds_per_wave) {
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char wave_smem[];
    const int lane = threadIdx.x & 63, wave = 0;
    const int64_t r = (int64_t)blockIdx.x;                   // row within this launch
    if (r >= n_rows) return;
    const WaveLds L = carve(wave_smem + (size_t)wave * lds_per_wave, a.peak_cap, a.groups);
    const int n = a.n, d = a.d;
    a.M += blockIdx.y * a.m_stride;
    a.idx += blockIdx.y * a.idx_stride;
    a.count += blockIdx.y * a.cnt_stride;
    if (a.unit) a.unit += blockIdx.y * a.unit_stride;
    if (a.unit_norm) a.unit_norm += blockIdx.y * (a.unit_stride / a.unit_pitch);
    const int64_t j = a.row0 + r;                            // absolute row (mode 1: current frame)
    float dlt = a.delta;                                     // 0: no refinement

    // element i of the row for an index known to lie in [0, n), without a branch: loads of a batch must not each sit behind
    // their own control-flow join (the value is needed AT the join, so every load was waited for before the next was issued)
    auto fetch_inside = [&](int i) -> float {
        int l = (int)(j - i) % n;
        l += l < 0 ? n : 0;
        const int64_t at = a.mode == 0 ? j * a.pitch + i : (a.mode == 2 ? j - a.shift : j - l - a.shift) * a.pitch + l;    // mode 2: the look-back band, a contiguous row
        return a.M[at];
    };
    const bool vec_ok = (a.mode == 0) && ((a.pitch & 3) == 0);
    const float* src = a.M + j * a.pitch;
    int w = 1;
    while (2 * w <= d) w *= 2;                               // w = 2^floor(log2 d) (1 when d <= 1), at most 32 here
    const int halo = a.dl;                                   // round_up(d, 4), both sides
    const int step = kChunkElems - 2 * halo;                 // elements tested per chunk
    const int n_chunks = (n + step - 1) / step;
    const float4 ninf = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);

    // M[4g+off .. 4g+off+3] of the chunk for an offset off = 4 q + RR with a compile-time remainder RR
    auto read4 = [&](int gq, auto rr_tag) -> float4 {
        constexpr int RR = decltype(rr_tag)::value;
        const float4 x = L.buf[phys4(gq)];
        if constexpr (RR == 0) return x;
        const float4 y = L.buf[phys4(gq + 1)];
        if constexpr (RR == 1) return make_float4(x.y, x.z, x.w, y.x);
        if constexpr (RR == 2) return make_float4(x.z, x.w, y.x, y.y);
        return make_float4(x.w, y.x, y.y, y.z);
    };
    constexpr int RR_LEFT = (4 - RD) & 3;                    // (-d) & 3
    constexpr int RR_RIGHT = (RD + 1) & 3;                   // (d - w + 1) & 3, w a multiple of 4
    const int q_left = (-d) >> 2, q_wl = -(w >> 2), q_right = (d - w + 1) >> 2;

    // Full decision for the elements the sweep could not rule out (the arithmetic of peaks.hip on value v and the larger
    // of its two window maxima): survivors to pval / pidx, decisions inside the rounding band to the near-tie list.
    int n_peak = 0, n_amb = 0, n_cand = 0;
    auto decide = [&]() {
        wave_sync();
        for (int k0 = 0; k0 < n_cand; k0 += 64) {
            const int k = k0 + lane;
            const bool have = k < n_cand;
            const int i = have ? L.cand_i[k] : 0;
            const float v = have ? L.cand_v[k] : 0.f, mx = have ? L.cand_m[k] : 0.f;
            bool ok = have && (v >= a.min_value) && (v > mx);
            bool near = false;
            const bool was_ok = ok;
            if (dlt > 0.0f) {
                const bool sure_yes = (v >= a.min_value + dlt) && (v > mx + dlt);
                const bool sure_no = (v < a.min_value - dlt) || (v < mx - dlt);
                near = have && !sure_yes && !sure_no;
                ok = ok && !near;
            }
            if (__any(near)) {
                int next;
                const int slot = ballot_slot(near, n_amb, lane, &next);
                if (near && slot < kAmbCap) { L.amb_idx[slot] = i; L.amb_val[slot] = v; L.amb_ok[slot] = was_ok; }
                n_amb = next;
            }
            if (__any(ok)) {
                int next;
                const int slot = ballot_slot(ok, n_peak, lane, &next);
                if (ok && slot < a.peak_cap) { L.pval[slot] = v; L.pidx[slot] = i; }
                n_peak = next;
            }
        }
        n_cand = 0;
        wave_sync();
    };

    int n_riv = 0, n_unl = 0;
    WSTAMP_DECL
    // ---- SEG: the row's records, staged once (they survive a redo: nothing below touches the buffer before the row ends)
    const int nseg = (n + kSegWidth - 1) / kSegWidth;
    float* r1 = reinterpret_cast<float*>(L.buf);
    float* r2 = r1 + a.seg_pitch;
    int* ra = reinterpret_cast<int*>(r2 + a.seg_pitch);
    if constexpr (SEG) {
        const float* rec = a.seg + j * 3 * (int64_t)a.seg_pitch;
        for (int s = lane; s < nseg; s += 64) {
            r1[s] = rec[s];
            r2[s] = rec[a.seg_pitch + s];
            ra[s] = reinterpret_cast<const int*>(rec)[2 * a.seg_pitch + s];
        }
        wave_sync();
        WSTAMP(0)
    }
    // One candidate per lane: element i of segment s, value v, `own` = the largest of the other elements of its segment.
    // Wave-uniform call sites only (ballots and shuffles inside).
    // The cut segments that have to be read are far round trips (S does not fit the caches): the candidates that are their
    // segment's maximum -- all but a handful -- wait on a list (segment | bits << 28, the window maximum so far) and have
    // them read side by side behind the last segment (resolve_pending); only the extra candidates of a near-tied segment
    // read theirs on the spot.
    int* pend_s = ra + a.seg_pitch;
    float* pend_m = reinterpret_cast<float*>(pend_s + a.seg_pitch);
    int n_pend = 0;
    auto process = [&](bool have, int s, int i, float v, float own, auto deferred_tag) {
        constexpr bool DEFER = decltype(deferred_tag)::value;
        const bool valid = have && (v < INFINITY) && !(v < a.min_value - dlt);
        const float lim = v - dlt;                           // a contribution below this decides nothing
        const int lo = max(i - d, 0), hi = min(i + d, n - 1);
        float mx = own;
        int need = 0;                                        // bit 0 / 1: the cut segment on the left / right must be read
#pragma unroll
        for (int dt = -2; dt <= 2; ++dt) {
            if (dt == 0) continue;
            const int t = s + dt;
            const int tc = min(max(t, 0), nseg - 1);
            const int b0 = tc * kSegWidth, b1 = min(b0 + kSegWidth - 1, n - 1);
            const bool overlap = valid && t >= 0 && t < nseg && b1 >= lo && b0 <= hi;
            const float m = r1[tc], m2t = r2[tc];
            const int at = b0 + ra[tc];
            const bool counts = (b0 >= lo && b1 <= hi) || (at >= lo && at <= hi);     // whole inside, or its maximum is
            if (overlap && counts) mx = fmaxf(mx, m);
            if (overlap && !counts && !(m2t < lim)) need |= dt < 0 ? 1 : 2;
        }
        bool open = valid && !(v < mx - dlt);
        if constexpr (DEFER) {
            const bool later = open && need != 0;
            if (__any(later)) {
                int next;
                const int slot = ballot_slot(later, n_pend, lane, &next);
                if (later) { pend_s[slot] = s | (need << 28); pend_m[slot] = mx; }
                n_pend = next;
            }
            open = open && need == 0;
            need = 0;
        }
        const bool waiting = DEFER && valid && !open;        // (on the list, or a safe "no")
        // the cut segments that must be read: four candidates at a time, sixteen lanes each (eight float4 per side)
        unsigned long long pend = __ballot(open && need != 0);
        while (pend) {                                       // (wave-uniform)
            int from[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                from[g] = pend ? (int)__builtin_ctzll(pend) : -1;
                if (pend) pend &= pend - 1;
            }
            const int g = lane >> 4, side = (lane >> 3) & 1, k = lane & 7;
            const int mine = g == 0 ? from[0] : (g == 1 ? from[1] : (g == 2 ? from[2] : from[3]));
            const int ci = __shfl(i, mine < 0 ? 0 : mine), cneed = __shfl(need, mine < 0 ? 0 : mine);
            const int clo = max(ci - d, 0), chi = min(ci + d, n - 1);
            const int t = (side ? chi : clo) / kSegWidth;
            const int e0 = kSegWidth * t + 4 * k;
            const int r_lo = side ? kSegWidth * t : clo, r_hi = side ? chi : min(kSegWidth * t + kSegWidth - 1, n - 1);
            const bool act = mine >= 0 && (cneed & (side ? 2 : 1));
            const float4 x = *reinterpret_cast<const float4*>(src + (act ? e0 : 0));
            const float xs[4] = {x.x, x.y, x.z, x.w};
            float mm = -INFINITY;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (act && e0 + e >= r_lo && e0 + e <= r_hi) mm = fmaxf(mm, nan_to_inf(xs[e]));
            // the largest of the sixteen lanes, in all of them
            auto dpp_max = [&](float q, auto ctrl) {
                const int o = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, q), __builtin_bit_cast(int, q), decltype(ctrl)::value, 0xf, 0xf, false);
                return fmaxf(q, __builtin_bit_cast(float, o));
            };
            mm = dpp_max(mm, std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]
            mm = dpp_max(mm, std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]
            mm = dpp_max(mm, std::integral_constant<int, 0x141>{});      // row_half_mirror
            mm = dpp_max(mm, std::integral_constant<int, 0x140>{});      // row_mirror
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float res = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mm), 16 * q));
                if (lane == from[q]) mx = fmaxf(mx, res);
            }
        }
        open = valid && !waiting && !(v < mx - dlt);
        if (__any(open)) {
            if (n_cand > kCandCap - 64) decide();
            int next;
            const int slot = ballot_slot(open, n_cand, lane, &next);
            if (open) { L.cand_i[slot] = i; L.cand_v[slot] = v; L.cand_m[slot] = mx; }
            n_cand = next;
        }
    };
    // Eight lanes per waiting candidate (four per side, two float4 each), eight candidates per instruction, four such groups
    // with their loads in flight together: 32 candidates per round trip.
    auto resolve_pending = [&]() {
        wave_sync();
        const int grp = lane >> 3, side = (lane >> 2) & 1, kk = lane & 3;
        for (int b0 = 0; b0 < n_pend; b0 += 32) {
            float4 x0[4], x1[4];
            int cs[4], r_lo[4], r_hi[4], e0[4];
            bool act[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = b0 + 8 * u + grp;
                const int code = pend_s[c < n_pend ? c : 0];
                cs[u] = code & 0x0fffffff;
                const int ci = kSegWidth * cs[u] + ra[cs[u]];
                const int clo = max(ci - d, 0), chi = min(ci + d, n - 1);
                const int t = (side ? chi : clo) / kSegWidth;
                e0[u] = kSegWidth * t + 4 * kk;
                r_lo[u] = side ? kSegWidth * t : clo;
                r_hi[u] = side ? chi : min(kSegWidth * t + kSegWidth - 1, n - 1);
                act[u] = c < n_pend && ((code >> 28) & (side ? 2 : 1));
                x0[u] = *reinterpret_cast<const float4*>(src + (act[u] ? e0[u] : 0));
                x1[u] = *reinterpret_cast<const float4*>(src + (act[u] ? e0[u] + 16 : 0));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float xs[8] = {x0[u].x, x0[u].y, x0[u].z, x0[u].w, x1[u].x, x1[u].y, x1[u].z, x1[u].w};
                float mm = -INFINITY;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int at = e0[u] + (e < 4 ? e : e + 12);
                    if (act[u] && at >= r_lo[u] && at <= r_hi[u]) mm = fmaxf(mm, nan_to_inf(xs[e]));
                }
                auto dpp_max = [&](float q, auto ctrl) {
                    const int o = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, q), __builtin_bit_cast(int, q), decltype(ctrl)::value, 0xf, 0xf, false);
                    return fmaxf(q, __builtin_bit_cast(float, o));
                };
                mm = dpp_max(mm, std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]
                mm = dpp_max(mm, std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]
                mm = dpp_max(mm, std::integral_constant<int, 0x141>{});      // row_half_mirror: both sides
                const int c = b0 + 8 * u + grp;
                const bool mine = (lane & 7) == 0 && c < n_pend;
                const float v = r1[cs[u]];
                const float mx = fmaxf(pend_m[mine ? c : 0], mm);
                const bool open = mine && !(v < mx - dlt);
                if (__any(open)) {
                    if (n_cand > kCandCap - 64) decide();
                    int next;
                    const int slot = ballot_slot(open, n_cand, lane, &next);
                    if (open) { L.cand_i[slot] = kSegWidth * cs[u] + ra[cs[u]]; L.cand_v[slot] = v; L.cand_m[slot] = mx; }
                    n_cand = next;
                }
            }
        }
        n_pend = 0;
    };
    (void)process; (void)resolve_pending;
    for (;;) {
        n_peak = 0; n_amb = 0; n_cand = 0;
        if constexpr (SEG) {
            for (int s0 = 0; s0 < nseg; s0 += 64) {
                const bool have = s0 + lane < nseg;
                const int s = have ? s0 + lane : nseg - 1;
                const float v = r1[s], own = r2[s];
                process(have, s, kSegWidth * s + ra[s], v, own, std::true_type{});
                // a segment whose two largest values are within delta of each other may hold more than one open element
                unsigned long long multi = __ballot(have && dlt > 0.0f && (v < INFINITY) && !(v < a.min_value - dlt) && (v - own <= dlt));
                while (multi) {                              // (wave-uniform)
                    const int ss = s0 + (int)__builtin_ctzll(multi);
                    multi &= multi - 1;
                    const float top = r1[ss];
                    const int ai = kSegWidth * ss + ra[ss];
                    const int e = kSegWidth * ss + (lane & (kSegWidth - 1));
                    const bool in = lane < kSegWidth && e < n;
                    const float val = nan_to_inf(src[in ? e : ai]);
                    process(in && e != ai && val >= top - dlt, ss, e, val, top, std::false_type{});
                }
            }
            resolve_pending();
            WSTAMP(2)
        } else {
        // Chunk c + 1 is fetched (into registers) while chunk c is worked on -- the row walk is a chain of dependent steps
        // per wave, and a global load at its head would otherwise be waited for nine times per row. Only chunks that
        // lie wholly inside the row (plain 16-byte loads, no boundary cases: few registers) are fetched ahead.
        auto interior = [&](int c) { return vec_ok && c * step - halo >= 0 && c * step - halo + kChunkElems <= n; };
        float4 ahead[4];
        bool have_ahead = false;
        for (int c = 0; c < n_chunks; ++c) {
            const int s0 = c * step - halo;                  // first element of the chunk (a multiple of 4; may be negative)
            const int t_lo = c * step, t_hi = (t_lo + step < n) ? t_lo + step : n;     // elements tested: [t_lo, t_hi)
            wave_sync();                                     // the previous chunk's reads of the buffer are done
            // coalesced 16-byte loads -> padded LDS
            if (have_ahead) {                                // wave-uniform
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = ahead[q];
                    L.buf[phys4(64 * q + lane)] = make_float4(nan_to_inf(v.x), nan_to_inf(v.y), nan_to_inf(v.z), nan_to_inf(v.w));
                }
            } else if (interior(c)) {                        // (a chunk inside the row that was not fetched ahead)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(src + s0 + 4 * (64 * q + lane));
                    L.buf[phys4(64 * q + lane)] = make_float4(nan_to_inf(v.x), nan_to_inf(v.y), nan_to_inf(v.z), nan_to_inf(v.w));
                }
            } else if (a.mode == 2 && n_chunks == 1 && (a.pitch & 3) == 0) {
                // simonline on the look-back band (round 4): element i of the row is lag l = (j - i) mod n of band row j -- ONE
                // contiguous row read backwards from a rotation point. The row comes in with 16-byte loads in lag order and
                // every value is put at its element's place in the buffer (the buffer is filled with -inf first: halo and
                // tail); element by element the chunk cost sixteen integer divisions per lane, 40 % of such a row's time.
                const float* row = a.M + (j - a.shift) * a.pitch;
                const int jm = (int)(j % n);                                   // element jm is lag 0
                float4 got[4];
                const int n4 = (n + 3) >> 2;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    got[q] = (64 * q + lane < n4) ? *reinterpret_cast<const float4*>(row + 4 * (64 * q + lane)) : ninf;
#pragma unroll
                for (int q = 0; q < 4; ++q) L.buf[phys4(64 * q + lane)] = ninf;
                wave_sync();
                float* flat = reinterpret_cast<float*>(L.buf);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float vals[4] = {got[q].x, got[q].y, got[q].z, got[q].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int l = 4 * (64 * q + lane) + k;
                        int i = jm - l;
                        i += i < 0 ? n : 0;                                   // (l < n: one wrap at most)
                        const int p_ = i - s0;                                 // place in the chunk
                        if (l < n) flat[4 * phys4(p_ >> 2) + (p_ & 3)] = nan_to_inf(vals[k]);
                    }
                }
            } else {
                // the first and the last chunk of a row, circular-buffer order (simonline: element i of the row is a walk
                // down a diagonal of the banded matrix) or an unaligned pitch: sixteen single loads per lane, ALL issued
                // before the first is used -- element by element behind their bounds checks they were up to sixteen memory
                // round trips per chunk (half of a short row's time, two chunks in nine of a long one's)
                float e[16];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = s0 + 4 * (64 * q + lane) + k;
                        e[4 * q + k] = fetch_inside((i >= 0 && i < n) ? i : 0);      // (one modulo per element: a lag recurrence
                    }                                                                // over the four of a group measured slower)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i0 = s0 + 4 * (64 * q + lane);
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = (i0 + k >= 0 && i0 + k < n) ? nan_to_inf(e[4 * q + k]) : -INFINITY;
                    L.buf[phys4(64 * q + lane)] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            have_ahead = c + 1 < n_chunks && interior(c + 1);
            if (have_ahead) {
                const float* nx = src + (c + 1) * step - halo + 4 * lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) ahead[q] = *reinterpret_cast<const float4*>(nx + 256 * q);
            }
            wave_sync();
            WSTAMP(0)
            // lane-contiguous: this lane owns chunk elements 16 lane .. 16 lane + 15
            float raw[16], m[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 v = L.buf[5 * lane + k];        // phys4(4 lane + k)
                raw[4 * k] = v.x; raw[4 * k + 1] = v.y; raw[4 * k + 2] = v.z; raw[4 * k + 3] = v.w;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) m[e] = raw[e];
            // window maxima by doubling, partners beyond the lane's 16 elements from the next lane's registers
#pragma unroll
            for (int h = 1; h <= 8; h *= 2) {
                if (h < w) {                                 // wave-uniform
                    float t[8];
#pragma unroll
                    for (int k = 0; k < h; ++k) t[k] = from_next_lane(m[k], -INFINITY);
#pragma unroll
                    for (int e = 0; e + h < 16; ++e) m[e] = fmaxf(m[e], m[e + h]);
#pragma unroll
                    for (int e = 16 - h; e < 16; ++e) m[e] = fmaxf(m[e], t[e + h - 16]);
                }
            }
            if (w > 16) {
#pragma unroll
                for (int e = 0; e < 16; ++e) m[e] = fmaxf(m[e], from_next_lane(m[e], -INFINITY));
            }
            wave_sync();                                     // every lane has read its raw values
#pragma unroll
            for (int k = 0; k < 4; ++k) L.buf[5 * lane + k] = make_float4(m[4 * k], m[4 * k + 1], m[4 * k + 2], m[4 * k + 3]);
            wave_sync();
            WSTAMP(1)
            // The sweep: almost every element is below the larger of its window maxima by more than delta -- a safe "no".
            // Only the others (peaks and near-ties: one element in forty) are written down for the full decision.
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int gl = 4 * lane + k;                 // chunk-local group
                const int i0 = s0 + 4 * gl;
                const bool real = i0 >= t_lo && i0 < t_hi;   // t_lo, halo, s0 are multiples of 4: a group is all tested or not at all
                float4 mx = ninf;
                if (real && d > 0) {
                    const float4 left = max4(read4(gl + q_left, std::integral_constant<int, RR_LEFT>{}), read4(gl + q_wl, std::integral_constant<int, 0>{}));
                    const float4 right = max4(read4(gl, std::integral_constant<int, 1>{}), read4(gl + q_right, std::integral_constant<int, RR_RIGHT>{}));
                    mx = max4(left, right);
                }
                const float vals[4] = {raw[4 * k], raw[4 * k + 1], raw[4 * k + 2], raw[4 * k + 3]};
                const float mxs[4] = {mx.x, mx.y, mx.z, mx.w};
                bool open[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = vals[e];
                    const bool valid = real && i0 + e < t_hi && (v < INFINITY);
                    open[e] = valid && !((v < a.min_value - dlt) || (v < mxs[e] - dlt));           // not a safe "no"
                }
                // peaks lie more than d >= 4 apart: a group of four holds at most one, so one compaction serves the group;
                // a second open element in a group (two near-ties side by side) takes the per-element path
                const bool any_open = open[0] || open[1] || open[2] || open[3];
                if (__any(any_open)) {
                    const int first = open[0] ? 0 : (open[1] ? 1 : (open[2] ? 2 : 3));
                    if (n_cand > kCandCap - 64) decide();
                    int next;
                    const int slot = ballot_slot(any_open, n_cand, lane, &next);
                    if (any_open) {
                        L.cand_i[slot] = i0 + first;
                        L.cand_v[slot] = open[0] ? vals[0] : (open[1] ? vals[1] : (open[2] ? vals[2] : vals[3]));
                        L.cand_m[slot] = open[0] ? mxs[0] : (open[1] ? mxs[1] : (open[2] ? mxs[2] : mxs[3]));
                    }
                    n_cand = next;
                    const int n_open = (int)open[0] + (int)open[1] + (int)open[2] + (int)open[3];
                    if (__any(n_open > 1)) {
#pragma unroll
                        for (int e = 1; e < 4; ++e) {
                            const bool more = open[e] && e != first;
                            if (__any(more)) {
                                if (n_cand > kCandCap - 64) decide();
                                int nx;
                                const int sl = ballot_slot(more, n_cand, lane, &nx);
                                if (more) { L.cand_i[sl] = i0 + e; L.cand_v[sl] = vals[e]; L.cand_m[sl] = mxs[e]; }
                                n_cand = nx;
                            }
                        }
                    }
                }
            }
            WSTAMP(2)
        }
        }   // (!SEG)
        decide();
        WSTAMP(3)
        wave_sync();
        bool redo = false;
        n_riv = 0; n_unl = 0;
        if (dlt > 0.0f && n_amb > 0) {
            if (n_amb > kAmbCap) redo = true;                // a flat row: more near-ties than the list holds
            else {
                // Rivals of the near-tied elements: element k is a rival of near-tied i when it lies in i's window and
                // within delta below it (nothing in the window is more than delta above i, or i would have been a safe
                // "no"). The row is read again where it lies (it went through this CU's caches a moment ago).
                const int n_near = n_amb;
                for (int k = lane; k < n_near; k += 64) L.amb_lose[k] = 0;
                if constexpr (SEG) {
                    // From the records: a segment whose largest value is below the near-tied element's band holds no rival;
                    // one whose SECOND value is below it holds at most its maximum; only the others -- and the element's
                    // own segment, unless the element is its maximum and the second value is below the band -- are read.
                    int* req = L.cand_i;                     // (element, segment) pairs to read: the sweep's buffer is free
                    int n_req = 0;
                    const int total = n_near * 5;
                    for (int e0 = 0; e0 < total && n_riv <= kRivalCap; e0 += 64) {
                        const int e = e0 + lane;
                        const bool have = e < total;
                        const int q = have ? (e * 52429) >> 18 : 0;                   // e / 5 (e < 2^14)
                        const int dt = e - 5 * q - 2;
                        const int i = L.amb_idx[q];
                        const float lim = L.amb_val[q] - dlt;
                        const int t = i / kSegWidth + dt;
                        const int tc = min(max(t, 0), nseg - 1);
                        const int b0 = tc * kSegWidth, b1 = min(b0 + kSegWidth - 1, n - 1);
                        const int lo = max(i - d, 0), hi = min(i + d, n - 1);
                        const bool overlap = have && t >= 0 && t < nseg && b1 >= lo && b0 <= hi;
                        const float m = r1[tc], m2 = r2[tc];
                        const int at = b0 + ra[tc];
                        const bool scan = overlap && (dt == 0 ? !(at == i && m2 < lim) : !(m2 < lim));
                        const bool single = overlap && dt != 0 && !scan && !(m < lim) && at >= lo && at <= hi;
                        if (__any(single)) {
                            int next;
                            const int entry = ballot_slot(single, n_riv, lane, &next);
                            if (single && entry < kRivalCap) { L.riv_owner[entry] = (short)q; L.riv_idx[entry] = at; }
                            n_riv = next;
                        }
                        if (__any(scan)) {
                            int next;
                            const int slot = ballot_slot(scan, n_req, lane, &next);
                            if (scan && slot < kCandCap) req[slot] = q | (tc << 8);
                            n_req = next;
                        }
                    }
                    wave_sync();
                    if (n_req > kCandCap) n_riv = kRivalCap + 1;                      // (more than the buffer holds: a flat row)
                    for (int rq = 0; rq < n_req && n_riv <= kRivalCap; rq += 2) {      // two requests per step, 32 lanes each
                        const int mine = rq + (lane >> 5);
                        const int code = req[mine < n_req ? mine : rq];
                        const int q = code & 255, t = code >> 8;
                        const int i = L.amb_idx[q];
                        const float lim = L.amb_val[q] - dlt;
                        const int lo = max(i - d, 0), hi = min(i + d, n - 1);
                        const int k = kSegWidth * t + (lane & 31);
                        const bool in = mine < n_req && k < n && k >= lo && k <= hi && k != i;
                        const float val = nan_to_inf(src[in ? k : i]);
                        const bool rival = in && val >= lim;
                        if (__any(rival)) {
                            int next;
                            const int entry = ballot_slot(rival, n_riv, lane, &next);
                            if (rival && entry < kRivalCap) { L.riv_owner[entry] = (short)q; L.riv_idx[entry] = k; }
                            n_riv = next;
                        }
                    }
                } else {
                // All (near-tied element, window position) pairs are walked lane-parallel, 256 at a time with their
                // loads in flight together: one near-tie after the other, each behind its own global round trip, was
                // 7 000 cycles per near-tie and the whole of the slow rows' time. Entries come out in the same order
                // (element, then position).
                const int win = 2 * d + 1, total = n_near * win;
                const float inv_win = 1.0f / (float)win;
                for (int e0 = 0; e0 < total && n_riv <= kRivalCap; e0 += 256) {
                    float val[4], lim[4];
                    int pos[4], own[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int e = e0 + 64 * b + lane;
                        const int ec = e < total ? e : 0;
                        const int s_ = (int)(((float)ec + 0.5f) * inv_win);         // ec / win: ec < 2^14, never near a boundary
                        const int i = L.amb_idx[s_];
                        const int k = i - d + (ec - s_ * win);
                        own[b] = s_; pos[b] = (e < total && k != i && k >= 0 && k < n) ? k : -1;
                        lim[b] = L.amb_val[s_] - dlt;
                        val[b] = fetch_inside(pos[b] >= 0 ? pos[b] : i);
                    }
#pragma unroll
                    for (int b = 0; b < 4; ++b) val[b] = nan_to_inf(val[b]);
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int k = pos[b];
                        const bool rival = k >= 0 && val[b] >= lim[b];
                        if (__any(rival)) {
                            int next;
                            const int entry = ballot_slot(rival, n_riv, lane, &next);
                            if (rival && entry < kRivalCap) { L.riv_owner[entry] = (short)own[b]; L.riv_idx[entry] = k; }
                            n_riv = next;
                        }
                    }
                }
                }   // (!SEG)
                wave_sync();
                // which rivals are near-tied elements themselves (they have a float64 value already), which get one of
                // their own: one pass over the entries, the near-tie list read as LDS broadcasts
                const int n_kept = n_riv < kRivalCap ? n_riv : kRivalCap;
                for (int e0 = 0; e0 < n_kept; e0 += 64) {
                    const int e = e0 + lane;
                    const int k = e < n_kept ? L.riv_idx[e] : -1;
                    int ref = -1;
                    for (int t = 0; t < n_near; ++t) if (L.amb_idx[t] == k) ref = t;
                    if (e < n_kept) L.riv_ref[e] = (short)ref;
                    int next_u;
                    const bool unlisted = e < n_kept && ref < 0;
                    const int us = ballot_slot(unlisted, n_unl, lane, &next_u);
                    if (unlisted) L.unl_list[us] = (short)e;
                    n_unl = next_u;
                }
                if (n_riv > kRivalCap) redo = true;
            }
        }
        WSTAMP(6)                                             // rivals of the near-tied elements
        if (!redo) break;
        if (a.stats && lane == 0) { stat_add(a.stats, 3, 1u); flag_row_for_exact(a, r, blockIdx.y); }   // redo the test with the plain
        dlt = 0.0f;                                                // fp32 decisions (the second level decides the row again)
    }
    wave_sync();

    WSTAMP(4)
    wave_finish_row<false>(a, L, lane, r, j, (int)blockIdx.y, dlt, n_peak, n_amb, n_riv, n_unl, nullptr, nullptr, nullptr);
    WSTAMP(5)
    WSTAMP_OUT
}

// The rows the first pass left records of (see wave_finish_row), one wavefront per row, taken from the list until it is
// empty (fixed grid). A row whose float64 unit rows are not all there (its band changed with the level-2 verdicts) goes
// to the general kernel (peaks_exact.hip), which runs next on the stream.
__global__ __launch_bounds__(64) void local_maxima_lite_kernel(PeakArgs a0, LiteSource ls) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lite_smem[];
    const int lane = threadIdx.x & 63;
    const WaveLds L = carve(lite_smem, a0.peak_cap);
    const unsigned int n_rows = a0.stats[12];
    // (rows dealt by position, not from a shared cursor: one word hands out about 88 slots per microsecond, and 2 048
    // wavefronts asking at once -- most of them only to learn that the list is empty -- took 25 of the kernel's 45 us)
    for (unsigned int slot = blockIdx.x; slot < n_rows; slot += gridDim.x) {
        PeakArgs a = a0;
        const int64_t r = a.lite_list[2 * slot];
        const int clip = a.lite_list[2 * slot + 1];
        a.M += clip * a.m_stride;
        a.idx += clip * a.idx_stride;
        a.count += clip * a.cnt_stride;
        a.unit += clip * a.unit_stride;
        const int64_t j = a.row0 + r;
        const LiteRecord q = carve_record(a.records + (size_t)slot * a.record_bytes, a.peak_cap);
        const LiteHeader h = *q.h;
        bool missing = false, unchanged = false;
        wave_sync();
        if (h.type == 1) {
            for (int k = lane; k < h.n_near; k += 64) { L.amb_idx[k] = q.amb_idx[k]; L.amb_exact[k] = q.amb_exact[k]; }
            for (int e = lane; e < h.n_rival; e += 64) {
                L.riv_owner[e] = q.riv_owner[e]; L.riv_ref[e] = q.riv_ref[e]; L.riv_idx[e] = q.riv_idx[e]; L.riv_exact[e] = q.riv_exact[e];
            }
            LiteSource lr = ls;
            lr.rec_pval = q.pval; lr.rec_pidx = q.pidx; lr.rec_np = h.n_peak < a.peak_cap ? h.n_peak : a.peak_cap;
            wave_sync();
            wave_finish_row<true>(a, L, lane, r, j, clip, a.delta, h.n_peak, h.n_near, h.n_rival, h.n_unl, &lr, &missing, &unchanged);
        }
        if (h.type == 2 || (unchanged && !missing && h.n_band > 0)) {
            // the recorded band of the cut: its close members get level-2 values, the band is ranked again
            wave_sync();
            for (int k = lane; k < h.n_band; k += 64) { L.amb_idx[k] = q.band_idx[k]; L.amb_exact[k] = q.band_exact[k]; L.amb_ok[k] = 0; }
            wave_sync();
            const int n = a.n;
            auto elem_frame = [&](int i) -> int64_t {
                if (a.mode == 0) return i;
                int l = (int)(j - i) % n;
                if (l < 0) l += n;
                return j - l - a.shift;
            };
            auto out_index = [&](int i) -> int { return a.mode == 0 ? i : (int)elem_frame(i); };
            const int64_t self_frame = j - a.shift;
            int* out = a.idx + r * (int64_t)a.idx_pitch;
            if (!rank_cut_band<true>(a, L, lane, h.n_band, h.n_above, out, out_index, false))
                finish_band_level2(a, L, lane, h.n_band, h.n_above, out, out_index, elem_frame, ls, clip, self_frame, &missing);
        }
        if (missing && lane == 0) { stat_add(a.stats, 14, 1u); flag_row_for_exact(a, r, clip); }
        wave_sync();
    }
}

#if defined(REPET_PEAK_STAMPS) || defined(REPET_PEAK_SPANS)
extern "C" int repet_debug_wave_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_stamps), sizeof(unsigned long long) * 64);
}
extern "C" int repet_debug_wave_phases(unsigned int* out, int rows) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_phase), sizeof(unsigned int) * 10 * rows);
}
extern "C" int repet_debug_wave_spans(unsigned long long* out, int rows) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_span), sizeof(unsigned long long) * 2 * rows);
}
#endif

static bool wave_shape(int n_cols, int d, int* cap_out);
static bool wave_kernel_off();

template <int RD, bool SEG = false>
static hipError_t launch_wave_rd(const PeakArgs& a, int64_t n_rows, int n_batch, int bytes, int per_wave, hipStream_t s) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&local_maxima_wave_kernel<RD, SEG>), bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((local_maxima_wave_kernel<RD, SEG>), dim3((unsigned)n_rows, (unsigned)n_batch), dim3(64), bytes, s, a, n_rows, per_wave);
    return hipGetLastError();
}

// ---- segment records of a matrix that did not come with them (the Gram kernel of the long clips writes them in its
// epilogue, gram_f16_big.hip): one wavefront per row, eight lanes per segment (one 128-byte line per segment and step).
// Merging two (largest, second, position) triples; the lower position wins between equal values.
struct SegTop { float m1, m2; int at; };
__device__ __forceinline__ SegTop seg_merge(SegTop x, SegTop y) {
    SegTop o;
    const bool take_y = (y.m1 > x.m1) || (y.m1 == x.m1 && y.at < x.at);
    o.m2 = fmaxf(fminf(x.m1, y.m1), fmaxf(x.m2, y.m2));
    o.m1 = take_y ? y.m1 : x.m1;
    o.at = take_y ? y.at : x.at;
    return o;
}
template <int CTRL>
__device__ __forceinline__ SegTop seg_dpp(SegTop x) {
    SegTop y;
    y.m1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x.m1), __builtin_bit_cast(int, x.m1), CTRL, 0xf, 0xf, false));
    y.m2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x.m2), __builtin_bit_cast(int, x.m2), CTRL, 0xf, 0xf, false));
    y.at = __builtin_amdgcn_update_dpp(x.at, x.at, CTRL, 0xf, 0xf, false);
    return seg_merge(x, y);
}
__global__ __launch_bounds__(256) void segment_maxima_kernel(const float* __restrict__ M, int64_t n_rows, int n, int64_t pitch,
                                                             float* __restrict__ seg, int seg_pitch) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* src = M + row * pitch;
    float* rec = seg + row * 3 * (int64_t)seg_pitch;
    const int nseg = (n + kSegWidth - 1) / kSegWidth;
    const bool vec_ok = (pitch & 3) == 0;
    for (int s0 = 0; s0 < nseg; s0 += 8) {
        const int s = s0 + (lane >> 3), k = lane & 7;
        const bool have = s < nseg;
        const int e0 = kSegWidth * s + 4 * k;
        float xs[4];
        if (vec_ok) {
            const float4 x = *reinterpret_cast<const float4*>(src + (have ? e0 : 0));     // (readable up to the pitch)
            xs[0] = x.x; xs[1] = x.y; xs[2] = x.z; xs[3] = x.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) xs[e] = src[(have && e0 + e < n) ? e0 + e : 0];
        }
        SegTop t{-INFINITY, -INFINITY, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float val = (have && e0 + e < n) ? nan_to_inf(xs[e]) : -INFINITY;
            t = seg_merge(t, SegTop{val, -INFINITY, 4 * k + e});
        }
        t = seg_dpp<0xB1>(t);        // quad_perm [1,0,3,2]
        t = seg_dpp<0x4E>(t);        // quad_perm [2,3,0,1]
        t = seg_dpp<0x141>(t);       // row_half_mirror: the other quad of the eight
        if (have && k == 0) {
            rec[s] = t.m1;
            rec[seg_pitch + s] = t.m2;
            reinterpret_cast<int*>(rec)[2 * seg_pitch + s] = t.at;
        }
    }
}

// norms[r] = float64 L2 norm of fp32 row r (PeakRefine::unit_norms): one wavefront per row, 16-byte loads
__global__ __launch_bounds__(256) void unit_row_norms_kernel(const float* __restrict__ rows, int64_t n_rows, int pitch, double* __restrict__ norms) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const float4* r4 = reinterpret_cast<const float4*>(rows + r * pitch);
    const int len4 = pitch >> 2;
    double acc = 0.0;
    for (int k0 = 0; k0 < len4; k0 += 320) {
        float4 p[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) p[u] = r4[min(k0 + 64 * u + lane, len4 - 1)];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const double live = (k0 + 64 * u + lane < len4) ? 1.0 : 0.0;
            const double p0 = p[u].x, p1 = p[u].y, p2 = p[u].z, p3 = p[u].w;
            acc += live * (p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3);
        }
    }
    acc = wave_sum_f64(acc);
    if (lane == 0) norms[r] = sqrt(acc);
}
hipError_t launch_unit_row_norms(const float* rows, int64_t n_rows, int32_t pitch, double* norms, hipStream_t s) {
    if (n_rows <= 0) return hipSuccess;
    if (!rows || !norms || (pitch & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(unit_row_norms_kernel, dim3((unsigned)ceil_div(n_rows, 4)), dim3(256), 0, s, rows, n_rows, pitch, norms);
    return hipGetLastError();
}

hipError_t launch_segment_maxima(const float* M, int64_t n_rows, int n_cols, int64_t pitch, float* seg, int seg_pitch, hipStream_t s) {
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(segment_maxima_kernel, dim3((unsigned)ceil_div(n_rows, 4)), dim3(256), 0, s, M, n_rows, n_cols, pitch, seg, seg_pitch);
    return hipGetLastError();
}
int segment_pitch(int n_cols) { return (int)round_up(ceil_div(n_cols, kSegWidth), 4); }
bool local_maxima_segments_apply(int n_cols, int d, int64_t pitch, int mode, int n_batch) {
    static const bool off = [] { const char* e = getenv("REPET_PEAK_SEGMENTS"); return e && e[0] == '0'; }();   // REPET_PEAK_SEGMENTS=0: the sweep
    if (d > n_cols) d = n_cols;
    int cap = 0;
    return !off && !wave_kernel_off() && mode == 0 && n_batch == 1 && (pitch & 3) == 0 && d >= kSegWidth - 1 && wave_shape(n_cols, d, &cap);
}

static bool wave_shape(int n_cols, int d, int* cap_out) {
    // windows of 4 .. 32 elements per doubling chain (4 <= d <= 63), lists that fit the wave's LDS share
    if (d < 4 || d > 63 || n_cols < 1) return false;
    const int cap = (int)round_up(n_cols / (d + 1) + 2, 4);
    if (cap > kMaxWaveCap) return false;
    *cap_out = cap;
    return true;
}
static bool wave_kernel_off() {
    static const bool off = [] { const char* e = getenv("REPET_PEAKS"); return e && e[0] == 'b'; }();   // REPET_PEAKS=block: the workgroup kernel
    return off;
}

bool local_maxima_wave_supported(int n_cols, int d, int* record_bytes) {
    int cap = 0;
    if (d > n_cols) d = n_cols;
    if (wave_kernel_off() || !wave_shape(n_cols, d, &cap)) return false;
    if (record_bytes) *record_bytes = (int)round_up((int64_t)lite_record_bytes(cap), 16);
    return true;
}

hipError_t launch_local_maxima_wave(const PeakArgs& a0, int64_t n_rows, int n_batch, hipStream_t s) {
    if (wave_kernel_off()) return hipErrorNotSupported;
    PeakArgs a = a0;
    int cap = 0;
    if (!wave_shape(a.n, a.d, &cap)) return hipErrorNotSupported;
    a.dl = (int)round_up(a.d, 4);
    a.peak_cap = cap;
    a.groups = kBufGroups;
    hipError_t e;
    if (a.seg && local_maxima_segments_apply(a.n, a.d, a.pitch, a.mode, n_batch)) {
        a.groups = std::max<int>(kBufGroups, (int)ceil_div(5 * (int64_t)a.seg_pitch * 4, 16));    // three planes + the waiting list
        const int seg_bytes = (int)round_up((int64_t)wave_lds_bytes(cap, a.groups), 16);
        e = launch_wave_rd<0, true>(a, n_rows, n_batch, seg_bytes, seg_bytes, s);
    } else {
        a.seg = nullptr;
        const int per_wave = (int)round_up((int64_t)wave_lds_bytes(cap), 16);
        const int bytes = per_wave;
        switch (a.d & 3) {
            case 0: e = launch_wave_rd<0>(a, n_rows, n_batch, bytes, per_wave, s); break;
            case 1: e = launch_wave_rd<1>(a, n_rows, n_batch, bytes, per_wave, s); break;
            case 2: e = launch_wave_rd<2>(a, n_rows, n_batch, bytes, per_wave, s); break;
            default: e = launch_wave_rd<3>(a, n_rows, n_batch, bytes, per_wave, s); break;
        }
    }
    return e;
}

// the rows on PeakArgs::lite_list (count on the device): fixed grid of one-wave workgroups
hipError_t launch_local_maxima_lite(const PeakArgs& a0, const ExactSource& src, hipStream_t s) {
    PeakArgs a = a0;
    int cap = 0;
    if (!a.lite_list || !wave_shape(a.n, a.d, &cap)) return hipSuccess;
    a.dl = (int)round_up(a.d, 4);
    a.peak_cap = cap;
    a.groups = kBufGroups;
    LiteSource ls{src.u64, src.u64_gen, src.u64_clip_stride, src.gen_clip_stride, src.FS, a.gen, nullptr, nullptr, 0};
    const int bytes = (int)round_up((int64_t)wave_lds_bytes(cap), 16);
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&local_maxima_lite_kernel), bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(local_maxima_lite_kernel, dim3(2048), dim3(64), bytes, s, a, ls);
    return hipGetLastError();
}

}  // namespace repet
