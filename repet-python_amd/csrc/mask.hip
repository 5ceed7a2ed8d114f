// K5 / K8 / K8b: repeating-model median + soft mask for gfx950 -- replaces _simmask, _mask and
// _adaptivemask (repet.py:1386-1545) and the per-frame median of the online variant (repet.py:872-887).
//
// Spectrograms are frame-major V[c][t][FS], so "the same 64 frequency bins of frame idx[k]" is one
// coalesced 256-byte read per wave. A wave owns 64 consecutive bins of one frame; every lane gathers
// its <= 128 values into registers (the index list is wave-uniform, so the row offsets sit in SGPRs),
// runs a pruned odd-even merge-sort network (median_networks.inc) and reads the two middle order
// statistics; np.median's even/odd rule and its NaN for an empty list are reproduced. Lists longer
// than 128 fall back to a 31-step bisection on the float bit patterns that re-reads the values.
// The soft mask (min(V,model)+eps)/(V+eps) (repet.py:1441-1448), the high-pass override
// mask[1..cutoff] = 1 (repet.py:185) and the multiplication into the STFT happen in the same pass.
#include "common.h"

#include <utility>

#include <algorithm>

#include <type_traits>

namespace repet {

#define REPET_NET MedianNet
#define REPET_T float
#define REPET_OP_MIN "v_min_i32"
#define REPET_OP_MAX "v_max_i32"
#define REPET_OP_MIN3 "v_min3_i32"
#define REPET_OP_MAX3 "v_max3_i32"
#include "median_networks.inc"
#undef REPET_NET
#undef REPET_T
#undef REPET_OP_MIN
#undef REPET_OP_MAX
#undef REPET_OP_MIN3
#undef REPET_OP_MAX3
// the same networks on two 16-bit rank codes per register (rank.hip)
#define REPET_NET MedianNetPk
#define REPET_T unsigned
#define REPET_OP_MIN "v_pk_min_u16"
#define REPET_OP_MAX "v_pk_max_u16"
#define REPET_OP_MIN3 "v_pk_minimum3_f16"
#define REPET_OP_MAX3 "v_pk_maximum3_f16"
#include "median_networks.inc"
#undef REPET_NET
#undef REPET_T
#undef REPET_OP_MIN
#undef REPET_OP_MAX
#undef REPET_OP_MIN3
#undef REPET_OP_MAX3

// Median of n gathered values with the N-wire network. `load(k)` must return the k-th value for
// k < n and, for n <= k < N, the pad of slot k: -1.0f for the first (N-n)/2 pad slots, +inf for the
// rest, which keeps the two middle order statistics of the n real values at wires N/2-1 and N/2.
// (The kernels fetch the pads from two constant rows behind each channel's spectrogram, selected
// with scalar arithmetic, so there is ONE branch-free gather whatever the list length.)
// a[w] = load(w) for every wire, issued in the order the network first reads the wires (MedianNet<N>::kLoadOrder):
// the first comparators start while the tail of the gather is still in flight. Indices are compile-time constants.
template <int N, class Load, size_t... Q>
__device__ __forceinline__ void gather_in_network_order(float (&a)[N], Load load, std::index_sequence<Q...>) {
    ((a[MedianNet<N>::kLoadOrder[Q]] = load((int)MedianNet<N>::kLoadOrder[Q])), ...);
}

// NANS: np.median's rule -- a NaN among the values makes the median NaN (the min / max instructions of the network drop a
// NaN operand). The values are magnitudes (>= 0, possibly +inf) and the pads -1 / +inf, so their plain sum is NaN exactly when
// one of them is: N - 1 additions. The period family asks for it (a NaN frame -- strict reference mode -- marks its position in
// every period, as in repet.py); the lists of `sim` never hold a NaN frame.
template <int N, bool NANS = false, class Load>
__device__ __forceinline__ float median_network(int n, Load load) {
    float a[N];
    gather_in_network_order<N>(a, load, std::make_index_sequence<N>{});
    float total = 0.f;
    if constexpr (NANS) {
#pragma unroll
        for (int k = 0; k < N; ++k) total += a[k];
    }
    MedianNet<N>::run(a);
    const float med = (n & 1) ? a[N / 2 - 1] : 0.5f * (a[N / 2 - 1] + a[N / 2]);
    return (NANS && total != total) ? total : med;
}

// Order statistics by bisection over the (non-negative) float bit patterns (lists longer than 128).
template <class Load>
__device__ __forceinline__ float median_bisect(int n, Load load) {
    const int k = (n - 1) >> 1;               // lower median rank
    unsigned lo = 0u, hi = 0x7f800000u;       // answer in [lo, hi]
    while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int c = 0;
        for (int q = 0; q < n; ++q) c += (__float_as_uint(load(q)) <= mid);
        if (c >= k + 1) hi = mid; else lo = mid + 1;
    }
    const float lower = __uint_as_float(lo);
    if (n & 1) return lower;
    int c_le = 0;
    float next = INFINITY;
    for (int q = 0; q < n; ++q) {
        const float v = load(q);
        c_le += (v <= lower);
        if (v > lower) next = fminf(next, v);
    }
    const float upper = (c_le >= k + 2) ? lower : next;
    return 0.5f * (lower + upper);
}

// NET = network size compiled into the kernel (0 = bisection). One instantiation per size keeps the
// register allocation of the small networks small (occupancy) instead of the maximum over all sizes.
template <int NET, bool NANS = false, class Load>
__device__ __forceinline__ float median_of(int n, Load load) {
    if (n <= 0) return __uint_as_float(0x7fc00000u);     // np.median of an empty slice
    if constexpr (NET == 0) {
        if constexpr (NANS) {
            float total = 0.f;
            for (int q = 0; q < n; ++q) total += load(q);
            if (total != total) return total;
        }
        return median_bisect(n, load);
    } else return median_network<NET, NANS>(n, load);
}

// Row gather through a buffer resource: one shared per-lane VGPR offset (the bin) plus a wave-uniform
// SGPR offset per row (buffer_load_dword v, v_bin, s[rsrc], s_row offen) -- no per-row address VGPRs.
struct RowGather {
    __amdgpu_buffer_rsrc_t rsrc;
    int bin_bytes;
    __device__ __forceinline__ float operator()(int row_bytes_offset) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, bin_bytes, row_bytes_offset, 0));
    }
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t channel_rsrc(const float* Vc, int64_t chan_stride) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Vc), 0, (int)(chan_stride * 4), 0x00020000);
}
// byte offset of the pad row for network slot k >= n (wave-uniform)
template <int NET>
__device__ __forceinline__ int pad_offset(int k, int n, int pad_row_bytes, int row_bytes) {
    const int low_pads = (NET - n) >> 1;
    return pad_row_bytes + ((k - n < low_pads) ? 0 : row_bytes);
}


__device__ __forceinline__ void emit(const MaskArgs& a, int c, int64_t t, int f, float m) {
    const int64_t o = c * a.chan_stride + t * a.FS + f;
    if (a.mask) a.mask[o] = m;
    if (a.X) { float2 x = a.X[o]; x.x *= m; x.y *= m; a.X[o] = x; }
}

// Pick the compiled network size for lists of at most `max_n` entries and call fn(integral_constant).
template <class Fn>
static void dispatch_net(int max_n, Fn&& fn) {
    if (max_n <= 2) fn(std::integral_constant<int, 2>{});
    else if (max_n <= 4) fn(std::integral_constant<int, 4>{});
    else if (max_n <= 8) fn(std::integral_constant<int, 8>{});
    else if (max_n <= 10) fn(std::integral_constant<int, 10>{});      // `simonline`'s default: a 10-s buffer holds ten peaks 1 s apart
    else if (max_n <= 12) fn(std::integral_constant<int, 12>{});      // (45 / 52 min-max instructions against the 16-wire network's 82)
    else if (max_n <= 16) fn(std::integral_constant<int, 16>{});
    else if (max_n <= 24) fn(std::integral_constant<int, 24>{});
    else if (max_n <= 32) fn(std::integral_constant<int, 32>{});
    else if (max_n <= 48) fn(std::integral_constant<int, 48>{});
    else if (max_n <= 64) fn(std::integral_constant<int, 64>{});
    else if (max_n <= 80) fn(std::integral_constant<int, 80>{});
    else if (max_n <= 100) fn(std::integral_constant<int, 100>{});
    else if (max_n <= 128) fn(std::integral_constant<int, 128>{});
    else fn(std::integral_constant<int, 0>{});
}

// instructions of the selection network compiled for lists of at most max_n entries (0: bisection, no network)
int median_network_instructions(int max_n, int* net_size) {
    int out = 0, size = 0;
    dispatch_net(max_n, [&](auto net) {
        constexpr int NET = decltype(net)::value;
        size = NET;
        if constexpr (NET > 0) out = MedianNet<NET>::kInstructions;
    });
    if (net_size) *net_size = size;
    return out;
}

// Housekeeping in front of a pipeline, one launch: the two pad rows of V (-1 and +inf: what the networks read for the slots
// past a list's end), optionally zeros over `z_count` floats of each of `n_z` regions of Z (the rows [T, Tpad) of the unit
// spectra, which the Gram tiles read), optionally the four counters of the peak refinement. (Three launches before: a
// fill kernel, a 2-D memset, a 16-byte memset -- 14 us in front of a 68-us STFT.)
__global__ void fill_pad_rows_kernel(float* V, int64_t chan_stride, int n_channels, int64_t pad_row, int FS,
                                     float* Z, int64_t z_stride, int64_t z_count, int n_z, unsigned int* stats, float* Z2) {
    const int y = blockIdx.y;
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (y < n_channels && k < 2 * FS) V[y * chan_stride + pad_row * FS + k] = (k < FS) ? -1.0f : INFINITY;
    if (y < n_z && 4 * k < z_count) {
        *reinterpret_cast<float4*>(Z + y * z_stride + 4 * k) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (Z2) *reinterpret_cast<float4*>(Z2 + y * z_stride + 4 * k) = make_float4(0.f, 0.f, 0.f, 0.f);     // a second array of the same geometry
    }
    if (stats && y == 0 && k < kStatWords) stats[k] = 0u;           // (every copy of the counters: common.h, kStatShards)
}

hipError_t launch_fill_pad_rows(float* V, int64_t chan_stride, int32_t n_channels, int64_t pad_row, int32_t FS,
                                hipStream_t s, float* Z, int64_t z_stride, int64_t z_count, int32_t n_z, unsigned int* stats, float* Z2) {
    if (!Z && Z2) { Z = Z2; Z2 = nullptr; }
    if (!Z || z_count <= 0) { Z = nullptr; Z2 = nullptr; z_count = 0; n_z = 0; }
    if ((z_count & 3) || (z_stride & 3)) return hipErrorInvalidValue;
    const int64_t per_row = std::max<int64_t>(std::max<int64_t>(2 * FS, z_count / 4), stats ? kStatWords : 0);
    hipLaunchKernelGGL(fill_pad_rows_kernel, dim3((unsigned)ceil_div(per_row, 256), (unsigned)std::max(n_channels, n_z)), dim3(256), 0, s,
                       V, chan_stride, n_channels, pad_row, FS, Z, z_stride, z_count, n_z, stats, Z2);
    return hipGetLastError();
}

// The row offsets of all network slots are computed ONCE in vector form -- lane l holds slots l and l + 64: the list entry
// (one coalesced load of the whole list) times the row pitch, or the pad row of a slot past the list's end -- and handed to
// the gathers through v_readlane. Written slot by slot ("k < n ? list[k] * row_bytes : pad") the compiler made a chain of
// scalar branches with ONE s_load_dword + s_waitcnt per slot: 100 scalar-memory round trips in a row, 13 900 of a wave's
// 30 400 cycles (tools/mask_spans.py). (The same offsets from a per-frame table in load order, read with seven
// s_load_dwordx16 and no vector instruction per slot, measured 0.436 against 0.440 ms: not worth the extra kernel.)
template <int NET>
struct SlotOffsets {
    int lo, hi;                     // byte offsets of slots lane and lane + 64
    __device__ __forceinline__ SlotOffsets(const int* list, int n, int row_bytes, int pad_bytes, int lane) {
        const int low_pads = (NET - n) >> 1;
        const int e0 = list[lane], e1 = NET > 64 ? list[lane + 64] : 0;          // rows of idx hold at least 128 entries
        lo = lane < n ? e0 * row_bytes : pad_bytes + ((lane - n < low_pads) ? 0 : row_bytes);
        hi = lane + 64 < n ? e1 * row_bytes : pad_bytes + ((lane + 64 - n < low_pads) ? 0 : row_bytes);
    }
    template <int K>
    __device__ __forceinline__ int of() const { return __builtin_amdgcn_readlane(K < 64 ? lo : hi, K & 63); }
};

// ---- REPET-SIM / online: list of similar frames per frame ----------------------------------------
// SPLIT: F-1 (= W/2) is a multiple of 64, so the wave-per-64-bins kernel covers bins [0, F-1) in whole
// blocks (4 per wave, balanced) and the lone Nyquist bin F-1 of 64 FRAMES is packed into one wave by
// mask_sim_nyquist_kernel; without it the 17th block would cost a full network pass for one lane.
template <int NET, bool SPLIT>
__global__ __launch_bounds__(256) void mask_sim_kernel(MaskArgs a, const int* __restrict__ idx, int idx_pitch,
                                                       const int* __restrict__ count, int64_t first_frame) {
    const int64_t t = a.frame0 + blockIdx.x;
    const int c = blockIdx.y;
    a.V += blockIdx.z * a.batch_stride;              // clip of a batch (n_batch == 1: blockIdx.z == 0)
    if (a.X) a.X += blockIdx.z * a.batch_stride;
    if (a.mask) a.mask += blockIdx.z * a.batch_stride;
    idx += blockIdx.z * a.idx_batch_stride;
    count += blockIdx.z * a.cnt_batch_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nbins = SPLIT ? a.F - 1 : a.F;
    const int nfb = (nbins + 63) >> 6;
    const float* Vc = a.V + c * a.chan_stride;
    if (t < first_frame) {           // online warm-up frames contribute nothing (repet.py:834)
        for (int f = threadIdx.x; f < a.F; f += 256) emit(a, c, t, f, 0.f);
        return;
    }
    const int64_t r = t - first_frame;
    const int n = count[r];
    const int* list = idx + r * (int64_t)idx_pitch;
    const int row_bytes = a.FS * 4, pad_bytes = (int)a.pad_row * row_bytes;
    RowGather g{channel_rsrc(Vc, a.chan_stride), 0};
    // the slots' row offsets once per wave in vector form, handed to the gathers by v_readlane (see SlotOffsets); lists of
    // more than 128 entries (bisection, NET == 0) keep the slot-by-slot form
    int off_lo = 0, off_hi = 0;
    if constexpr (NET > 0) {
        const SlotOffsets<NET> so(list, n, row_bytes, pad_bytes, lane);
        off_lo = so.lo; off_hi = so.hi;
    }
    for (int fb = wave; fb < nfb; fb += 4) {
        const int f = fb * 64 + lane;
        const bool active = f < nbins;
        const int fc = active ? f : nbins - 1;
        g.bin_bytes = fc * 4;
        // opaque copies: keep the per-slot offsets (readlanes / scalar ALU) inside the loop; hoisted, hipcc parks all NET
        // offsets in registers and halves the occupancy
        int n_it = n;
        asm volatile("" : "+s"(n_it));
        asm volatile("" : "+v"(off_lo), "+v"(off_hi));
        // the frame's own magnitude and spectrum bin do not depend on the median: fetch them before the network so
        // they arrive while it runs (3 more live registers) instead of after it
        const int64_t o = c * a.chan_stride + t * a.FS + fc;
        const float v_own = Vc[t * a.FS + fc];
        float2 x_own = make_float2(0.f, 0.f);
        if (a.X) x_own = a.X[o];
        const float med = median_of<NET>(n_it, [&](int k) {
            if constexpr (NET > 0) return g(__builtin_amdgcn_readlane(k < 64 ? off_lo : off_hi, k & 63));
            else return g(k < n_it ? list[k] * row_bytes : pad_offset<NET>(k, n_it, pad_bytes, row_bytes)); });
        if (active) {
            const float m = soft_mask(v_own, med, f, a.cutoff);
            if (a.mask) a.mask[o] = m;
            if (a.X) a.X[o] = make_float2(x_own.x * m, x_own.y * m);
        }
    }
}

// The same selection with FOUR bins per lane and a RUN of frames per workgroup (round 5; lists of at most 16 entries:
// `simonline`'s ten, short clips of `sim`). With one workgroup per frame the stage was bound by neither bytes (0.42 of HBM at
// cfg 5) nor instructions nor the gather path (16-byte gathers alone: -6 %), but by dependent round trips at the occupancy
// the registers allow: launch -> the frame's list -> its gathers -> the store, three latencies per workgroup that lives 4 us,
// twenty waves per CU. Here a workgroup walks kRun consecutive frames of one (clip, channel): the NEXT frame's list is
// fetched while the current frame's gathers are in flight, so a frame costs one round trip, and a lane owns bins
// 4 l .. 4 l + 3 of a 256-bin block -- every slot ONE 16-byte load per lane, four networks on the four components.
// The same values through the same networks: bit-identical to mask_sim_kernel (test_wide_mask_kernel_gives_the_same_bits).
// Measured at cfg 5 (64 x 30 s, ten-entry lists), stage ms: one bin per lane, 16-wire network 0.567; four bins per lane 0.537;
// + runs of frames with the list prefetched 0.511; + the 10-wire network (45 instructions against 82) 0.47 (0.50 on the
// one-bin kernel). Tried and dropped: XCD-ordered units (a clip's frames on ONE XCD: 0.557 -- the gathers' locality is not the
// bound: all of them from one row, 0.495); a second register set with the next frame's gathers in flight under the networks
// (0.489: three waves per SIMD instead of four). Without its stores the stage takes 0.40: it is the min / max issue rate and
// 0.68 GB of mask plane, in sum rather than in overlap.
constexpr int kWideRun = 16;
template <int NET>
__global__ __launch_bounds__(256) void mask_sim_wide_kernel(MaskArgs a, const int* __restrict__ idx, int idx_pitch,
                                                            const int* __restrict__ count, int64_t first_frame, unsigned n_frames) {
    static_assert(NET >= 2 && NET <= 64, "wide selection: short lists only");
    const int c = blockIdx.y;
    a.V += blockIdx.z * a.batch_stride;
    if (a.X) a.X += blockIdx.z * a.batch_stride;
    if (a.mask) a.mask += blockIdx.z * a.batch_stride;
    idx += blockIdx.z * a.idx_batch_stride;
    count += blockIdx.z * a.cnt_batch_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nbins = a.F - 1;                         // a multiple of 256 (launch_mask_sim); the Nyquist bin: mask_sim_nyquist_kernel
    const int nfb = nbins >> 8;
    const float* Vc = a.V + c * a.chan_stride;
    const int row_bytes = a.FS * 4, pad_bytes = (int)a.pad_row * row_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = channel_rsrc(Vc, a.chan_stride);
    const unsigned f_begin = blockIdx.x * kWideRun, f_end = min(f_begin + kWideRun, n_frames);
    // the list of a frame, as the gathers need it: its length (wave-uniform) and entry `lane` (rows of idx hold >= 128 entries)
    auto fetch_list = [&](int64_t t, int& n, int& e) {
        if (t < first_frame) { n = -1; e = 0; return; }                         // warm-up frame: nothing to gather
        const int64_t r = t - first_frame;
        n = count[r];
        e = idx[r * (int64_t)idx_pitch + lane];
    };
    for (int fb = wave; fb < nfb; fb += 4) {
        const int f0 = (fb << 8) + 4 * lane;
        const int bin_bytes = f0 * 4;
        int n_next, e_next;
        fetch_list(a.frame0 + f_begin, n_next, e_next);
        for (unsigned fr = f_begin; fr < f_end; ++fr) {
            const int64_t t = a.frame0 + fr;
            const int n = __builtin_amdgcn_readfirstlane(n_next), e = e_next;
            if (fr + 1 < f_end) fetch_list(t + 1, n_next, e_next);              // in flight beside this frame's gathers
            const int64_t o = c * a.chan_stride + t * a.FS + f0;
            if (n < 0) {                                                        // online warm-up frames contribute nothing (repet.py:834)
                if (a.mask) *reinterpret_cast<float4*>(a.mask + o) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.X) {
                    float4 x01 = *reinterpret_cast<const float4*>(a.X + o), x23 = *reinterpret_cast<const float4*>(a.X + o + 2);
                    *reinterpret_cast<float4*>(a.X + o) = make_float4(x01.x * 0.f, x01.y * 0.f, x01.z * 0.f, x01.w * 0.f);
                    *reinterpret_cast<float4*>(a.X + o + 2) = make_float4(x23.x * 0.f, x23.y * 0.f, x23.z * 0.f, x23.w * 0.f);
                }
                continue;
            }
            // byte offset of slot `lane`'s row: the list entry, or the low / high pad row past the list's end (SlotOffsets)
            const int low_pads = (NET - n) >> 1;
            int off = lane < n ? e * row_bytes : pad_bytes + ((lane - n < low_pads) ? 0 : row_bytes);
            asm volatile("" : "+v"(off));
            const float4 v_own = *reinterpret_cast<const float4*>(Vc + t * a.FS + f0);
            float4 x01 = make_float4(0.f, 0.f, 0.f, 0.f), x23 = x01;
            if (a.X) { x01 = *reinterpret_cast<const float4*>(a.X + o); x23 = *reinterpret_cast<const float4*>(a.X + o + 2); }
            float4 w[NET];
#pragma unroll
            for (int q = 0; q < NET; ++q) {
                const int k = MedianNet<NET>::kLoadOrder[q];
                w[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, bin_bytes, __builtin_amdgcn_readlane(off, k), 0));
            }
            float med[4];
            if (n == 0) { med[0] = med[1] = med[2] = med[3] = __uint_as_float(0x7fc00000u); }      // np.median of an empty slice
            else {
                med[0] = median_network<NET>(n, [&](int k) { return w[k].x; });
                med[1] = median_network<NET>(n, [&](int k) { return w[k].y; });
                med[2] = median_network<NET>(n, [&](int k) { return w[k].z; });
                med[3] = median_network<NET>(n, [&](int k) { return w[k].w; });
            }
            const float m0 = soft_mask(v_own.x, med[0], f0, a.cutoff), m1 = soft_mask(v_own.y, med[1], f0 + 1, a.cutoff);
            const float m2 = soft_mask(v_own.z, med[2], f0 + 2, a.cutoff), m3 = soft_mask(v_own.w, med[3], f0 + 3, a.cutoff);
            if (a.mask) *reinterpret_cast<float4*>(a.mask + o) = make_float4(m0, m1, m2, m3);
            if (a.X) {
                *reinterpret_cast<float4*>(a.X + o) = make_float4(x01.x * m0, x01.y * m0, x01.z * m1, x01.w * m1);
                *reinterpret_cast<float4*>(a.X + o + 2) = make_float4(x23.x * m2, x23.y * m2, x23.z * m3, x23.w * m3);
            }
        }
    }
    // (the Nyquist bin of a warm-up frame: mask_sim_kernel's emit(0) wrote all F bins; here bin F - 1 of frames < first_frame)
    if (first_frame > a.frame0 + f_begin && threadIdx.x < kWideRun) {
        const int64_t t = a.frame0 + f_begin + threadIdx.x;
        if (t < first_frame && f_begin + threadIdx.x < f_end) emit(a, c, t, a.F - 1, 0.f);
    }
}
// REPET_MASK_WIDE=0: the one-bin-per-lane kernel for every list length (agreement test / A-B)
static bool mask_wide_enabled() {
    static const bool on = [] { const char* e = getenv("REPET_MASK_WIDE"); return !(e && e[0] == '0'); }();
    return on;
}

// Rank-domain form of the kernel above (rank.hip): a wave owns 128 bins of a frame, two per lane. The gather reads one
// dword = the 16-bit rank codes of bins (2 lane, 2 lane + 1) of a similar frame, the network runs on both bins at
// once with packed 16-bit min/max, and the one or two middle codes of each bin are turned back into magnitudes through
// the sorted-column table Vs. Selecting on "number of smaller values in the column" is selecting on the values, so the
// mask is bit-identical to mask_sim_kernel's -- at half the instructions per bin and half the gather bytes.
template <int NET, size_t... Q>
__device__ __forceinline__ void gather_codes_in_network_order(unsigned (&w)[NET], __amdgpu_buffer_rsrc_t rsrc, int bin_bytes,
                                                              const SlotOffsets<NET>& offsets, std::index_sequence<Q...>) {
    ((w[MedianNetPk<NET>::kLoadOrder[Q]] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(
          rsrc, bin_bytes, offsets.template of<MedianNetPk<NET>::kLoadOrder[Q]>(), 0)), ...);
}

// Scheduling: the unit of work is (channel, block of 128 bins, 4 consecutive frames) = one workgroup, one frame per
// wave. Workgroups b and b + 8 land on the same XCD (round-robin placement), so XCD x is given the (channel, bin block)
// combinations x, x + 8, ... one after the other, each over all frames: its L2 then holds the 2 MB of codes and the
// 4 MB of sorted columns of ONE combination at a time instead of the whole 48 MB, and most gathers hit L2 rather than
// the Infinity Cache. (Placement only decides speed; any placement gives the same result.)
// Lookups: the frame's own code tells whether its magnitude is at or below the lower middle value -- then
// min(V, median) = V and the mask is exactly 1 (about half of all cells: the frame is usually in its own list), and
// the lane skips the scattered table reads.
#ifdef REPET_MASK_STAMPS
// diagnostic build (make stamps, tools/mask_spans.py): wave 0 of every fourth workgroup records its start and end on the
// chip-wide 100 MHz clock and, in core cycles since its start, when its gathers were issued and when the network was done
__device__ unsigned long long g_mask_span[4 * 16384];
#define MSTAMP_BEGIN const bool ms_on = (blockIdx.x & 3) == 1 && (blockIdx.x >> 2) < 16384 && threadIdx.x == 0; \
    const unsigned long long ms_r0 = __builtin_amdgcn_s_memrealtime(), ms_c0 = __builtin_amdgcn_s_memtime(); unsigned long long ms_c1 = 0, ms_c2 = 0;
#define MSTAMP(v) v = __builtin_amdgcn_s_memtime() - ms_c0;
#define MSTAMP_END if (ms_on) { unsigned long long* q_ = g_mask_span + 4 * (blockIdx.x >> 2); q_[0] = ms_r0; q_[1] = __builtin_amdgcn_s_memrealtime(); q_[2] = ms_c1; q_[3] = ms_c2; }
#else
#define MSTAMP_BEGIN
#define MSTAMP(v)
#define MSTAMP_END
#endif

template <int NET>
__global__ __launch_bounds__(256) void mask_sim_rank_kernel(MaskArgs a, const int* __restrict__ idx, int idx_pitch,
                                                            const int* __restrict__ count, int n_quads) {
    static_assert(NET >= 2, "the rank path has no bisection fallback");
    MSTAMP_BEGIN
    // the wave number as a scalar: everything derived from it (frame, list, row offsets) then lives in SGPRs
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nfb = (a.F - 1) >> 7;                   // whole blocks of 128 bins; bin F-1: mask_sim_nyquist_kernel
    const int b = blockIdx.x, x = b & 7, i = b >> 3;
    const int combo = (i / n_quads) * 8 + x;
    if (combo >= a.n_channels * nfb) return;
    const int c = combo / nfb, fb = combo % nfb;
    const int64_t t_end = a.frame_end > 0 ? a.frame_end : a.T;
    const int64_t t = a.frame0 + 4 * (int64_t)(i % n_quads) + wave;
    if (t >= t_end) return;
    const float* Vc = a.V + c * a.chan_stride;
    const int n = count[t];
    const int* list = idx + t * (int64_t)idx_pitch;
    const int row_bytes = a.FS * 2, pad_bytes = (int)a.pad_row * row_bytes;
    const unsigned short* Rc = a.R + c * a.r_chan_stride;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Rc), 0, (int)(a.r_chan_stride * 2), 0x00020000);
    const unsigned t_last = (unsigned)(a.T - 1);
    const int f0 = fb * 128 + 2 * lane;
    const int64_t o = c * a.chan_stride + t * a.FS + f0;
    const float2 v_own = *reinterpret_cast<const float2*>(Vc + t * a.FS + f0);
    const unsigned own = *reinterpret_cast<const unsigned*>(Rc + t * a.FS + f0);
    float4 x_own = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.X) x_own = *reinterpret_cast<const float4*>(a.X + o);
    float med0 = __uint_as_float(0x7fc00000u), med1 = med0;              // np.median of an empty slice
    if (n > 0) {
        unsigned w[NET];
        const SlotOffsets<NET> offsets(list, n, row_bytes, pad_bytes, lane);
        gather_codes_in_network_order<NET>(w, rsrc, f0 * 2, offsets, std::make_index_sequence<NET>{});
        MSTAMP(ms_c1)
#ifdef REPET_MASK_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // diagnostic build only: all gathers landed
        { const unsigned long long landed = __builtin_amdgcn_s_memtime() - ms_c0; ms_c1 |= landed << 32; }
#endif
        MedianNetPk<NET>::run(w);
        asm volatile("" :: "v"(w[NET / 2 - 1]), "v"(w[NET / 2]));
        MSTAMP(ms_c2)
        const float* vs = a.Vs + ((int64_t)c * a.n_rank_cols + f0) * a.vs_pitch;
        const unsigned lo = w[NET / 2 - 1], hi = w[NET / 2];
        const unsigned lo0 = lo & 0xffffu, lo1 = lo >> 16;
        // lower middle value >= own value  =>  median >= own value  =>  mask = (V + eps) / (V + eps) = 1: any model >= V
        // gives the same bits, so V itself stands in for it
        med0 = v_own.x; med1 = v_own.y;
        // both bins' lookups behind ONE branch, all four loads issued before the first is used: a load inside a branch is
        // waited for at its join, and two branches in a row were two table round trips per wave
        const bool need0 = lo0 < (own & 0xffffu), need1 = lo1 < (own >> 16);
        if (need0 || need1) {
            const float a0 = vs[min(lo0 - kRankCodeBase, t_last)], b0 = vs[min((hi & 0xffffu) - kRankCodeBase, t_last)];
            const float a1 = vs[a.vs_pitch + min(lo1 - kRankCodeBase, t_last)], b1 = vs[a.vs_pitch + min((hi >> 16) - kRankCodeBase, t_last)];
            if (need0) med0 = (n & 1) ? a0 : 0.5f * (a0 + b0);
            if (need1) med1 = (n & 1) ? a1 : 0.5f * (a1 + b1);
        }
    }
    const float m0 = soft_mask(v_own.x, med0, f0, a.cutoff), m1 = soft_mask(v_own.y, med1, f0 + 1, a.cutoff);
    if (a.mask) *reinterpret_cast<float2*>(a.mask + o) = make_float2(m0, m1);
    if (a.X) *reinterpret_cast<float4*>(a.X + o) = make_float4(x_own.x * m0, x_own.y * m0, x_own.z * m1, x_own.w * m1);
    MSTAMP_END
}

#ifdef REPET_MASK_STAMPS
}  // namespace repet
extern "C" int repet_debug_mask_spans(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(repet::g_mask_span), sizeof(unsigned long long) * 4 * n);
}
namespace repet {
#endif

// One lane per frame, bin F-1 only: the index list, its length and every row offset are per-lane here.
// PRELOAD (round 6; lists of at most 128 entries in rows of a multiple of four): a lane fetches its whole list first, as 16-byte
// loads that do not depend on each other, and the NET row gathers follow side by side. Entry by entry ("list[k]", then the
// row it names) the kernel was a chain of 2 x 100 memory round trips per lane with 64 lines per instruction: 84 us for 244
// waves at cfg 2 -- hidden beside the selection kernel, but serial latency all the same. The selection network and its
// inputs are unchanged: the same bits.
template <int NET, bool PRELOAD = false>
__global__ __launch_bounds__(64) void mask_sim_nyquist_kernel(MaskArgs a, const int* __restrict__ idx, int idx_pitch,
                                                              const int* __restrict__ count, int64_t first_frame) {
    const int c = blockIdx.y;
    a.V += blockIdx.z * a.batch_stride;
    if (a.X) a.X += blockIdx.z * a.batch_stride;
    if (a.mask) a.mask += blockIdx.z * a.batch_stride;
    idx += blockIdx.z * a.idx_batch_stride;
    count += blockIdx.z * a.cnt_batch_stride;
    const int64_t r0 = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t n_rows = a.T - first_frame;       // first_frame >= frame0: rows before it are warm-up frames
    const bool active = r0 < n_rows;
    const int64_t r = active ? r0 : n_rows - 1;
    const int64_t t = first_frame + r;
    const float* Vc = a.V + c * a.chan_stride;
    const int n = count[r];
    const int* list = idx + r * (int64_t)idx_pitch;
    const int f = a.F - 1;
    const int row_bytes = a.FS * 4, pad_bytes = (int)a.pad_row * row_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = channel_rsrc(Vc, a.chan_stride);
    if constexpr (PRELOAD && NET >= 2) {
        constexpr int NQ = (NET + 3) / 4;
        int e[4 * NQ];
        const int4* list4 = reinterpret_cast<const int4*>(list);
#pragma unroll
        for (int q = 0; q < NQ; ++q) { const int4 v = list4[q]; e[4 * q] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w; }
        const float v_own = Vc[t * a.FS + f];
        const float med = median_of<NET>(n, [&](int k) {
            const int row_off = k < n ? e[k] * row_bytes : pad_offset<NET>(k, n, pad_bytes, row_bytes);
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, row_off + f * 4, 0, 0)); });
        if (active) emit(a, c, t, f, soft_mask(v_own, med, f, a.cutoff));
        return;
    }
    const float med = median_of<NET>(n, [&](int k) {
        const int row_off = k < n ? list[k] * row_bytes : pad_offset<NET>(k, n, pad_bytes, row_bytes);
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, row_off + f * 4, 0, 0)); });
    if (active) emit(a, c, t, f, soft_mask(Vc[t * a.FS + f], med, f, a.cutoff));
}

// Round 6: one WAVE per (frame, channel) for that bin. The lane-per-frame kernel above walks its list entry by entry -- a
// hundred dependent list -> row round trips per lane and the 776-comparator network for ONE cell per lane: 84 us beside the
// selection at cfg 2 (244 waves), 7 % of all kernel time. Here lane l holds entries l and l + 64 of the list (rows of idx
// hold at least 128), so the whole gather is two loads per lane in flight at once, and the two middle order statistics come
// from counting: the rank of an entry = entries below it (+ equal ones in front of it), one v_readlane per entry. The
// network's pad slots (-1 below, +inf above, pad_offset) take part as counts, so the wire read is the network's own --
// same values in the same (signed-integer, as v_min_i32) order: the same bits.
__global__ __launch_bounds__(64) void mask_sim_nyquist_wave_kernel(MaskArgs a, const int* __restrict__ idx, int idx_pitch,
                                                                   const int* __restrict__ count, int64_t first_frame, int net) {
    const int c = blockIdx.y;
    a.V += blockIdx.z * a.batch_stride;
    if (a.X) a.X += blockIdx.z * a.batch_stride;
    if (a.mask) a.mask += blockIdx.z * a.batch_stride;
    idx += blockIdx.z * a.idx_batch_stride;
    count += blockIdx.z * a.cnt_batch_stride;
    const int64_t r = blockIdx.x, t = first_frame + r;
    const int lane = threadIdx.x;
    const float* Vc = a.V + c * a.chan_stride;
    const int f = a.F - 1;
    const int n = __builtin_amdgcn_readfirstlane(count[r]);
    const int* list = idx + r * (int64_t)idx_pitch;
    const int e0 = list[lane], e1 = list[lane + 64];
    const int64_t o = c * a.chan_stride + t * a.FS + f;
    const float v_own = Vc[t * a.FS + f];
    float2 x_own = make_float2(0.f, 0.f);
    if (a.X && lane == 0) x_own = a.X[o];
    const int v0 = lane < n ? __float_as_int(Vc[(int64_t)e0 * a.FS + f]) : 0;
    const int v1 = lane + 64 < n ? __float_as_int(Vc[(int64_t)e1 * a.FS + f]) : 0;
    int c0 = 0, c1 = 0;
    const int n_lo = n < 64 ? n : 64;
    for (int k = 0; k < n_lo; ++k) {
        const int s = __builtin_amdgcn_readlane(v0, k);
        c0 += (s < v0 || (s == v0 && k < lane)) ? 1 : 0;
        c1 += (s <= v1) ? 1 : 0;                                        // (entry k < 64 stands in front of entry lane + 64)
    }
    for (int k = 64; k < n; ++k) {
        const int s = __builtin_amdgcn_readlane(v1, k - 64);
        c0 += (s < v0) ? 1 : 0;
        c1 += (s < v1 || (s == v1 && k - 64 < lane)) ? 1 : 0;
    }
    // the pads as the network sees them: (net - n) / 2 slots of -1.0f, the rest +inf
    const int low_pads = (net - n) >> 1, high_pads = net - n - low_pads;
    const int kLow = (int)0xbf800000u, kHigh = 0x7f800000;
    const int m0 = c0 + (kLow < v0 ? low_pads : 0) + (kHigh < v0 ? high_pads : 0);
    const int m1 = c1 + (kLow < v1 ? low_pads : 0) + (kHigh < v1 ? high_pads : 0);
    auto wire = [&](int w) -> float {                                   // what the sorted network holds at wire w
        const unsigned long long h0 = __ballot(lane < n && m0 == w), h1 = __ballot(lane + 64 < n && m1 == w);
        if (h0) return __int_as_float(__builtin_amdgcn_readlane(v0, (int)__ffsll((long long)h0) - 1));
        if (h1) return __int_as_float(__builtin_amdgcn_readlane(v1, (int)__ffsll((long long)h1) - 1));
        return w < low_pads ? -1.0f : INFINITY;
    };
    float med = __uint_as_float(0x7fc00000u);                           // np.median of an empty slice
    if (n > 0) {
        const float lo = wire(net / 2 - 1);
        med = (n & 1) ? lo : 0.5f * (lo + wire(net / 2));
    }
    if (lane == 0) {
        const float m = soft_mask(v_own, med, f, a.cutoff);
        if (a.mask) a.mask[o] = m;
        if (a.X) a.X[o] = make_float2(x_own.x * m, x_own.y * m);
    }
}
// REPET_NYQUIST=wave: the wave-per-frame kernel; =lane: the lane-per-frame kernel walking its list entry by entry (round 5);
// default: the lane-per-frame kernel with its list fetched up front (agreement test / A-B)
static int nyquist_path() {
    static const int path = [] { const char* e = getenv("REPET_NYQUIST"); return (e && e[0] == 'w') ? 1 : (e && e[0] == 'l' && e[1] == 'a') ? 2 : 0; }();   // wave | lane | (lists)
    return path;
}

hipError_t launch_mask_sim(const MaskArgs& m, const int32_t* idx, int32_t idx_pitch, const int32_t* count,
                           int64_t first_frame, int32_t max_count, hipStream_t s, hipStream_t side,
                           hipEvent_t fork, hipEvent_t join, int parts, bool lookups_by_caller) {
    const int64_t t_end = m.frame_end > 0 ? m.frame_end : m.T;
    if (t_end - m.frame0 <= 0) return hipSuccess;
    // Round 6: the Nyquist-bin kernel runs on the MAIN stream, in front of the selection. With its list fetched up front it is short,
    // and beside the selection (rounds 2-5: a fork and a join) it cost that kernel what it took itself: selection stage 0.181 ->
    // 0.176 ms at cfg 2 with the kernel in line, cfg 5 2.27 -> 2.21 ms (profiles/r06_nyquist_stream_ab.txt).
    // REPET_NYQUIST_STREAM=side: the forked form.
    static const bool nyq_side = [] { const char* e = getenv("REPET_NYQUIST_STREAM"); return e && e[0] == 's'; }();
    const bool forked = nyq_side && side != nullptr && fork != nullptr && join != nullptr && parts == 3;
    const bool split = m.F > 64 && ((m.F - 1) & 63) == 0;
    const int64_t rows = m.T - first_frame;
    const unsigned n_launch = (unsigned)(t_end - m.frame0);   // frames [frame0, t_end) are processed
    const unsigned nb = (unsigned)(m.n_batch > 1 ? m.n_batch : 1);
    hipError_t bits_error = hipSuccess;
    dispatch_net(max_count, [&](auto net) {
        constexpr int NET = decltype(net)::value;
        if (split) {
            if (rows > 0 && (parts & 2)) {
                if (forked) { (void)hipEventRecord(fork, s); (void)hipStreamWaitEvent(side, fork, 0); }
                if (NET >= 2 && idx_pitch >= 128 && nyquist_path() == 1)
                    hipLaunchKernelGGL(mask_sim_nyquist_wave_kernel, dim3((unsigned)rows, (unsigned)m.n_channels, nb),
                                       dim3(64), 0, forked ? side : s, m, idx, idx_pitch, count, first_frame, (int)NET);
                else if (NET >= 2 && idx_pitch >= 128 && (idx_pitch & 3) == 0 && nyquist_path() == 0)
                    hipLaunchKernelGGL((mask_sim_nyquist_kernel<NET, true>), dim3((unsigned)ceil_div(rows, 64), (unsigned)m.n_channels, nb),
                                       dim3(64), 0, forked ? side : s, m, idx, idx_pitch, count, first_frame);
                else
                    hipLaunchKernelGGL(mask_sim_nyquist_kernel<NET>, dim3((unsigned)ceil_div(rows, 64), (unsigned)m.n_channels, nb),
                                       dim3(64), 0, forked ? side : s, m, idx, idx_pitch, count, first_frame);
                if (forked) (void)hipEventRecord(join, side);
            }
            if (parts & 1) {
                if constexpr (NET >= 2) {
                    if (m.P != nullptr && nb == 1 && first_frame == 0 &&
                        mask_sim_bits_supported(m.T, m.n_channels, m.n_rank_cols, max_count) && m.n_rank_cols == m.F - 1) {
                        bits_error = launch_mask_sim_bits(m, idx, idx_pitch, count, max_count, n_launch, s);
                        if (bits_error == hipSuccess && !lookups_by_caller) bits_error = launch_mask_from_codes(m, count, s);
                    } else if (m.R != nullptr && ((m.F - 1) & 127) == 0 && nb == 1 && first_frame == 0) {
                        const int n_quads = (int)ceil_div(n_launch, 4);
                        const int combos = m.n_channels * ((m.F - 1) >> 7);
                        hipLaunchKernelGGL(mask_sim_rank_kernel<NET>, dim3((unsigned)(8 * ceil_div(combos, 8) * n_quads)), dim3(256), 0, s,
                                           m, idx, idx_pitch, count, n_quads);
                    } else if (NET <= 16 && ((m.F - 1) & 255) == 0 && (m.FS & 3) == 0 && mask_wide_enabled()) {
                        if constexpr (NET <= 16)
                            hipLaunchKernelGGL(mask_sim_wide_kernel<NET>, dim3((unsigned)ceil_div(n_launch, kWideRun), (unsigned)m.n_channels, nb), dim3(256), 0, s,
                                               m, idx, idx_pitch, count, first_frame, n_launch);
                    } else
                        hipLaunchKernelGGL((mask_sim_kernel<NET, true>), dim3(n_launch, (unsigned)m.n_channels, nb), dim3(256), 0, s,
                                           m, idx, idx_pitch, count, first_frame);
                } else {
                    hipLaunchKernelGGL((mask_sim_kernel<NET, true>), dim3(n_launch, (unsigned)m.n_channels, nb), dim3(256), 0, s,
                                       m, idx, idx_pitch, count, first_frame);
                }
            }
            if (forked && rows > 0) (void)hipStreamWaitEvent(s, join, 0);
        } else if (parts & 1) {
            hipLaunchKernelGGL((mask_sim_kernel<NET, false>), dim3(n_launch, (unsigned)m.n_channels, nb), dim3(256), 0, s,
                               m, idx, idx_pitch, count, first_frame);
        }
    });
    if (bits_error != hipSuccess) return bits_error;
    return hipGetLastError();
}

// ---- adaptive: taps at i + {..}*period[i] (repet.py:1478-1498) ------------------------------------
template <int NET>
__global__ __launch_bounds__(256) void mask_adaptive_kernel(MaskArgs a, const int* __restrict__ periods, int order) {
    const int64_t t = blockIdx.x;
    const int c = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nfb = (a.F + 63) >> 6;
    const float* Vc = a.V + c * a.chan_stride;
    const int64_t per = periods[t];
    // center_indices = arange(1, order+1) - ceil(order/2); the in-range taps form a contiguous run
    const int first_tap = 1 - ((order + 1) >> 1);
    int k_lo = 0, n = 0;
    for (int k = 0; k < order; ++k) {
        const int64_t j = t + (int64_t)(first_tap + k) * per;
        if (j >= 0 && j < a.T) { if (n == 0) k_lo = k; ++n; }
    }
    const int row_bytes = a.FS * 4, pad_bytes = (int)a.pad_row * row_bytes;
    const int base_bytes = (int)(t + (int64_t)(first_tap + k_lo) * per) * row_bytes, step_bytes = (int)per * row_bytes;
    RowGather g{channel_rsrc(Vc, a.chan_stride), 0};
    for (int fb = wave; fb < nfb; fb += 4) {
        const int f = fb * 64 + lane;
        const bool active = f < a.F;
        const int fc = active ? f : a.F - 1;
        g.bin_bytes = fc * 4;
        // opaque copy: keeps the per-slot row-offset selection (scalar ALU) inside the loop; hoisted, hipcc
        // parks all NET offsets in VGPRs and halves the occupancy
        int n_it = n;
        asm volatile("" : "+s"(n_it));
        const int64_t o = c * a.chan_stride + t * a.FS + fc;     // own magnitude / spectrum bin: in flight during the network
        const float v_own = Vc[t * a.FS + fc];
        float2 x_own = make_float2(0.f, 0.f);
        if (a.X) x_own = a.X[o];
        const float med = median_of<NET, true>(n_it, [&](int k) {
            return g(k < n_it ? base_bytes + k * step_bytes : pad_offset<NET>(k, n_it, pad_bytes, row_bytes)); });
        if (active) {
            const float m = soft_mask(v_own, med, f, a.cutoff);
            if (a.mask) a.mask[o] = m;
            if (a.X) a.X[o] = make_float2(x_own.x * m, x_own.y * m);
        }
    }
}

hipError_t launch_mask_adaptive(const MaskArgs& m, const int32_t* periods, int32_t order, hipStream_t s) {
    if (m.T <= 0) return hipSuccess;
    dispatch_net(order, [&](auto net) {
        hipLaunchKernelGGL(mask_adaptive_kernel<decltype(net)::value>, dim3((unsigned)m.T, (unsigned)m.n_channels),
                           dim3(256), 0, s, m, periods, order);
    });
    return hipGetLastError();
}

// ---- original / extended: one median per position q inside the period (repet.py:1401-1446) ---------
// grid (period_max, C); workgroups with q >= period exit. The model of position q is the median over
// the segments that really contain frame s*p+q (all S for q < T-(S-1)p, else the first S-1); it is
// computed once and applied to every segment. The period is read on the device, so the network size
// is chosen for the smallest admissible period (most segments): max_segments = ceil(T / min_period).
// The period is estimated on the device just before, so the number of segments n = ceil(T / p) is only known inside
// the kernel: the network is picked THERE (one wave-uniform switch over the compiled sizes) instead of compiling the
// launch for the shortest admissible period -- which for a 3-minute clip meant 177 > 128 "possible" segments and the
// bisection fallback for a list that really has 107 entries (0.67 ms; 0.1 ms now).
template <int NET>
__device__ __forceinline__ float period_median(int n, const RowGather& g, int base_bytes, int step_bytes, int pad_bytes,
                                               int row_bytes) {
    // opaque copy: keeps the per-slot row-offset selection (scalar ALU) inside the caller's loop
    int n_it = n;
    asm volatile("" : "+s"(n_it));
    return median_of<NET, true>(n_it, [&](int k) {
        return g(k < n_it ? base_bytes + k * step_bytes : pad_offset<NET>(k, n_it, pad_bytes, row_bytes)); });
}

// NET > 0: the launch knows that no position has more than NET segments (short clips, the segments of `extended`) and
// compiles that one network -- fewer registers, more waves; NET == -1: the switch described above.
template <int NET>
__global__ __launch_bounds__(256) void mask_period_kernel(MaskArgs a, const int* __restrict__ period_dev,
                                                          int period_host, int parts) {
    const int bz = blockIdx.z;                  // clip of the batch (segments of `extended`)
    const int p = period_dev ? period_dev[bz] : period_host;
    const int q = blockIdx.x;
    if (q >= p) return;
    const int c = blockIdx.y % a.n_channels, part = blockIdx.y / a.n_channels;    // `parts` workgroups share the bins
    a.V += bz * a.batch_stride;
    if (a.X) a.X += bz * a.batch_stride;
    if (a.mask) a.mask += bz * a.batch_stride;
    float* model_row = a.model ? a.model + bz * a.model_batch_stride + c * a.model_chan_stride + (int64_t)q * a.FS : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nfb = (a.F + 63) >> 6;
    const float* Vc = a.V + c * a.chan_stride;
    const int S = (int)((a.T + p - 1) / p);
    const int n = (q < a.T - (int64_t)(S - 1) * p) ? S : S - 1;     // segments holding real data at q
    const int row_bytes = a.FS * 4, pad_bytes = (int)a.pad_row * row_bytes;
    const int base_bytes = q * row_bytes, step_bytes = p * row_bytes;
    RowGather g{channel_rsrc(Vc, a.chan_stride), 0};
    for (int fb = wave + 4 * part; fb < nfb; fb += 4 * parts) {
        const int f = fb * 64 + lane;
        const bool active = f < a.F;
        const int fc = active ? f : a.F - 1;
        g.bin_bytes = fc * 4;
        float med;
        if constexpr (NET >= 0) med = period_median<NET>(n, g, base_bytes, step_bytes, pad_bytes, row_bytes);
        else if (n <= 8) med = period_median<8>(n, g, base_bytes, step_bytes, pad_bytes, row_bytes);
        else if (n <= 16) med = period_median<16>(n, g, base_bytes, step_bytes, pad_bytes, row_bytes);
        else if (n <= 32) med = period_median<32>(n, g, base_bytes, step_bytes, pad_bytes, row_bytes);
        else if (n <= 64) med = period_median<64>(n, g, base_bytes, step_bytes, pad_bytes, row_bytes);
        else if (n <= 100) med = period_median<100>(n, g, base_bytes, step_bytes, pad_bytes, row_bytes);
        else if (n <= 128) med = period_median<128>(n, g, base_bytes, step_bytes, pad_bytes, row_bytes);
        else med = period_median<0>(n, g, base_bytes, step_bytes, pad_bytes, row_bytes);
        if (model_row) {                 // the inverse STFT applies the model itself (IstftOlaArgs::model)
            if (active) model_row[f] = med;
            continue;
        }
        // every segment's frame at this position gets the mask of the shared model: four frames per round, all their
        // loads first (a load -> multiply -> store chain per frame would pay the memory latency n times in a row)
        if (active)
            for (int s0 = 0; s0 < n; s0 += 4) {
                float v[4];
                float2 x[4];
                int64_t o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int s = s0 + u < n ? s0 + u : n - 1;
                    const int64_t t = (int64_t)s * p + q;
                    o[u] = c * a.chan_stride + t * a.FS + f;
                    v[u] = Vc[t * a.FS + f];
                    x[u] = a.X ? a.X[o[u]] : make_float2(0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (s0 + u >= n) break;
                    const float m = soft_mask(v[u], med, f, a.cutoff);
                    if (a.mask) a.mask[o[u]] = m;
                    if (a.X) a.X[o[u]] = make_float2(x[u].x * m, x[u].y * m);
                }
            }
    }
}

hipError_t launch_mask_period(const MaskArgs& m, const int32_t* period_dev, int32_t period_host,
                              int32_t min_period, hipStream_t s) {
    if (m.T <= 0) return hipSuccess;
    // the period lives on the device when it was just estimated there; the grid covers the largest
    // admissible period (a third of the frames, repet.py:1266) and surplus workgroups exit
    unsigned gx = period_dev ? (unsigned)(m.T / 3 + 2) : (unsigned)period_host;
    if (gx < 1) gx = 1;
    const int nb = m.n_batch > 0 ? m.n_batch : 1;
    const int pmin = period_dev ? (min_period > 0 ? min_period : 1) : period_host;
    const int max_segments = (int)((m.T + pmin - 1) / pmin);
    // few positions x channels x clips (one long clip): let several workgroups share a position's frequency blocks
    const int64_t useful = (int64_t)(period_dev ? std::max(min_period, 1) : period_host) * m.n_channels * nb;
    const int nfb = (m.F + 63) >> 6;
    int parts = 1;
    while (parts < 4 && useful * parts < 2048 && 4 * parts < nfb) parts *= 2;
    const dim3 grid(gx, (unsigned)(m.n_channels * parts), (unsigned)nb);
    if (max_segments <= 32) {        // a tight bound and a small network: the single-network kernel
        dispatch_net(max_segments, [&](auto net) {
            hipLaunchKernelGGL(mask_period_kernel<decltype(net)::value>, grid, dim3(256), 0, s, m, period_dev, period_host, parts);
        });
    } else {
        hipLaunchKernelGGL(mask_period_kernel<-1>, grid, dim3(256), 0, s, m, period_dev, period_host, parts);
    }
    return hipGetLastError();
}

}  // namespace repet
