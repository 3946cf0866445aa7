// Persistent bidirectional-LSTM recurrence for the hint/text encoder.
//
// Replaces nn.Embedding + pack_padded_sequence + nn.LSTM(D, D, bidirectional) + mean(h_n)
// (reference models/modules.py:77-90).
//
// Design:
//   * The vocabulary is tiny (43 rows), so the input projection  W_ih emb[v] + b_ih + b_hh  is a [V, 4D] gate
//     table per direction built once per call by the tiled GEMM (tg_gemm.hip); the recurrence gathers table rows
//     instead of multiplying by W_ih at every token.
//   * One workgroup owns 32 sequences of one direction for the whole recurrence (no inter-workgroup exchange,
//     no per-step launches): h lives in LDS (MFMA A operand), c in registers.  Wave w owns hidden units
//     [w*D/4, (w+1)*D/4) for all four gates, so the cell update is lane-local in the MFMA C layout.
//   * W_hh (k-major [D][4D]) is streamed from L2 straight into the B-operand registers, prefetched one k-step
//     ahead; each wave reads a disjoint column set, so LDS staging would add nothing.
//   * Variable lengths: a sequence is updated while step < len; the reverse direction reads token len-1-step, so
//     its final state is the one after consuming token 0 -- the same states pack_padded_sequence produces.
// T2P_LSTM_ABL (development only, results are wrong): 1 = no weight stream (the first fragments are reused), 2 = no gate-table
// gather, 4 = no gate functions
#ifndef T2P_LSTM_ABL
#define T2P_LSTM_ABL 0
#endif
#ifndef T2P_LSTM_RING
#define T2P_LSTM_RING 4
#endif
#include "t2p_common.h"

namespace t2p {
namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <int D>
__global__ __launch_bounds__(256, 1) void k_bilstm(const float* __restrict__ gate_table, const float* __restrict__ whh,
                                                    const int32_t* __restrict__ tokens,
                                                    const int32_t* __restrict__ lengths, int B, int T, int V,
                                                    float* __restrict__ hout /*[2][B][D]*/) {
    constexpr int G4 = 4 * D;
    constexpr int UT = D / 128;  // 32-unit tiles per wave (D/4 units per wave)
    constexpr int KS = D / 2;
    constexpr int LDH = D + 4;
    __shared__ __attribute__((aligned(16))) float h_lds[32 * LDH];
    __shared__ int len_lds[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int dir = blockIdx.y;
    const int row0 = blockIdx.x * 32;
    const float* gt = gate_table + (int64_t)dir * V * G4;
    const float* wd = whh + (int64_t)dir * D * G4;

    for (int i = tid; i < 32 * LDH; i += 256) h_lds[i] = 0.f;
    if (tid < 32) len_lds[tid] = (row0 + tid < B) ? lengths[row0 + tid] : 0;
    __syncthreads();
    int max_len = 0;
    for (int i = 0; i < 32; i++) max_len = len_lds[i] > max_len ? len_lds[i] : max_len;

    // rows of this lane in the 32x32 C layout: (e&3) + 8*(e>>2) + 4*h
    int my_len[16];
#pragma unroll
    for (int e = 0; e < 16; e++) my_len[e] = len_lds[(e & 3) + 8 * (e >> 2) + 4 * h];

    float c[UT][16];
    float hreg[UT][16];
#pragma unroll
    for (int u = 0; u < UT; u++)
#pragma unroll
        for (int e = 0; e < 16; e++) { c[u][e] = 0.f; hreg[u][e] = 0.f; }

    const int unit0 = wave * (D / 4);

    for (int step = 0; step < max_len; step++) {
        // ---- gates = table[token] (+ biases folded in the table) ------------------------------------------------
        f32x16 acc[4][UT];
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * h;
            const int len = my_len[e];
            const bool active = step < len;
            int tok = 0;
            if (active) {
                const int t = dir == 0 ? step : (len - 1 - step);
                tok = tokens[(int64_t)(row0 + r) * T + t];
            }
            const float* trow = gt + (int64_t)tok * G4;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int u = 0; u < UT; u++) acc[q][u][e] = active ? trow[q * D + unit0 + u * 32 + l31] : 0.f;
        }
        // ---- gates += h_{t-1} W_hh^T : [32 x D] x [D x 4D-slice] ----------------------------------------------
        const float* hrow = h_lds + l31 * LDH + h * KS;
        const float* wcol = wd + (int64_t)(h * KS) * G4 + unit0 + l31;
        float bcur[4][UT], bnext[4][UT];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int u = 0; u < UT; u++) bcur[q][u] = wcol[q * D + u * 32];
#pragma unroll 2
        for (int s4 = 0; s4 < KS / 4; s4++) {
            const f32x4 a = *(const f32x4*)(hrow + s4 * 4);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int s = s4 * 4 + j;
                const int sn = (s + 1 < KS) ? s + 1 : s;
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int u = 0; u < UT; u++) bnext[q][u] = wcol[(int64_t)sn * G4 + q * D + u * 32];
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int u = 0; u < UT; u++)
                        acc[q][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bcur[q][u], acc[q][u], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int u = 0; u < UT; u++) bcur[q][u] = bnext[q][u];
            }
        }
        __syncthreads();  // every wave has finished reading h_{t-1}
        // ---- cell update (PyTorch gate order i, f, g, o) --------------------------------------------------------
#pragma unroll
        for (int u = 0; u < UT; u++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * h;
                if (step < my_len[e]) {
                    const float ig = sigmoidf_(acc[0][u][e]);
                    const float fg = sigmoidf_(acc[1][u][e]);
                    const float gg = tanhf(acc[2][u][e]);
                    const float og = sigmoidf_(acc[3][u][e]);
                    const float cn = fg * c[u][e] + ig * gg;
                    c[u][e] = cn;
                    const float hn = og * tanhf(cn);
                    hreg[u][e] = hn;
                    h_lds[r * LDH + unit0 + u * 32 + l31] = hn;
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < UT; u++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int r = row0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (r < B) hout[((int64_t)dir * B + r) * D + unit0 + u * 32 + l31] = hreg[u][e];
        }
}

// ---- the same recurrence on the f16x3 path -------------------------------------------------------------------------------
// The fp32 kernel above spends a time step in 1,024 v_mfma_f32_32x32x2_f32 per wave (64 cycles each: 27 us) - and a call's
// latency is max_len steps whatever the batch, because a sequence's steps cannot overlap.  Here the recurrent product runs as
// three v_mfma_f32_32x32x16_f16 per 16 k (W_hh and h split hi + lo in fp16, fp32 accumulation: fp32-class error, as in the
// cell branch): 384 MFMAs of 32 cycles per wave and step.  W_hh arrives as the host-packed register image of
// packing.py::pack_f16x3_scaled (w' = s w, s a power of two; hi = fp16(w'), lo = fp16(w' - hi); one accumulator for
// hi.hi + hi.lo + lo.hi that starts at the gate-table row - the launcher scales the table by s - and is drained as acc / s),
// streamed from L2 with 16-byte loads (1 MB per step and workgroup, as the fp32 kernel).  h_{t-1} lives in LDS as two fp16
// planes.  The product runs tile by tile (gate x 32 units), so that a tile's MFMAs wait for their own 16 table values only:
// measured on the first form of this kernel (k-step-major, all 128 table values in front of the first MFMA), the gate-table
// gather was the largest single cost of a time step (9 of 28 us; the weight stream 6, the gate functions 2).
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// Gate functions on the hardware exponential / reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each; absolute error of the gate
// values <= 2e-7).  The library expf / tanhf of the fp32 kernel cost ~35 instructions per value: 160 values per lane and time
// step made the cell update as long as the recurrent product.
__device__ __forceinline__ float fast_sigmoid(float x) {
    if constexpr (T2P_LSTM_ABL & 4) return x;
    return __builtin_amdgcn_rcpf(1.f + __expf(-x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    if constexpr (T2P_LSTM_ABL & 4) return x;
    const float t = __expf(-2.f * fabsf(x));            // in (0, 1]: no overflow; 1 - t is exact-ish near t = 1 (small |x|)
    return copysignf((1.f - t) * __builtin_amdgcn_rcpf(1.f + t), x);
}

template <int D>
__global__ __launch_bounds__(256, 1) void k_bilstm_x3(const float* __restrict__ gate_table, const uint4* __restrict__ whh_x3,
                                                       float scale, const int32_t* __restrict__ tokens,
                                                       const int32_t* __restrict__ lengths, int B, int T, int V,
                                                       float* __restrict__ hout /*[2][B][D]*/) {
    constexpr int G4 = 4 * D;
    constexpr int UT = D / 128;      // 32-unit tiles per wave and gate
    constexpr int S16 = D / 16;      // k-steps of 16
    constexpr int LDHH = D + 8;      // halves; 16-byte pad keeps ds_read_b128 conflict-free
    constexpr int PLANE_U4 = (G4 / 32) * S16 * 64;   // uint4 per plane of one direction's image
    __shared__ __attribute__((aligned(16))) _Float16 h_hi[32 * LDHH];
    __shared__ __attribute__((aligned(16))) _Float16 h_lo[32 * LDHH];
    __shared__ int len_lds[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, l31 = lane & 31;
    const int dir = blockIdx.y;
    const float* gt = gate_table + (int64_t)dir * V * G4;
    const uint4* wd = whh_x3 + (int64_t)dir * 2 * PLANE_U4;
    const float inv_scale = 1.f / scale;
    // D = 256: one direction's image is 1 MB, four times the registers of a workgroup: it is streamed again every time step,
    // and the compiler must not hoist the (loop-invariant) loads of all 16 k-steps out of the time loop - the pointer is made
    // opaque once per step.  D = 128: the 256 KB image fits (64 fragments of 4 registers per lane), the loads ARE hoisted
    // and the recurrence runs from registers.
    constexpr bool STREAM = D > 128;

    const int unit0 = wave * (D / 4);
    // image tile of (gate q, unit tile u) of this wave: columns q D + unit0 + 32 u ...; uint4 index of (tile, step s, half h, lane)
    auto widx = [&](int q, int u, int s) { return (((q * (D / 32) + wave * UT + u) * S16 + s) * 2 + h) * 32 + l31; };
    uint4 wn_hi[4][UT], wn_lo[4][UT];   // (dummies of kstep's signature)
    auto load_w = [&](int s, uint4 (&whi)[4][UT], uint4 (&wlo)[4][UT]) {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int u = 0; u < UT; u++) {
                const int i = widx(q, u, s);
                whi[q][u] = wd[i];
                wlo[q][u] = wd[PLANE_U4 + i];
            }
    };
    // D = 128: the whole image of this wave (S16 x 4 tiles x hi, lo = 64 fragments = 256 registers) is loaded once
    uint4 wr_hi[STREAM ? 1 : S16][4][UT], wr_lo[STREAM ? 1 : S16][4][UT];
    // D = 256: tile-major stream through a ring of RING fragment pairs: item i = (tile, k-step) sits in slot i % RING and the
    // slot is refilled with item i + RING as soon as its MFMAs are issued (static register indices).  Measured per time step:
    // RING 4 / 8 / 16 = 17.2 / 17.5 / 19.3 us (8 and 16 spill): the stream runs at the ~60 GB/s one CU draws from L2, not at
    // its latency.  Tile t = (gate t & 3, unit tile t >> 2).
    constexpr int NTILE = 4 * UT;
    constexpr int RING = T2P_LSTM_RING;   // fragment pairs in flight (<= S16, a divisor of it): item i = (tile, k-step) sits in slot i % RING
    uint4 rg_hi[STREAM ? RING : 1], rg_lo[STREAM ? RING : 1];
    // (a tile's S16 fragments are contiguous: 64 uint4 per k-step.  The tile's base travels in SGPRs and is made opaque, so
    // that the 2 x 128 fragment addresses of a time step are formed where they are used instead of being hoisted into
    // 500 registers)
    const int lane_u4 = h * 32 + l31;
    auto tile_base = [&](int t) {
        const uint4* b = wd + (((t & 3) * (D / 32) + wave * UT + (t >> 2)) * S16) * 64;
        asm volatile("" : "+s"(b));
        return b;
    };
    if constexpr (STREAM) {
        const uint4* b0 = tile_base(0);
#pragma unroll
        for (int s = 0; s < RING; s++) {
            rg_hi[s] = b0[s * 64 + lane_u4];
            rg_lo[s] = b0[PLANE_U4 + s * 64 + lane_u4];
        }
    } else {
#pragma unroll
        for (int s = 0; s < S16; s++) load_w(s, wr_hi[s], wr_lo[s]);
    }

    // persistent over the groups of 32 sequences (grid.x <= groups): with many short sequences (the fine stage encodes 60,000
    // hint sentences of ~9 tokens) a workgroup per group would spend its life fetching the weight image
    const int n_groups = (B + 31) / 32;
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int row0 = grp * 32;
    __syncthreads();   // (the previous group's last reads of the planes / of len_lds)
    for (int i = tid; i < 32 * LDHH; i += 256) { h_hi[i] = (_Float16)0.f; h_lo[i] = (_Float16)0.f; }
    if (tid < 32) len_lds[tid] = (row0 + tid < B) ? lengths[row0 + tid] : 0;
    __syncthreads();
    int max_len = 0;
    for (int i = 0; i < 32; i++) max_len = len_lds[i] > max_len ? len_lds[i] : max_len;
    int my_len[16];
#pragma unroll
    for (int e = 0; e < 16; e++) my_len[e] = len_lds[(e & 3) + 8 * (e >> 2) + 4 * h];

    float c[UT][16], hreg[UT][16];
#pragma unroll
    for (int u = 0; u < UT; u++)
#pragma unroll
        for (int e = 0; e < 16; e++) { c[u][e] = 0.f; hreg[u][e] = 0.f; }

    // the token of every row of this lane, fetched one time step ahead (token -> table row -> accumulator is a chain of two
    // dependent loads in front of every step's MFMAs otherwise)
    int tok_next[16];
    auto fetch_tokens = [&](int step) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * h;
            const int len = my_len[e];
            tok_next[e] = V - 1;   // the table's zero row: a row past its length adds nothing (and is not updated)
            if (step < len) tok_next[e] = tokens[(int64_t)(row0 + r) * T + (dir == 0 ? step : (len - 1 - step))];
        }
    };
    fetch_tokens(0);

    for (int step = 0; step < max_len; step++) {
        // accumulators start at the (pre-scaled) gate-table rows: plain loads, issued tile by tile in the order the tiles are
        // multiplied, so that a tile's MFMAs wait for its own 16 values only
        f32x16 acc[4][UT];
#pragma unroll
        for (int t = 0; t < NTILE; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float* trow = gt + (int64_t)tok_next[e] * G4;
                acc[t & 3][t >> 2][e] = (T2P_LSTM_ABL & 2) ? 0.f : trow[(t & 3) * D + unit0 + (t >> 2) * 32 + l31];
            }
        fetch_tokens(step + 1);
        const _Float16* hr_hi = h_hi + l31 * LDHH + h * (D / 2);
        const _Float16* hr_lo = h_lo + l31 * LDHH + h * (D / 2);
        // one k-step: fragments `cur` feed the MFMAs while those of the following k-step land in `nxt` (the first of the next
        // time step behind the last one: the image does not change)
        auto kstep = [&](int s, uint4 (&chi)[4][UT], uint4 (&clo)[4][UT], uint4 (&nhi)[4][UT], uint4 (&nlo)[4][UT]) {
            const half8 a_hi = *(const half8*)(hr_hi + 8 * s), a_lo = *(const half8*)(hr_lo + 8 * s);
            if constexpr (STREAM && !(T2P_LSTM_ABL & 1)) load_w(s + 1 < S16 ? s + 1 : 0, nhi, nlo);
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int u = 0; u < UT; u++) {
                    const half8 bh = __builtin_bit_cast(half8, chi[q][u]), bl = __builtin_bit_cast(half8, clo[q][u]);
                    acc[q][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, bh, acc[q][u], 0, 0, 0);
                    acc[q][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, bl, acc[q][u], 0, 0, 0);
                    acc[q][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, bh, acc[q][u], 0, 0, 0);
                }
        };
        if constexpr (STREAM) {
            half8 a_hi = *(const half8*)(hr_hi), a_lo = *(const half8*)(hr_lo), n_hi = a_hi, n_lo = a_lo;
#pragma unroll
            for (int t = 0; t < NTILE; t++) {
                const uint4* cb = tile_base(t);
                const uint4* nb = tile_base((t + 1) % NTILE);
#pragma unroll
                for (int s = 0; s < S16; s++) {
                    const int sn = (s + 1) % S16;
                    n_hi = *(const half8*)(hr_hi + 8 * sn);
                    n_lo = *(const half8*)(hr_lo + 8 * sn);
                    const half8 bh = __builtin_bit_cast(half8, rg_hi[s % RING]), bl = __builtin_bit_cast(half8, rg_lo[s % RING]);
                    acc[t & 3][t >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, bh, acc[t & 3][t >> 2], 0, 0, 0);
                    acc[t & 3][t >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, bl, acc[t & 3][t >> 2], 0, 0, 0);
                    acc[t & 3][t >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, bh, acc[t & 3][t >> 2], 0, 0, 0);
                    if constexpr (!(T2P_LSTM_ABL & 1)) {   // (the tile behind the last one is the first of the next time step)
                        // item RING ahead: (t, s + RING), or (t + 1, s + RING - S16)
                        const uint4* rb = s + RING < S16 ? cb : nb;
                        const int rs = (s + RING) % S16;
                        rg_hi[s % RING] = rb[rs * 64 + lane_u4];
                        rg_lo[s % RING] = rb[PLANE_U4 + rs * 64 + lane_u4];
                    }
                    a_hi = n_hi;
                    a_lo = n_lo;
                    __builtin_amdgcn_sched_barrier(0);   // keeps the refills where they are (hoisted, 256 loads would spill)
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < S16; s++) kstep(s, wr_hi[s], wr_lo[s], wn_hi, wn_lo);
        }
        __syncthreads();  // every wave has finished reading h_{t-1}
#pragma unroll
        for (int u = 0; u < UT; u++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * h;
                if (step < my_len[e]) {
                    const float ig = fast_sigmoid(acc[0][u][e] * inv_scale);
                    const float fg = fast_sigmoid(acc[1][u][e] * inv_scale);
                    const float gg = fast_tanh(acc[2][u][e] * inv_scale);
                    const float og = fast_sigmoid(acc[3][u][e] * inv_scale);
                    const float cn = fg * c[u][e] + ig * gg;
                    c[u][e] = cn;
                    const float hn = og * fast_tanh(cn);
                    hreg[u][e] = hn;
                    const _Float16 hh = (_Float16)hn;
                    h_hi[r * LDHH + unit0 + u * 32 + l31] = hh;
                    h_lo[r * LDHH + unit0 + u * 32 + l31] = (_Float16)(hn - (float)hh);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < UT; u++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int r = row0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (r < B) hout[((int64_t)dir * B + r) * D + unit0 + u * 32 + l31] = hreg[u][e];
        }
    }   // groups
}

__global__ void k_scale_inplace(float* __restrict__ x, int64_t n, float s) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= s;
}

__global__ void k_mean2(const float* __restrict__ hdir, int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (hdir[i] + hdir[n + i]) / 2.f;
}

}  // namespace

// hdir_ws: [2][B][D] scratch.
int launch_bilstm_impl(const float* gate_table, const float* whh, const void* whh_x3, float whh_scale, const int32_t* tokens,
                       const int32_t* lengths, int B, int T, int V, int D, float* hdir_ws, float* out, hipStream_t st) {
    if (B == 0) return 0;
    dim3 grid((unsigned)((B + 31) / 32), 2);
    if (whh_x3 != nullptr) {   // the f16x3 kernel's accumulators start at scale x table row: scale the table once (a power of two)
        const int64_t n = (int64_t)2 * V * 4 * D;
        hipLaunchKernelGGL(k_scale_inplace, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, (float*)gate_table, n, whh_scale);
        T2P_CHECK_LAUNCH("bilstm_table_scale");
    }
    {
    ProfScope ps_(whh_x3 ? "bilstm_x3" : "bilstm", st);
    if (whh_x3 != nullptr) {
        const unsigned per_dir = (unsigned)(num_cus() / 2 > 0 ? num_cus() / 2 : 1);   // one workgroup per CU, two directions
        if (grid.x > per_dir) grid.x = per_dir;
        T2P_CHECK_ARG(((uintptr_t)whh_x3 & 15) == 0 && whh_scale > 0.f, "bilstm: the f16x3 image must be 16-byte aligned, its scale > 0");
        if (D == 256) {
            hipLaunchKernelGGL(k_bilstm_x3<256>, grid, dim3(256), 0, st, gate_table, (const uint4*)whh_x3, whh_scale, tokens, lengths,
                               B, T, V, hdir_ws);
        } else if (D == 128) {
            hipLaunchKernelGGL(k_bilstm_x3<128>, grid, dim3(256), 0, st, gate_table, (const uint4*)whh_x3, whh_scale, tokens, lengths,
                               B, T, V, hdir_ws);
        } else if (D == 384) {   // (embed_dim 300 zero-padded by the host: a padded hidden unit stays exactly 0)
            hipLaunchKernelGGL(k_bilstm_x3<384>, grid, dim3(256), 0, st, gate_table, (const uint4*)whh_x3, whh_scale, tokens, lengths,
                               B, T, V, hdir_ws);
        } else {
            set_error("bilstm: embed_dim=%d not instantiated (128, 256, 384)", D);
            return T2P_E_UNSUPPORTED;
        }
    } else if (D == 256) {
        hipLaunchKernelGGL(k_bilstm<256>, grid, dim3(256), 0, st, gate_table, whh, tokens, lengths, B, T, V, hdir_ws);
    } else if (D == 128) {
        hipLaunchKernelGGL(k_bilstm<128>, grid, dim3(256), 0, st, gate_table, whh, tokens, lengths, B, T, V, hdir_ws);
    } else if (D == 384) {
        hipLaunchKernelGGL(k_bilstm<384>, grid, dim3(256), 0, st, gate_table, whh, tokens, lengths, B, T, V, hdir_ws);
    } else {
        set_error("bilstm: embed_dim=%d not instantiated (128, 256, 384)", D);
        return T2P_E_UNSUPPORTED;
    }
    }
    T2P_CHECK_LAUNCH("bilstm");
    const int64_t n = (int64_t)B * D;
    hipLaunchKernelGGL(k_mean2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hdir_ws, n, out);
    T2P_CHECK_LAUNCH("bilstm_mean");
    return 0;
}

// ---- training-mode recurrence, step by step (text branch) ---------------------------------------------
// The persistent kernel above keeps no activations.  For training (training/coarse.py:44: anchor = model.encode_text(...),
// loss.backward()) the recurrence runs one step per launch - pre-activations of the recurrent term from the tiled GEMM, this
// cell kernel - and stores what the backward pass needs: the gate activations, the cell and the hidden state of every step.
// Same semantics as above: a sequence is updated while step < len; the reverse direction reads token len-1-step.
namespace {

// one thread = one hidden unit of one sequence
__global__ void k_lstm_cell_fwd(const float* __restrict__ pre /*[B][4D] h_{s-1} W_hh*/, const float* __restrict__ table /*[V][4D]*/,
                                const int32_t* __restrict__ tokens, const int32_t* __restrict__ lengths, int64_t B, int T,
                                int D, int step, int reverse, const float* __restrict__ c_prev,
                                const float* __restrict__ h_prev, float* __restrict__ gates /*[B][4D] i f g o*/,
                                float* __restrict__ c, float* __restrict__ h) {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    const int64_t b = idx / D;
    const int u = (int)(idx % D);
    const int len = lengths[b];
    float* g = gates + b * 4 * D;
    if (step >= len) {  // past the end of this sequence: the state is carried, the step contributes no gradient
        c[idx] = c_prev[idx];
        h[idx] = h_prev[idx];
        g[u] = g[D + u] = g[2 * D + u] = g[3 * D + u] = 0.f;
        return;
    }
    const int tok = tokens[b * T + (reverse ? len - 1 - step : step)];
    const float* tr = table + (int64_t)tok * 4 * D;
    const float* pr = pre + b * 4 * D;
    const float gi = sigmoidf_(pr[u] + tr[u]), gf = sigmoidf_(pr[D + u] + tr[D + u]);
    const float gg = tanhf(pr[2 * D + u] + tr[2 * D + u]), go = sigmoidf_(pr[3 * D + u] + tr[3 * D + u]);
    const float cn = gf * c_prev[idx] + gi * gg;
    c[idx] = cn;
    h[idx] = go * tanhf(cn);
    g[u] = gi;
    g[D + u] = gf;
    g[2 * D + u] = gg;
    g[3 * D + u] = go;
}

// dh = dh_gemm (= d_pre of the later step x W_hh^T; NULL at the last step) + dh_carry_in; writes the gradient of this step's
// gate pre-activations, the cell gradient of the previous step and the part of dh that bypasses the step (finished sequences)
__global__ void k_lstm_cell_bwd(const float* __restrict__ dh_gemm, const float* __restrict__ dh_carry_in,
                                const float* __restrict__ dc_in, const float* __restrict__ gates,
                                const float* __restrict__ c_prev, const float* __restrict__ c,
                                const int32_t* __restrict__ lengths, int64_t B, int D, int step, float* __restrict__ d_pre,
                                float* __restrict__ dc_out, float* __restrict__ dh_carry_out) {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    const int64_t b = idx / D;
    const int u = (int)(idx % D);
    const float dh = (dh_gemm ? dh_gemm[idx] : 0.f) + dh_carry_in[idx];
    const float dc = dc_in[idx];
    float* dp = d_pre + b * 4 * D;
    if (step >= lengths[b]) {
        dp[u] = dp[D + u] = dp[2 * D + u] = dp[3 * D + u] = 0.f;
        dc_out[idx] = dc;
        dh_carry_out[idx] = dh;
        return;
    }
    const float* g = gates + b * 4 * D;
    const float gi = g[u], gf = g[D + u], gg = g[2 * D + u], go = g[3 * D + u];
    const float tc = tanhf(c[idx]);
    const float dct = dc + dh * go * (1.f - tc * tc);
    dp[u] = dct * gg * gi * (1.f - gi);
    dp[D + u] = dct * c_prev[idx] * gf * (1.f - gf);
    dp[2 * D + u] = dct * gi * (1.f - gg * gg);
    dp[3 * D + u] = dh * tc * go * (1.f - go);
    dc_out[idx] = dct * gf;
    dh_carry_out[idx] = 0.f;
}

}  // namespace

int launch_lstm_cell_fwd(const float* pre, const float* table, const int32_t* tokens, const int32_t* lengths, int64_t B, int T,
                         int D, int step, int reverse, const float* c_prev, const float* h_prev, float* gates, float* c,
                         float* h, hipStream_t st) {
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_lstm_cell_fwd, dim3((unsigned)((B * D + 255) / 256)), dim3(256), 0, st, pre, table, tokens, lengths, B,
                       T, D, step, reverse, c_prev, h_prev, gates, c, h);
    T2P_CHECK_LAUNCH("lstm_cell_fwd");
    return 0;
}

int launch_lstm_cell_bwd(const float* dh_gemm, const float* dh_carry_in, const float* dc_in, const float* gates,
                         const float* c_prev, const float* c, const int32_t* lengths, int64_t B, int D, int step, float* d_pre,
                         float* dc_out, float* dh_carry_out, hipStream_t st) {
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_lstm_cell_bwd, dim3((unsigned)((B * D + 255) / 256)), dim3(256), 0, st, dh_gemm, dh_carry_in, dc_in,
                       gates, c_prev, c, lengths, B, D, step, d_pre, dc_out, dh_carry_out);
    T2P_CHECK_LAUNCH("lstm_cell_bwd");
    return 0;
}

}  // namespace t2p
