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

__global__ void k_mean2(const float* __restrict__ hdir, int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (hdir[i] + hdir[n + i]) / 2.f;
}

}  // namespace

// hdir_ws: [2][B][D] scratch.
int launch_bilstm_impl(const float* gate_table, const float* whh, const int32_t* tokens, const int32_t* lengths, int B,
                       int T, int V, int D, float* hdir_ws, float* out, hipStream_t st) {
    if (B == 0) return 0;
    dim3 grid((unsigned)((B + 31) / 32), 2);
    {
    ProfScope ps_("bilstm", st);
    if (D == 256) {
        hipLaunchKernelGGL(k_bilstm<256>, grid, dim3(256), 0, st, gate_table, whh, tokens, lengths, B, T, V, hdir_ws);
    } else if (D == 128) {
        hipLaunchKernelGGL(k_bilstm<128>, grid, dim3(256), 0, st, gate_table, whh, tokens, lengths, B, T, V, hdir_ws);
    } else {
        set_error("bilstm: embed_dim=%d not instantiated (128, 256)", D);
        return T2P_E_UNSUPPORTED;
    }
    }
    T2P_CHECK_LAUNCH("bilstm");
    const int64_t n = (int64_t)B * D;
    hipLaunchKernelGGL(k_mean2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hdir_ws, n, out);
    T2P_CHECK_LAUNCH("bilstm_mean");
    return 0;
}

}  // namespace t2p
