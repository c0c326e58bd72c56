// EdgewiseMSA low-rank core -- generic multi-kernel path (any N / dk / V<=8 / r<=8).
//
// Every N x N map lives in fp32 in the caller's `saved` / `workspace` buffers and
// every contraction is a bgemm (bgemm.h), so this path is HBM-bound by design: it
// is the fp32-exact path (MOPK_PREC_FP32) and the any-shape fallback.  The NS hot
// shape (N<=224, dk multiple of 16) runs edgewise_fused.hip.
//
// Math follows reference mop/models/attention_variants.py:500-562; the gate-head
// only needs row/col MEANS of the feature stack (:323-324), so the (BH,2V+2,N,N)
// stack of :534 is never built.  Backward is the hand-derived gradient checked in
// oracle/edgewise.py::core_bwd against reference autograd.
#include "bgemm.h"
#include "common.h"

namespace mopk {

constexpr int MAXV = 8, MAXR = 8;
constexpr float EPS_CHAIN = 1e-6f;  // attention_variants.py:516

struct EwDims {
    int B, H, N, dk, V, r, Vk, C, LD;
    int64_t BH;
};
static EwDims ew_dims(const MopkEdgewiseArgs *a) {
    EwDims d;
    d.B = a->B; d.H = a->H; d.N = a->N; d.dk = a->dk; d.V = a->V; d.r = a->r;
    d.Vk = a->k.sv == 0 ? 1 : a->V;
    d.C = 2 * a->V + 2;
    d.LD = (int)round_up(a->N, 4);
    d.BH = (int64_t)a->B * a->H;
    return d;
}

struct EwSaved {
    float *Qe, *Kc, *V0, *VL;     // (V,BH,N,dk) (Vk,BH,N,dk) (BH,N,dk) x2
    float *S, *A;                 // (V,BH,N,LD)
    float *T, *U;                 // (V-1,BH,N,LD) prefix products T[m] (m>=1), U[m]
    float *Cr, *Cl;               // (BH,N,LD)
    float *rS, *cS;               // (V,BH,N)
    float *rCr, *cCr, *rCl, *cCl; // (BH,N)
    float *ga, *gb;               // (BH,4r,N)
    float *P;                     // (BH,N,LD)
    float *ychain;                // (BH,N,dk)
    float *wsig;                  // 1
};
static EwSaved ew_carve_saved(void *p, const EwDims &d, size_t *total) {
    Carver c(p);
    EwSaved s;
    const size_t nd = d.BH * d.N * d.dk, nn = d.BH * (size_t)d.N * d.LD, n1 = d.BH * d.N;
    s.Qe = c.take<float>(d.V * nd); s.Kc = c.take<float>(d.Vk * nd);
    s.V0 = c.take<float>(nd); s.VL = c.take<float>(nd);
    s.S = c.take<float>(d.V * nn); s.A = c.take<float>(d.V * nn);
    s.T = c.take<float>((d.V - 1) * nn); s.U = c.take<float>((d.V - 1) * nn);
    s.Cr = c.take<float>(nn); s.Cl = c.take<float>(nn);
    s.rS = c.take<float>(d.V * n1); s.cS = c.take<float>(d.V * n1);
    s.rCr = c.take<float>(n1); s.cCr = c.take<float>(n1); s.rCl = c.take<float>(n1); s.cCl = c.take<float>(n1);
    s.ga = c.take<float>(d.BH * 4 * d.r * d.N); s.gb = c.take<float>(d.BH * 4 * d.r * d.N);
    s.P = c.take<float>(nn); s.ychain = c.take<float>(nd); s.wsig = c.take<float>(64);
    if (total) *total = c.off;
    return s;
}
struct EwWork {
    float *ybase;                  // fwd: (BH,N,dk)
    float *dyc;                    // (BH,N,dk)
    float *dP;                     // (BH,N,LD)  dP then dSmix in place
    float *dCf, *dCb, *D0, *D1;    // (BH,N,LD)
    float *dA;                     // (V,BH,N,LD) dA then dS in place
    float *da, *db;                // (BH,4r,N)
    float *drS, *dcS;              // (V,BH,N)
    float *drCr, *dcCr, *drCl, *dcCl;
    float *dQe, *dKc, *dV0, *dVL;
};
static EwWork ew_carve_work(void *p, const EwDims &d, size_t *total) {
    Carver c(p);
    EwWork w;
    const size_t nd = d.BH * d.N * d.dk, nn = d.BH * (size_t)d.N * d.LD, n1 = d.BH * d.N;
    w.ybase = c.take<float>(nd); w.dyc = c.take<float>(nd);
    w.dP = c.take<float>(nn); w.dCf = c.take<float>(nn); w.dCb = c.take<float>(nn);
    w.D0 = c.take<float>(nn); w.D1 = c.take<float>(nn);
    w.dA = c.take<float>(d.V * nn);
    w.da = c.take<float>(d.BH * 4 * d.r * d.N); w.db = c.take<float>(d.BH * 4 * d.r * d.N);
    w.drS = c.take<float>(d.V * n1); w.dcS = c.take<float>(d.V * n1);
    w.drCr = c.take<float>(n1); w.dcCr = c.take<float>(n1); w.drCl = c.take<float>(n1); w.dcCl = c.take<float>(n1);
    w.dQe = c.take<float>(d.V * nd); w.dKc = c.take<float>(d.Vk * nd);
    w.dV0 = c.take<float>(nd); w.dVL = c.take<float>(nd);
    if (total) *total = c.off;
    return w;
}

size_t ew_generic_saved_bytes(const MopkEdgewiseArgs *a) {
    size_t t; ew_carve_saved(nullptr, ew_dims(a), &t); return t;
}
size_t ew_generic_workspace_bytes(const MopkEdgewiseArgs *a) {
    size_t t; ew_carve_work(nullptr, ew_dims(a), &t); return t;
}

// ------------------------------------------------------------------ kernels
template <typename T>
__global__ void ew_prep_kernel(MopkEdgewiseArgs a, EwDims d, EwSaved s) {
    const int64_t total = d.BH * d.N * d.dk;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) s.wsig[0] = sigmoidf_(*a.chain_logit);
    if (idx >= total) return;
    const int dd = idx % d.dk;
    const int n = (idx / d.dk) % d.N;
    const int64_t bh = idx / ((int64_t)d.dk * d.N);
    const int b = bh / d.H, h = bh % d.H;
    for (int v = 0; v < d.V; ++v) {
        const T *q = (const T *)a.q.ptr + v * a.q.sv + b * a.q.sb + h * a.q.sh + n * a.q.sn + dd;
        s.Qe[v * total + idx] = ld_as_f32(q) * a.sqk[(v * d.H + h) * d.dk + dd];
    }
    for (int v = 0; v < d.Vk; ++v) {
        const T *k = (const T *)a.k.ptr + v * a.k.sv + b * a.k.sb + h * a.k.sh + n * a.k.sn + dd;
        s.Kc[v * total + idx] = ld_as_f32(k);
    }
    const T *p0 = (const T *)a.v0.ptr + b * a.v0.sb + h * a.v0.sh + n * a.v0.sn + dd;
    const T *pL = (const T *)a.vL.ptr + b * a.vL.sb + h * a.vL.sh + n * a.vL.sn + dd;
    s.V0[idx] = ld_as_f32(p0) * a.vs0[h * d.dk + dd];
    s.VL[idx] = ld_as_f32(pL) * a.vsL[h * d.dk + dd];
}

// one wave per row: out = softmax(in), rowmean(in)
__global__ void softmax_rows_kernel(const float *in, float *out, float *rowmean, int64_t rows, int N, int LD) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float *p = in + row * LD;
    float mx = -INFINITY, sm = 0.f;
    for (int j = lane; j < N; j += 64) { float v = p[j]; mx = fmaxf(mx, v); sm += v; }
    mx = wave_max(mx); sm = wave_sum(sm);
    float den = 0.f;
    for (int j = lane; j < N; j += 64) den += expf(p[j] - mx);
    den = wave_sum(den);
    const float inv = 1.f / den;
    float *o = out + row * LD;
    for (int j = lane; j < N; j += 64) o[j] = expf(p[j] - mx) * inv;
    if (lane == 0) rowmean[row] = sm / N;
}
// thread per column
__global__ void colmean_kernel(const float *in, float *colmean, int N, int LD) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const float *p = in + (int64_t)blockIdx.y * N * LD + j;
    float sm = 0.f;
    for (int i = 0; i < N; ++i) sm += p[(int64_t)i * LD];
    colmean[(int64_t)blockIdx.y * N + j] = sm / N;
}
__global__ void log_rows_kernel(const float *in, float *out, float *rowmean, int64_t rows, int N, int LD) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float sm = 0.f;
    for (int j = lane; j < N; j += 64) { float v = logf(in[row * LD + j] + EPS_CHAIN); out[row * LD + j] = v; sm += v; }
    sm = wave_sum(sm);
    if (lane == 0) rowmean[row] = sm / N;
}

__device__ __forceinline__ float row_feat(const EwSaved &s, const EwDims &d, int c, int64_t bh, int n, bool col) {
    const int64_t n1 = d.BH * d.N, o = bh * d.N + n;
    if (c < d.V) return (col ? s.cS : s.rS)[c * n1 + o];
    if (c < 2 * d.V) return (col ? s.rS : s.cS)[(c - d.V) * n1 + o];
    if (c == 2 * d.V) return (col ? s.cCr : s.rCr)[o];
    return (col ? s.cCl : s.rCl)[o];
}
// a = Wr row_feat + br ; b = Wc col_feat + bc     (attention_variants.py:323-326)
__global__ void gate_ab_kernel(MopkEdgewiseArgs a, EwDims d, EwSaved s) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.BH * d.N) return;
    const int n = idx % d.N;
    const int64_t bh = idx / d.N;
    float fr[2 * MAXV + 2], fc[2 * MAXV + 2];
    for (int c = 0; c < d.C; ++c) { fr[c] = row_feat(s, d, c, bh, n, false); fc[c] = row_feat(s, d, c, bh, n, true); }
    for (int o = 0; o < 4 * d.r; ++o) {
        float sa = a.br[o], sb = a.bc[o];
        for (int c = 0; c < d.C; ++c) { sa = fmaf(a.Wr[o * d.C + c], fr[c], sa); sb = fmaf(a.Wc[o * d.C + c], fc[c], sb); }
        s.ga[(bh * 4 * d.r + o) * d.N + n] = sa;
        s.gb[(bh * 4 * d.r + o) * d.N + n] = sb;
    }
}

struct RowGates { float a[4][MAXR]; };
__device__ __forceinline__ void load_row_gates(RowGates &g, const float *ga_bh, int r, int N, int i) {
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < MAXR; ++k) g.a[q][k] = k < r ? ga_bh[(q * r + k) * N + i] : 0.f;
}
__device__ __forceinline__ void gates_at(float G[4], const RowGates &g, const float *gb_bh, int r, int N, int j) {
    for (int q = 0; q < 4; ++q) {
        float z = 0.f;
        for (int k = 0; k < r; ++k) z = fmaf(g.a[q][k], gb_bh[(q * r + k) * N + j], z);
        G[q] = sigmoidf_(z);
    }
}
// per-element view statistics: S0, O = sum_{v>=1} S_v, lse over views
__device__ __forceinline__ void view_stats(const float *S, int64_t vstride, int64_t off, int V, float sv[MAXV],
                                           float &O, float &lse) {
    float mx = -INFINITY;
    for (int v = 0; v < V; ++v) { sv[v] = S[v * vstride + off]; mx = fmaxf(mx, sv[v]); }
    float e = 0.f; O = 0.f;
    for (int v = 0; v < V; ++v) { e += expf(sv[v] - mx); if (v) O += sv[v]; }
    lse = mx + logf(e);
}

// Smix (:543-547) + softmax (:551); one wave per (bh,i) row
__global__ void mix_fwd_kernel(MopkEdgewiseArgs a, EwDims d, EwSaved s) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63;
    const int i = row % d.N;
    const int64_t bh = row / d.N;
    RowGates rg;
    load_row_gates(rg, s.ga + bh * 4 * d.r * d.N, d.r, d.N, i);
    const float *gb = s.gb + bh * 4 * d.r * d.N;
    const int64_t vstride = d.BH * (int64_t)d.N * d.LD;
    const float nb = a.beta_not / (float)max(1, d.V - 1);
    float *P = s.P + row * d.LD;
    float mx = -INFINITY;
    for (int j = lane; j < d.N; j += 64) {
        float sv[MAXV], O, lse, G[4];
        view_stats(s.S, vstride, row * d.LD + j, d.V, sv, O, lse);
        gates_at(G, rg, gb, d.r, d.N, j);
        const float cr = s.Cr[row * d.LD + j];
        const float sm = sv[0] + G[0] * O + G[1] * (lse - sv[0]) - G[2] * (nb * O) + G[3] * cr;
        P[j] = sm; mx = fmaxf(mx, sm);
    }
    mx = wave_max(mx);
    float den = 0.f;
    for (int j = lane; j < d.N; j += 64) { float e = expf(P[j] - mx); P[j] = e; den += e; }
    den = wave_sum(den);
    const float inv = 1.f / den;
    for (int j = lane; j < d.N; j += 64) P[j] *= inv;
}

template <typename T>
__global__ void combine_y_kernel(MopkEdgewiseArgs a, EwDims d, const float *ybase, const float *ychain, const float *wsig) {
    const int64_t total = d.BH * d.N * d.dk;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int dd = idx % d.dk; const int n = (idx / d.dk) % d.N; const int64_t bh = idx / ((int64_t)d.dk * d.N);
    const int b = bh / d.H, h = bh % d.H;
    T *y = (T *)a.y.ptr + b * a.y.sb + h * a.y.sh + n * a.y.sn + dd;
    st_from_f32(y, ybase[idx] + wsig[0] * ychain[idx]);          // :562
}

// ------------------------------------------------------------------ backward kernels
template <typename T>
__global__ void prep_dy_kernel(MopkEdgewiseArgs a, EwDims d, float *dyc) {
    const int64_t total = d.BH * d.N * d.dk;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int dd = idx % d.dk; const int n = (idx / d.dk) % d.N; const int64_t bh = idx / ((int64_t)d.dk * d.N);
    const int b = bh / d.H, h = bh % d.H;
    dyc[idx] = ld_as_f32((const T *)a.dy.ptr + b * a.dy.sb + h * a.dy.sh + n * a.dy.sn + dd);
}
// dlogit_part[bh] = w(1-w) sum dy * ychain
__global__ void dlogit_kernel(const float *dyc, const float *ychain, const float *wsig, float *out, int64_t per_bh) {
    __shared__ float red[256];
    const int64_t bh = blockIdx.x;
    float sm = 0.f;
    for (int64_t i = threadIdx.x; i < per_bh; i += 256) sm += dyc[bh * per_bh + i] * ychain[bh * per_bh + i];
    red[threadIdx.x] = sm; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { const float w = wsig[0]; out[bh] = red[0] * w * (1.f - w); }
}

__device__ __forceinline__ void gate_terms(float tg[4], const float sv[MAXV], float O, float lse, float nb, float cr) {
    tg[0] = O; tg[1] = lse - sv[0]; tg[2] = -nb * O; tg[3] = cr;
}
// wave per (bh,i): dSmix = P (dP - sum P dP) (in place over dP); da[g,k,i] = sum_j dZ_g b[g,k,j]
__global__ void mix_bwd_rows_kernel(MopkEdgewiseArgs a, EwDims d, EwSaved s, EwWork w) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63;
    const int i = row % d.N;
    const int64_t bh = row / d.N;
    const float *P = s.P + row * d.LD;
    float *dP = w.dP + row * d.LD;
    float dot = 0.f;
    for (int j = lane; j < d.N; j += 64) dot += P[j] * dP[j];
    dot = wave_sum(dot);
    RowGates rg;
    load_row_gates(rg, s.ga + bh * 4 * d.r * d.N, d.r, d.N, i);
    const float *gb = s.gb + bh * 4 * d.r * d.N;
    const int64_t vstride = d.BH * (int64_t)d.N * d.LD;
    const float nb = a.beta_not / (float)max(1, d.V - 1);
    float da[4][MAXR];
    for (int q = 0; q < 4; ++q) for (int k = 0; k < MAXR; ++k) da[q][k] = 0.f;
    for (int j = lane; j < d.N; j += 64) {
        const float dsm = P[j] * (dP[j] - dot);
        dP[j] = dsm;
        float sv[MAXV], O, lse, G[4], tg[4];
        view_stats(s.S, vstride, row * d.LD + j, d.V, sv, O, lse);
        gates_at(G, rg, gb, d.r, d.N, j);
        gate_terms(tg, sv, O, lse, nb, s.Cr[row * d.LD + j]);
        for (int q = 0; q < 4; ++q) {
            const float dz = dsm * tg[q] * G[q] * (1.f - G[q]);
            for (int k = 0; k < d.r; ++k) da[q][k] = fmaf(dz, gb[(q * d.r + k) * d.N + j], da[q][k]);
        }
    }
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < d.r; ++k) {
            const float v = wave_sum(da[q][k]);
            if (lane == 0) w.da[(bh * 4 * d.r + q * d.r + k) * d.N + i] = v;
        }
}
// thread per (bh,j): db[g,k,j] = sum_i dZ_g[i,j] a[g,k,i]
__global__ void mix_bwd_cols_kernel(MopkEdgewiseArgs a, EwDims d, EwSaved s, EwWork w) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t bh = blockIdx.y;
    if (j >= d.N) return;
    const float *ga = s.ga + bh * 4 * d.r * d.N, *gb = s.gb + bh * 4 * d.r * d.N;
    float bj[4][MAXR], db[4][MAXR];
    for (int q = 0; q < 4; ++q) for (int k = 0; k < MAXR; ++k) { bj[q][k] = k < d.r ? gb[(q * d.r + k) * d.N + j] : 0.f; db[q][k] = 0.f; }
    const int64_t vstride = d.BH * (int64_t)d.N * d.LD;
    const float nb = a.beta_not / (float)max(1, d.V - 1);
    for (int i = 0; i < d.N; ++i) {
        const int64_t off = (bh * d.N + i) * d.LD + j;
        const float dsm = w.dP[off];
        float sv[MAXV], O, lse, tg[4];
        view_stats(s.S, vstride, off, d.V, sv, O, lse);
        gate_terms(tg, sv, O, lse, nb, s.Cr[off]);
        for (int q = 0; q < 4; ++q) {
            float z = 0.f;
            for (int k = 0; k < d.r; ++k) z = fmaf(ga[(q * d.r + k) * d.N + i], bj[q][k], z);
            const float G = sigmoidf_(z);
            const float dz = dsm * tg[q] * G * (1.f - G);
            for (int k = 0; k < d.r; ++k) db[q][k] = fmaf(dz, ga[(q * d.r + k) * d.N + i], db[q][k]);
        }
    }
    for (int q = 0; q < 4; ++q) for (int k = 0; k < d.r; ++k) w.db[(bh * 4 * d.r + q * d.r + k) * d.N + j] = db[q][k];
}
// thread per (bh,n): d row_feat = Wr^T da, d col_feat = Wc^T db -> gradients of the 2V+2 means
__global__ void gate_ab_bwd_kernel(MopkEdgewiseArgs a, EwDims d, EwWork w) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.BH * d.N) return;
    const int n = idx % d.N;
    const int64_t bh = idx / d.N;
    float drow[2 * MAXV + 2], dcol[2 * MAXV + 2];
    for (int c = 0; c < d.C; ++c) { drow[c] = 0.f; dcol[c] = 0.f; }
    for (int o = 0; o < 4 * d.r; ++o) {
        const float va = w.da[(bh * 4 * d.r + o) * d.N + n], vb = w.db[(bh * 4 * d.r + o) * d.N + n];
        for (int c = 0; c < d.C; ++c) { drow[c] = fmaf(a.Wr[o * d.C + c], va, drow[c]); dcol[c] = fmaf(a.Wc[o * d.C + c], vb, dcol[c]); }
    }
    const int64_t n1 = d.BH * d.N;
    for (int v = 0; v < d.V; ++v) {
        w.drS[v * n1 + idx] = drow[v] + dcol[d.V + v];
        w.dcS[v * n1 + idx] = drow[d.V + v] + dcol[v];
    }
    w.drCr[idx] = drow[2 * d.V]; w.dcCr[idx] = dcol[2 * d.V];
    w.drCl[idx] = drow[2 * d.V + 1]; w.dcCl[idx] = dcol[2 * d.V + 1];
}
// block per (o, c|bias, row|col): dW[o,c] = sum_{bh,n} d{a,b}[bh,o,n] feat[bh,c,n]
__global__ void gate_w_bwd_kernel(MopkEdgewiseArgs a, EwDims d, EwSaved s, EwWork w) {
    __shared__ float red[256];
    const int o = blockIdx.x, c = blockIdx.y;
    const bool col = blockIdx.z == 1;
    const float *dab = col ? w.db : w.da;
    float sm = 0.f;
    const int64_t total = d.BH * d.N;
    for (int64_t idx = threadIdx.x; idx < total; idx += 256) {
        const int n = idx % d.N; const int64_t bh = idx / d.N;
        const float g = dab[(bh * 4 * d.r + o) * d.N + n];
        sm += c < d.C ? g * row_feat(s, d, c, bh, n, col) : g;
    }
    red[threadIdx.x] = sm; __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) {
        if (c < d.C) (col ? a.dWc : a.dWr)[o * d.C + c] = red[0];
        else (col ? a.dbc : a.dbr)[o] = red[0];
    }
}
// dCf += (dSmix G3 + (drCr_i + dcCr_j)/N) / (Cf+eps) ; dCb = ((drCl_i + dcCl_j)/N)/(Cb+eps)
__global__ void dchain_seed_kernel(MopkEdgewiseArgs a, EwDims d, EwSaved s, EwWork w) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63;
    const int i = row % d.N;
    const int64_t bh = row / d.N;
    const float *ga = s.ga + bh * 4 * d.r * d.N, *gb = s.gb + bh * 4 * d.r * d.N;
    float a3[MAXR];
    for (int k = 0; k < MAXR; ++k) a3[k] = k < d.r ? ga[(3 * d.r + k) * d.N + i] : 0.f;
    const float invN = 1.f / d.N;
    const float ri = w.drCr[row] * invN, li = w.drCl[row] * invN;
    for (int j = lane; j < d.N; j += 64) {
        float z = 0.f;
        for (int k = 0; k < d.r; ++k) z = fmaf(a3[k], gb[(3 * d.r + k) * d.N + j], z);
        const float G3 = sigmoidf_(z);
        const int64_t off = row * d.LD + j;
        const float dcr = w.dP[off] * G3 + ri + w.dcCr[bh * d.N + j] * invN;
        w.dCf[off] += dcr * expf(-s.Cr[off]);        // 1/(Cf+eps) = exp(-Cr)
        w.dCb[off] = (li + w.dcCl[bh * d.N + j] * invN) * expf(-s.Cl[off]);
    }
}
// wave per (v,bh,i): dS_v = A_v (dA_v - sum A_v dA_v) + dSmix coef_v + (drS_i + dcS_j)/N   (in place over dA)
__global__ void ds_final_kernel(MopkEdgewiseArgs a, EwDims d, EwSaved s, EwWork w) {
    const int64_t grow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t rows_per_v = d.BH * d.N;
    if (grow >= d.V * rows_per_v) return;
    const int lane = threadIdx.x & 63;
    const int v = grow / rows_per_v;
    const int64_t row = grow % rows_per_v;
    const int i = row % d.N;
    const int64_t bh = row / d.N;
    const int64_t vstride = rows_per_v * d.LD;
    const float *A = s.A + v * vstride + row * d.LD;
    float *dA = w.dA + v * vstride + row * d.LD;
    float dot = 0.f;
    for (int j = lane; j < d.N; j += 64) dot += A[j] * dA[j];
    dot = wave_sum(dot);
    RowGates rg;
    load_row_gates(rg, s.ga + bh * 4 * d.r * d.N, d.r, d.N, i);
    const float *gb = s.gb + bh * 4 * d.r * d.N;
    const float nb = a.beta_not / (float)max(1, d.V - 1);
    const float invN = 1.f / d.N;
    const float ri = w.drS[v * rows_per_v + row] * invN;
    for (int j = lane; j < d.N; j += 64) {
        float sv[MAXV], O, lse, G[4];
        view_stats(s.S, vstride, row * d.LD + j, d.V, sv, O, lse);
        gates_at(G, rg, gb, d.r, d.N, j);
        const float pi = expf(sv[v] - lse);
        const float coef = v == 0 ? (1.f - G[1] + G[1] * pi) : (G[0] - nb * G[2] + G[1] * pi);
        dA[j] = A[j] * (dA[j] - dot) + w.dP[row * d.LD + j] * coef + ri + w.dcS[v * rows_per_v + bh * d.N + j] * invN;
    }
}
// block (64 x 4) per (bh, d-chunk): scatter dq/dk/dv to the caller's layout, reduce scale grads over n
template <typename T>
__global__ void scale_bwd_kernel(MopkEdgewiseArgs a, EwDims d, EwSaved s, EwWork w) {
    __shared__ float red[4][64];
    const int64_t bh = blockIdx.x;
    const int dd = blockIdx.y * 64 + threadIdx.x;
    const int b = bh / d.H, h = bh % d.H;
    const bool ok = dd < d.dk;
    const int64_t nd = d.BH * d.N * d.dk;
    float asq[MAXV], av0 = 0.f, avL = 0.f;
    for (int v = 0; v < MAXV; ++v) asq[v] = 0.f;
    if (ok) {
        const float s0 = a.vs0[h * d.dk + dd], sL = a.vsL[h * d.dk + dd];
        for (int n = threadIdx.y; n < d.N; n += 4) {
            const int64_t o = (bh * d.N + n) * d.dk + dd;
            float dqsum = 0.f;
            for (int v = 0; v < d.V; ++v) {
                const float g = w.dQe[v * nd + o];
                const float sc = a.sqk[(v * d.H + h) * d.dk + dd];
                const T *q = (const T *)a.q.ptr + v * a.q.sv + b * a.q.sb + h * a.q.sh + n * a.q.sn + dd;
                asq[v] = fmaf(g, ld_as_f32(q), asq[v]);
                if (a.dq.sv != 0) st_from_f32((T *)a.dq.ptr + v * a.dq.sv + b * a.dq.sb + h * a.dq.sh + n * a.dq.sn + dd, g * sc);
                else dqsum = fmaf(g, sc, dqsum);
            }
            if (a.dq.sv == 0) st_from_f32((T *)a.dq.ptr + b * a.dq.sb + h * a.dq.sh + n * a.dq.sn + dd, dqsum);
            for (int v = 0; v < d.Vk; ++v)
                st_from_f32((T *)a.dk_.ptr + v * a.dk_.sv + b * a.dk_.sb + h * a.dk_.sh + n * a.dk_.sn + dd, w.dKc[v * nd + o]);
            const float g0 = w.dV0[o], gL = w.dVL[o];
            const float x0 = ld_as_f32((const T *)a.v0.ptr + b * a.v0.sb + h * a.v0.sh + n * a.v0.sn + dd);
            const float xL = ld_as_f32((const T *)a.vL.ptr + b * a.vL.sb + h * a.vL.sh + n * a.vL.sn + dd);
            av0 = fmaf(g0, x0, av0); avL = fmaf(gL, xL, avL);
            if (a.dv0.ptr == a.dvL.ptr) {   // shared v: one gradient tensor, dv = dv0*vs0 + dvL*vsL
                st_from_f32((T *)a.dv0.ptr + b * a.dv0.sb + h * a.dv0.sh + n * a.dv0.sn + dd, g0 * s0 + gL * sL);
            } else {
                st_from_f32((T *)a.dv0.ptr + b * a.dv0.sb + h * a.dv0.sh + n * a.dv0.sn + dd, g0 * s0);
                st_from_f32((T *)a.dvL.ptr + b * a.dvL.sb + h * a.dvL.sh + n * a.dvL.sn + dd, gL * sL);
            }
        }
    }
    for (int v = 0; v < d.V + 2; ++v) {
        const float val = v < d.V ? asq[v] : (v == d.V ? av0 : avL);
        red[threadIdx.y][threadIdx.x] = val; __syncthreads();
        if (threadIdx.y == 0 && ok) {
            const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            if (v < d.V) a.dsqk_part[((int64_t)(b * d.V + v) * d.H + h) * d.dk + dd] = t;
            else if (v == d.V) a.dvs0_part[(int64_t)(b * d.H + h) * d.dk + dd] = t;
            else a.dvsL_part[(int64_t)(b * d.H + h) * d.dk + dd] = t;
        }
        __syncthreads();
    }
}

__global__ void axpy_kernel(float *y, const float *x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i];
}

// ------------------------------------------------------------------ orchestration
static inline GemmDesc gd(int M, int N, int K, int nb0, int nb1) {
    GemmDesc g{}; g.M = M; g.N = N; g.K = K; g.nb0 = nb0; g.nb1 = nb1; g.alpha = 1.f; g.beta = 0.f; g.alpha_dev = nullptr;
    return g;
}
#define RET_IF(x) do { int rc_ = (x); if (rc_ != MOPK_OK) return rc_; } while (0)

template <typename T>
static int ew_generic_fwd_t(const MopkEdgewiseArgs *a, hipStream_t st) {
    const EwDims d = ew_dims(a);
    if (d.V > MAXV || d.r > MAXR || d.V < 2 || d.r < 1) return MOPK_ERR_UNSUPPORTED;
    const EwSaved s = ew_carve_saved(a->saved, d, nullptr);
    const EwWork w = ew_carve_work(a->workspace, d, nullptr);
    const bool mf = a->precision == MOPK_PREC_BF16;
    const int N = d.N, dk = d.dk, LD = d.LD, V = d.V;
    const int64_t nd1 = (int64_t)N * dk, nn1 = (int64_t)N * LD, ndv = d.BH * nd1, nnv = d.BH * nn1;
    const int64_t tot = d.BH * nd1;
    const int BHi = (int)d.BH;
    hipLaunchKernelGGL((ew_prep_kernel<T>), dim3((tot + 255) / 256), dim3(256), 0, st, *a, d, s);
    MOPK_CHECK_LAUNCH();
    {   // S_v = Qe_v K^T                                                :500-503
        GemmDesc g = gd(N, N, dk, V, BHi);
        g.A = s.Qe; g.a_rs = dk; g.a_cs = 1; g.a_b0 = ndv; g.a_b1 = nd1;
        g.B = s.Kc; g.b_rs = 1; g.b_cs = dk; g.b_b0 = d.Vk > 1 ? ndv : 0; g.b_b1 = nd1;
        g.C = s.S; g.c_rs = LD; g.c_b0 = nnv; g.c_b1 = nn1;
        RET_IF(bgemm(g, mf, st));
    }
    const int64_t rowsV = (int64_t)V * d.BH * N, rows1 = d.BH * N;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rowsV + 3) / 4), dim3(256), 0, st, s.S, s.A, s.rS, rowsV, N, LD);  // :507
    hipLaunchKernelGGL(colmean_kernel, dim3((N + 255) / 256, V * BHi), dim3(256), 0, st, s.S, s.cS, N, LD);
    MOPK_CHECK_LAUNCH();
    // chains                                                            :508-515
    for (int m = 1; m < V; ++m) {
        GemmDesc g = gd(N, N, N, 1, BHi);
        g.A = m == 1 ? s.A : s.T + (m - 2) * nnv; g.a_rs = LD; g.a_cs = 1; g.a_b1 = nn1;
        g.B = s.A + m * nnv; g.b_rs = LD; g.b_cs = 1; g.b_b1 = nn1;
        g.C = s.T + (m - 1) * nnv; g.c_rs = LD; g.c_b1 = nn1;
        RET_IF(bgemm(g, mf, st));
        g.A = m == 1 ? s.A + (V - 1) * nnv : s.U + (m - 2) * nnv;
        g.B = s.A + (V - 1 - m) * nnv;
        g.C = s.U + (m - 1) * nnv;
        RET_IF(bgemm(g, mf, st));
    }
    const float *Cf = s.T + (V - 2) * nnv, *Cb = s.U + (V - 2) * nnv;
    hipLaunchKernelGGL(log_rows_kernel, dim3((rows1 + 3) / 4), dim3(256), 0, st, Cf, s.Cr, s.rCr, rows1, N, LD);     // :520
    hipLaunchKernelGGL(log_rows_kernel, dim3((rows1 + 3) / 4), dim3(256), 0, st, Cb, s.Cl, s.rCl, rows1, N, LD);     // :521
    hipLaunchKernelGGL(colmean_kernel, dim3((N + 255) / 256, BHi), dim3(256), 0, st, s.Cr, s.cCr, N, LD);
    hipLaunchKernelGGL(colmean_kernel, dim3((N + 255) / 256, BHi), dim3(256), 0, st, s.Cl, s.cCl, N, LD);
    hipLaunchKernelGGL(gate_ab_kernel, dim3((rows1 + 255) / 256), dim3(256), 0, st, *a, d, s);                        // :325-326
    hipLaunchKernelGGL(mix_fwd_kernel, dim3((rows1 + 3) / 4), dim3(256), 0, st, *a, d, s);                            // :537-551
    MOPK_CHECK_LAUNCH();
    {   // y_base = P V0 ; y_chain = A0(A1(...(A_{V-1} VL))) = Cf VL      :554-560
        GemmDesc g = gd(N, dk, N, 1, BHi);
        g.A = s.P; g.a_rs = LD; g.a_cs = 1; g.a_b1 = nn1;
        g.B = s.V0; g.b_rs = dk; g.b_cs = 1; g.b_b1 = nd1;
        g.C = w.ybase; g.c_rs = dk; g.c_b1 = nd1;
        RET_IF(bgemm(g, mf, st));
        g.A = Cf; g.B = s.VL; g.C = s.ychain;
        RET_IF(bgemm(g, mf, st));
    }
    hipLaunchKernelGGL((combine_y_kernel<T>), dim3((tot + 255) / 256), dim3(256), 0, st, *a, d, w.ybase, s.ychain, s.wsig);
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}

template <typename T>
static int ew_generic_bwd_t(const MopkEdgewiseArgs *a, hipStream_t st) {
    const EwDims d = ew_dims(a);
    if (d.V > MAXV || d.r > MAXR || d.V < 2 || d.r < 1) return MOPK_ERR_UNSUPPORTED;
    const EwSaved s = ew_carve_saved(a->saved, d, nullptr);
    const EwWork w = ew_carve_work(a->workspace, d, nullptr);
    const bool mf = a->precision == MOPK_PREC_BF16;
    const int N = d.N, dk = d.dk, LD = d.LD, V = d.V;
    const int64_t nd1 = (int64_t)N * dk, nn1 = (int64_t)N * LD, ndv = d.BH * nd1, nnv = d.BH * nn1;
    const int64_t tot = d.BH * nd1, rows1 = d.BH * N, rowsV = (int64_t)V * rows1;
    const int BHi = (int)d.BH;
    const float *Cf = s.T + (V - 2) * nnv;
    hipLaunchKernelGGL((prep_dy_kernel<T>), dim3((tot + 255) / 256), dim3(256), 0, st, *a, d, w.dyc);
    hipLaunchKernelGGL(dlogit_kernel, dim3(BHi), dim3(256), 0, st, w.dyc, s.ychain, s.wsig, a->dlogit_part, nd1);
    MOPK_CHECK_LAUNCH();
    {
        GemmDesc g = gd(N, N, dk, 1, BHi);           // dP = dy V0^T
        g.A = w.dyc; g.a_rs = dk; g.a_cs = 1; g.a_b1 = nd1;
        g.B = s.V0; g.b_rs = 1; g.b_cs = dk; g.b_b1 = nd1;
        g.C = w.dP; g.c_rs = LD; g.c_b1 = nn1;
        RET_IF(bgemm(g, mf, st));
        g.B = s.VL; g.C = w.dCf; g.alpha_dev = s.wsig;   // dCf = w dy VL^T
        RET_IF(bgemm(g, mf, st));
        GemmDesc h = gd(N, dk, N, 1, BHi);           // dV0 = P^T dy
        h.A = s.P; h.a_rs = 1; h.a_cs = LD; h.a_b1 = nn1;
        h.B = w.dyc; h.b_rs = dk; h.b_cs = 1; h.b_b1 = nd1;
        h.C = w.dV0; h.c_rs = dk; h.c_b1 = nd1;
        RET_IF(bgemm(h, mf, st));
        h.A = Cf; h.C = w.dVL; h.alpha_dev = s.wsig;  // dVL = w Cf^T dy
        RET_IF(bgemm(h, mf, st));
    }
    hipLaunchKernelGGL(mix_bwd_rows_kernel, dim3((rows1 + 3) / 4), dim3(256), 0, st, *a, d, s, w);
    hipLaunchKernelGGL(mix_bwd_cols_kernel, dim3((N + 63) / 64, BHi), dim3(64), 0, st, *a, d, s, w);
    hipLaunchKernelGGL(gate_ab_bwd_kernel, dim3((rows1 + 255) / 256), dim3(256), 0, st, *a, d, w);
    hipLaunchKernelGGL(gate_w_bwd_kernel, dim3(4 * d.r, d.C + 1, 2), dim3(256), 0, st, *a, d, s, w);
    hipLaunchKernelGGL(dchain_seed_kernel, dim3((rows1 + 3) / 4), dim3(256), 0, st, *a, d, s, w);
    MOPK_CHECK_LAUNCH();
    // chain backward.  T_m = T_{m-1} A_m : dA_m += T_{m-1}^T D ; D <- D A_m^T ; dA_0 += D
    hipMemsetAsync(w.dA, 0, sizeof(float) * V * nnv, st);
    for (int dir = 0; dir < 2; ++dir) {
        const float *D = dir == 0 ? w.dCf : w.dCb;
        float *pp[2] = {w.D0, w.D1};
        int cur = 0;
        for (int m = V - 1; m >= 1; --m) {
            const int av = dir == 0 ? m : V - 1 - m;                 // view multiplied at step m
            const float *Tprev = dir == 0 ? (m == 1 ? s.A : s.T + (m - 2) * nnv)
                                          : (m == 1 ? s.A + (V - 1) * nnv : s.U + (m - 2) * nnv);
            GemmDesc g = gd(N, N, N, 1, BHi);        // dA[av] += Tprev^T D
            g.A = Tprev; g.a_rs = 1; g.a_cs = LD; g.a_b1 = nn1;
            g.B = D; g.b_rs = LD; g.b_cs = 1; g.b_b1 = nn1;
            g.C = w.dA + av * nnv; g.c_rs = LD; g.c_b1 = nn1; g.beta = 1.f;
            RET_IF(bgemm(g, mf, st));
            GemmDesc h = gd(N, N, N, 1, BHi);        // Dn = D A[av]^T
            h.A = D; h.a_rs = LD; h.a_cs = 1; h.a_b1 = nn1;
            h.B = s.A + av * nnv; h.b_rs = 1; h.b_cs = LD; h.b_b1 = nn1;
            h.C = pp[cur]; h.c_rs = LD; h.c_b1 = nn1;
            RET_IF(bgemm(h, mf, st));
            D = pp[cur]; cur ^= 1;
        }
        const int first = dir == 0 ? 0 : V - 1;       // dA[first] += D
        hipLaunchKernelGGL(axpy_kernel, dim3((nnv + 255) / 256), dim3(256), 0, st, w.dA + first * nnv, D, nnv);
        MOPK_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(ds_final_kernel, dim3((rowsV + 3) / 4), dim3(256), 0, st, *a, d, s, w);
    MOPK_CHECK_LAUNCH();
    {
        GemmDesc g = gd(N, dk, N, V, BHi);           // dQe_v = dS_v K
        g.A = w.dA; g.a_rs = LD; g.a_cs = 1; g.a_b0 = nnv; g.a_b1 = nn1;
        g.B = s.Kc; g.b_rs = dk; g.b_cs = 1; g.b_b0 = d.Vk > 1 ? ndv : 0; g.b_b1 = nd1;
        g.C = w.dQe; g.c_rs = dk; g.c_b0 = ndv; g.c_b1 = nd1;
        RET_IF(bgemm(g, mf, st));
        for (int v = 0; v < V; ++v) {                // dK (+)= dS_v^T Qe_v
            GemmDesc h = gd(N, dk, N, 1, BHi);
            h.A = w.dA + v * nnv; h.a_rs = 1; h.a_cs = LD; h.a_b1 = nn1;
            h.B = s.Qe + v * ndv; h.b_rs = dk; h.b_cs = 1; h.b_b1 = nd1;
            h.C = w.dKc + (d.Vk > 1 ? v * ndv : 0); h.c_rs = dk; h.c_b1 = nd1;
            h.beta = (d.Vk == 1 && v > 0) ? 1.f : 0.f;
            RET_IF(bgemm(h, mf, st));
        }
    }
    hipLaunchKernelGGL((scale_bwd_kernel<T>), dim3(BHi, (dk + 63) / 64), dim3(64, 4), 0, st, *a, d, s, w);
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}

int ew_generic_fwd(const MopkEdgewiseArgs *a, hipStream_t st) {
    return a->io_dtype == MOPK_BF16 ? ew_generic_fwd_t<unsigned short>(a, st) : ew_generic_fwd_t<float>(a, st);
}
int ew_generic_bwd(const MopkEdgewiseArgs *a, hipStream_t st) {
    return a->io_dtype == MOPK_BF16 ? ew_generic_bwd_t<unsigned short>(a, st) : ew_generic_bwd_t<float>(a, st);
}

}  // namespace mopk
