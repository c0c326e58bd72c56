// EdgewiseMSA core -- generic multi-kernel path (any N / dk / V<=8 / r<=8; low-rank or dense gate head, S lens bank).
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

constexpr int MAXV = 8, MAXR = 8, MAXL = MOPK_MAX_LENS, HID = MOPK_DENSE_HIDDEN;
constexpr int MAXC = 2 * MAXV + 2 + MAXL * MAXV;          // feature channels: S, S^T, Cr, Cl, lens[l*V+v]   :522-533
constexpr float EPS_CHAIN = 1e-6f;  // attention_variants.py:516

struct EwDims {
    int B, H, N, dk, V, r, Vk, C, LD;
    int64_t BH;
    int dense, k3, L;                // gate-head / lens-bank variant (MopkEdgewiseExt)
    int dil[MAXL];
};
static EwDims ew_dims(const MopkEdgewiseArgs *a) {
    EwDims d{};
    d.B = a->B; d.H = a->H; d.N = a->N; d.dk = a->dk; d.V = a->V; d.r = a->r;
    d.Vk = a->k.sv == 0 ? 1 : a->V;
    d.LD = (int)round_up(a->N, 4);
    d.BH = (int64_t)a->B * a->H;
    if (a->ext) {
        d.dense = a->ext->gate_mode == 1; d.k3 = d.dense && a->ext->use_k3; d.L = a->ext->n_lens;
        for (int l = 0; l < MAXL; ++l) d.dil[l] = l < d.L ? a->ext->lens_dil[l] : 0;
    }
    d.C = 2 * a->V + 2 + d.L * a->V;
    return d;
}
// device copy of the variant parameters (the ext struct itself is host memory)
struct EwExt {
    const float *lens_w, *W1, *b1, *W3, *b3, *W2, *b2;
    float *dlens_w, *dW1, *db1, *dW3, *db3, *dW2, *db2;
};
static EwExt ew_ext(const MopkEdgewiseArgs *a) {
    EwExt e{};
    if (a->ext) {
        const MopkEdgewiseExt &x = *a->ext;
        e.lens_w = x.lens_w; e.W1 = x.W1; e.b1 = x.b1; e.W3 = x.W3; e.b3 = x.b3; e.W2 = x.W2; e.b2 = x.b2;
        e.dlens_w = x.dlens_w; e.dW1 = x.dW1; e.db1 = x.db1; e.dW3 = x.dW3; e.db3 = x.db3; e.dW2 = x.dW2; e.db2 = x.db2;
    }
    return e;
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
    float *Lz, *rL, *cL;          // lens bank: (L*V,BH,N,LD) planes and their row / col means (L*V,BH,N)
    float *G, *X1, *H2, *X3;      // dense head: gates (4,BH,N,LD); conv1 pre-activation, conv3 input / output (HID,BH,N,LD)
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
    s.Lz = c.take<float>((size_t)d.L * d.V * nn); s.rL = c.take<float>((size_t)d.L * d.V * n1); s.cL = c.take<float>((size_t)d.L * d.V * n1);
    s.G = c.take<float>(d.dense ? 4 * nn : 0); s.X1 = c.take<float>(d.dense ? HID * nn : 0);
    s.H2 = c.take<float>(d.k3 ? HID * nn : 0); s.X3 = c.take<float>(d.k3 ? HID * nn : 0);
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
    float *dZ, *DX1, *DX3;         // dense head: (4,BH,N,LD), (HID,BH,N,LD) x2
    float *dSf, *dCrf, *dClf;      // gradients reaching S_v / Cr / Cl through per-edge features: (V,BH,N,LD), (BH,N,LD) x2
    float *dLz;                    // (L*V,BH,N,LD)
    float *drL, *dcL;              // low-rank head: gradients of the lens planes' row / col means (L*V,BH,N)
    float *part;                   // [PR_BLOCKS][PR_MAXOUT] partial sums of the weight-gradient reductions
};
constexpr int PR_BLOCKS = 768, PR_MAXOUT = HID * (HID * 9 + 1);   // largest reduction: dW3 | db3
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
    w.dZ = c.take<float>(d.dense ? 4 * nn : 0); w.DX1 = c.take<float>(d.dense ? HID * nn : 0); w.DX3 = c.take<float>(d.k3 ? HID * nn : 0);
    const bool edge = d.dense || d.L > 0;
    w.dSf = c.take<float>(edge ? (size_t)d.V * nn : 0); w.dCrf = c.take<float>(d.dense ? nn : 0); w.dClf = c.take<float>(d.dense ? nn : 0);
    w.dLz = c.take<float>((size_t)d.L * d.V * nn);
    w.drL = c.take<float>((size_t)d.L * d.V * n1); w.dcL = c.take<float>((size_t)d.L * d.V * n1);
    w.part = c.take<float>(edge ? (size_t)PR_BLOCKS * PR_MAXOUT : 0);
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
// optional attention mask (1 = keep) of row (bh, i): extension, acts on the probabilities only (see MopkEdgewiseArgs.mask)
struct RowMask { const uint8_t *p; int64_t sb, sh, si; int H, N; };
__device__ __forceinline__ const uint8_t *mask_row(const RowMask &m, int64_t bhi) {      // bhi = bh * N + i
    if (!m.p) return nullptr;
    const int64_t bh = bhi / m.N, i = bhi % m.N;
    return m.p + (bh / m.H) * m.sb + (bh % m.H) * m.sh + i * m.si;
}
// rows = V * BH * N rows of S (view-major); the row mean (a gate feature) is of the unmasked scores
__global__ void softmax_rows_kernel(const float *in, float *out, float *rowmean, int64_t rows, int N, int LD, RowMask m, int64_t rows1) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float *p = in + row * LD;
    const uint8_t *mk = mask_row(m, row % rows1);
    float mx = -INFINITY, sm = 0.f;
    for (int j = lane; j < N; j += 64) { float v = p[j]; if (!mk || mk[j]) mx = fmaxf(mx, v); sm += v; }
    mx = wave_max(mx); sm = wave_sum(sm);
    float den = 0.f;
    for (int j = lane; j < N; j += 64) den += (!mk || mk[j]) ? expf(p[j] - mx) : 0.f;
    den = wave_sum(den);
    const float inv = 1.f / den;
    float *o = out + row * LD;
    for (int j = lane; j < N; j += 64) o[j] = (!mk || mk[j]) ? expf(p[j] - mx) * inv : 0.f;
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
    if (c == 2 * d.V + 1) return (col ? s.cCl : s.rCl)[o];
    return (col ? s.cL : s.rL)[(c - 2 * d.V - 2) * n1 + o];      // lens planes l*V+v   :531-533
}
// a = Wr row_feat + br ; b = Wc col_feat + bc     (attention_variants.py:323-326)
__global__ void gate_ab_kernel(MopkEdgewiseArgs a, EwDims d, EwSaved s) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.BH * d.N) return;
    const int n = idx % d.N;
    const int64_t bh = idx / d.N;
    float fr[MAXC], fc[MAXC];
    for (int c = 0; c < d.C; ++c) { fr[c] = row_feat(s, d, c, bh, n, false); fc[c] = row_feat(s, d, c, bh, n, true); }
    for (int o = 0; o < 4 * d.r; ++o) {
        float sa = a.br[o], sb = a.bc[o];
        for (int c = 0; c < d.C; ++c) { sa = fmaf(a.Wr[o * d.C + c], fr[c], sa); sb = fmaf(a.Wc[o * d.C + c], fc[c], sb); }
        s.ga[(bh * 4 * d.r + o) * d.N + n] = sa;
        s.gb[(bh * 4 * d.r + o) * d.N + n] = sb;
    }
}

// ------------------------------------------------------------------ lens bank + dense gate head (MopkEdgewiseExt)
__device__ __forceinline__ float gelu_tanh(float x) {                     // nn.GELU(approximate="tanh")  :252
    return 0.5f * x * (1.f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float gelu_tanh_grad(float x) {
    const float t = tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x));
    return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * 0.7978845608028654f * (1.f + 3.f * 0.044715f * x * x);
}
// wave per row: row means of a stack of (rows, N) planes
__global__ void rowmean_kernel(const float *in, float *rowmean, int64_t rows, int N, int LD) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float sm = 0.f;
    for (int j = lane; j < N; j += 64) sm += in[row * LD + j];
    sm = wave_sum(sm);
    if (lane == 0) rowmean[row] = sm / N;
}
// Lz[l*V+v] = depthwise 3x3 cross-correlation of S_v, dilation = padding = dil[l]   :430-436, :526-528
__global__ void lens_fwd_kernel(EwDims d, EwSaved s, EwExt e) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.N * d.N) return;
    const int i = p / d.N, j = p % d.N, lv = blockIdx.z, l = lv / d.V, v = lv % d.V, dl = d.dil[l];
    const int64_t pl = (int64_t)d.N * d.LD, bh = blockIdx.y;
    const float *Sv = s.S + (v * d.BH + bh) * pl;
    float acc = 0.f;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            const int ii = i + (a - 1) * dl, jj = j + (b - 1) * dl;
            if (ii >= 0 && ii < d.N && jj >= 0 && jj < d.N) acc = fmaf(e.lens_w[(lv * 3 + a) * 3 + b], Sv[(int64_t)ii * d.LD + jj], acc);
        }
    s.Lz[(lv * d.BH + bh) * pl + (int64_t)i * d.LD + j] = acc;
}
// per-edge feature vector [S_v(i,j), S_v(j,i), Cr, Cl, lens]   :522-534
__device__ __forceinline__ float edge_feat(const EwDims &d, const EwSaved &s, int c, int64_t bh, int i, int j) {
    const int64_t pl = (int64_t)d.N * d.LD, nn = d.BH * pl, o = bh * pl + (int64_t)i * d.LD + j;
    if (c < d.V) return s.S[c * nn + o];
    if (c < 2 * d.V) return s.S[(c - d.V) * nn + bh * pl + (int64_t)j * d.LD + i];
    if (c == 2 * d.V) return s.Cr[o];
    if (c == 2 * d.V + 1) return s.Cl[o];
    return s.Lz[(c - 2 * d.V - 2) * nn + o];
}
// The two per-edge kernels that read planes transposed (S_v(j,i) as a feature, dX1(j,i) for its gradient) work on PAIRS of 16 x 16
// tiles {(ti,tj), (tj,ti)}: both tiles of every plane are staged in LDS with row-contiguous loads, and each tile's transposed
// operand is the other tile read with swapped indices (a per-pixel transposed read touches one 4-byte word per cache line).
constexpr int TP = 16;
__device__ __forceinline__ void pair_tiles(int pidx, int nt, int &ti, int &tj) {      // pair index -> ti <= tj, row-major over the upper triangle
    int rrow = 0;
    while (pidx >= nt - rrow) { pidx -= nt - rrow; ++rrow; }
    ti = rrow; tj = rrow + pidx;
}
inline dim3 pair_grid(int N, int BH) { const int nt = (N + TP - 1) / TP; return dim3(nt * (nt + 1) / 2, BH); }
// planes[v] tile (r0.., c0..) -> T[v][lr][lc] (zero outside the map)
__device__ __forceinline__ void stage_tile(float (*T)[TP][TP + 1], const float *planes, int nplanes, int64_t nn, int64_t base, int LD, int N, int r0, int c0) {
    const int lr = threadIdx.x / TP, lc = threadIdx.x % TP, i = r0 + lr, j = c0 + lc;
    const bool ok = i < N && j < N;
    for (int v = 0; v < nplanes; ++v) T[v][lr][lc] = ok ? planes[v * nn + base + (int64_t)i * LD + j] : 0.f;
}
// conv1 (1x1, C->16) + GELU [+ GELU for use_k3, :315-316]; without k3 also conv2 + sigmoid   :312-318
__global__ __launch_bounds__(TP * TP) void dense_gate_fwd_kernel(EwDims d, EwSaved s, EwExt e) {
    __shared__ float Ta[MAXV][TP][TP + 1], Tb[MAXV][TP][TP + 1];
    __shared__ __attribute__((aligned(16))) float Ws[(MAXC + 1) * HID + 4 * HID + 4];   // W1^T [c][k] | b1 [k] | W2^T [k][g] | b2 [g]: uniform 16-byte LDS reads
    for (int t = threadIdx.x; t < HID * d.C; t += TP * TP) Ws[(t % d.C) * HID + t / d.C] = e.W1[t];
    if (threadIdx.x < HID) Ws[MAXC * HID + threadIdx.x] = e.b1[threadIdx.x];
    if (threadIdx.x < 4 * HID) Ws[(MAXC + 1) * HID + (threadIdx.x % HID) * 4 + threadIdx.x / HID] = e.W2[threadIdx.x];
    if (threadIdx.x < 4) Ws[(MAXC + 1) * HID + 4 * HID + threadIdx.x] = e.b2[threadIdx.x];
    int ti, tj;
    pair_tiles(blockIdx.x, (d.N + TP - 1) / TP, ti, tj);
    const int64_t bh = blockIdx.y, pl = (int64_t)d.N * d.LD, nn = d.BH * pl;
    stage_tile(Ta, s.S, d.V, nn, bh * pl, d.LD, d.N, ti * TP, tj * TP);
    if (ti != tj) stage_tile(Tb, s.S, d.V, nn, bh * pl, d.LD, d.N, tj * TP, ti * TP);
    __syncthreads();
    const int lr = threadIdx.x / TP, lc = threadIdx.x % TP;
    for (int side = 0; side < (ti != tj ? 2 : 1); ++side) {
        const int i = (side ? tj : ti) * TP + lr, j = (side ? ti : tj) * TP + lc;
        if (i >= d.N || j >= d.N) continue;
        float (*Td)[TP][TP + 1] = side ? Tb : Ta, (*Tt)[TP][TP + 1] = (ti == tj) ? Ta : (side ? Ta : Tb);
        const int64_t o = bh * pl + (int64_t)i * d.LD + j;
        float x[HID], h[HID];
        for (int k4 = 0; k4 < HID / 4; ++k4) { const float4 b = *(const float4 *)&Ws[MAXC * HID + 4 * k4]; x[4 * k4] = b.x; x[4 * k4 + 1] = b.y; x[4 * k4 + 2] = b.z; x[4 * k4 + 3] = b.w; }
        for (int c = 0; c < d.C; ++c) {
            const float f = c < d.V ? Td[c][lr][lc] : c < 2 * d.V ? Tt[c - d.V][lc][lr] : edge_feat(d, s, c, bh, i, j);
#pragma unroll
            for (int k4 = 0; k4 < HID / 4; ++k4) {
                const float4 wv = *(const float4 *)&Ws[c * HID + 4 * k4];
                x[4 * k4] = fmaf(wv.x, f, x[4 * k4]); x[4 * k4 + 1] = fmaf(wv.y, f, x[4 * k4 + 1]);
                x[4 * k4 + 2] = fmaf(wv.z, f, x[4 * k4 + 2]); x[4 * k4 + 3] = fmaf(wv.w, f, x[4 * k4 + 3]);
            }
        }
#pragma unroll
        for (int k = 0; k < HID; ++k) { s.X1[k * nn + o] = x[k]; h[k] = gelu_tanh(x[k]); }
        if (d.k3) {
#pragma unroll
            for (int k = 0; k < HID; ++k) s.H2[k * nn + o] = gelu_tanh(h[k]);
        } else {
            const float4 b2 = *(const float4 *)&Ws[(MAXC + 1) * HID + 4 * HID];
            float z[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
            for (int k = 0; k < HID; ++k) {
                const float4 wv = *(const float4 *)&Ws[(MAXC + 1) * HID + 4 * k];
                z[0] = fmaf(wv.x, h[k], z[0]); z[1] = fmaf(wv.y, h[k], z[1]); z[2] = fmaf(wv.z, h[k], z[2]); z[3] = fmaf(wv.w, h[k], z[3]);
            }
            for (int g = 0; g < 4; ++g) s.G[g * nn + o] = sigmoidf_(z[g]);
        }
    }
}
// use_k3: mid3 (3x3, 16->16, pad 1) on H2, then conv2 + sigmoid.  Weights sit in LDS as [(c, tap)][k]: per input value four uniform
// 16-byte reads feed 16 independent FMA chains (as scalar loads of W3[k][c][tap] the kernel was bound by the scalar-memory latency)
__global__ __launch_bounds__(256) void dense_k3_fwd_kernel(EwDims d, EwSaved s, EwExt e) {
    __shared__ __attribute__((aligned(16))) float W3s[HID * 9 * HID + HID + 4 * HID + 4];   // W3 [(c*9+tap)][k] | b3 | W2^T [k][g] | b2
    for (int t = threadIdx.x; t < HID * HID * 9; t += 256) { const int k = t / (HID * 9), ct = t % (HID * 9); W3s[ct * HID + k] = e.W3[t]; }
    if (threadIdx.x < HID) W3s[HID * 9 * HID + threadIdx.x] = e.b3[threadIdx.x];
    if (threadIdx.x < 4 * HID) W3s[HID * 9 * HID + HID + (threadIdx.x % HID) * 4 + threadIdx.x / HID] = e.W2[threadIdx.x];
    if (threadIdx.x < 4) W3s[HID * 9 * HID + HID + 4 * HID + threadIdx.x] = e.b2[threadIdx.x];
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.N * d.N) return;
    const int i = p / d.N, j = p % d.N;
    const int64_t bh = blockIdx.y, pl = (int64_t)d.N * d.LD, nn = d.BH * pl, o = bh * pl + (int64_t)i * d.LD + j;
    float x3[HID];
#pragma unroll
    for (int k = 0; k < HID; ++k) x3[k] = W3s[HID * 9 * HID + k];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            const int ii = i + a - 1, jj = j + b - 1;
            if (ii < 0 || ii >= d.N || jj < 0 || jj >= d.N) continue;
            const int64_t q = bh * pl + (int64_t)ii * d.LD + jj;
            for (int c = 0; c < HID; ++c) {
                const float hv = s.H2[c * nn + q];
                const float4 *wr = (const float4 *)&W3s[(c * 9 + a * 3 + b) * HID];
#pragma unroll
                for (int k4 = 0; k4 < HID / 4; ++k4) {
                    const float4 wv = wr[k4];
                    x3[4 * k4] = fmaf(wv.x, hv, x3[4 * k4]); x3[4 * k4 + 1] = fmaf(wv.y, hv, x3[4 * k4 + 1]);
                    x3[4 * k4 + 2] = fmaf(wv.z, hv, x3[4 * k4 + 2]); x3[4 * k4 + 3] = fmaf(wv.w, hv, x3[4 * k4 + 3]);
                }
            }
        }
#pragma unroll
    for (int k = 0; k < HID; ++k) s.X3[k * nn + o] = x3[k];
    const float *w2 = &W3s[HID * 9 * HID + HID];
    const float4 b2 = *(const float4 *)&w2[4 * HID];
    float z[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
    for (int k = 0; k < HID; ++k) {
        const float4 wv = *(const float4 *)&w2[4 * k];
        z[0] = fmaf(wv.x, x3[k], z[0]); z[1] = fmaf(wv.y, x3[k], z[1]); z[2] = fmaf(wv.z, x3[k], z[2]); z[3] = fmaf(wv.w, x3[k], z[3]);
    }
    for (int g = 0; g < 4; ++g) s.G[g * nn + o] = sigmoidf_(z[g]);
}

struct RowGates { float a[4][MAXR]; };
// (loops over the rank are unrolled to MAXR with a `k < r` guard: with a runtime trip count the small arrays are indexed dynamically
//  and live in scratch memory -- 16 scratch loads per edge in the kernels below)
__device__ __forceinline__ void load_row_gates(RowGates &g, const float *ga_bh, int r, int N, int i) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < MAXR; ++k) g.a[q][k] = k < r ? ga_bh[(q * r + k) * N + i] : 0.f;
}
__device__ __forceinline__ void gates_at(float G[4], const RowGates &g, const float *gb_bh, int r, int N, int j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float z = 0.f;
#pragma unroll
        for (int k = 0; k < MAXR; ++k)
            if (k < r) z = fmaf(g.a[q][k], gb_bh[(q * r + k) * N + j], z);
        G[q] = sigmoidf_(z);
    }
}
// gates of edge (row, j): dense head reads the materialised maps, low-rank head evaluates sigmoid(a.b)
__device__ __forceinline__ void edge_gates(float G[4], const EwDims &d, const EwSaved &s, const RowGates &g, const float *gb_bh,
                                           int64_t row, int j) {
    if (d.dense) {
        const int64_t nn = d.BH * (int64_t)d.N * d.LD;
        for (int q = 0; q < 4; ++q) G[q] = s.G[q * nn + row * d.LD + j];
    } else {
        gates_at(G, g, gb_bh, d.r, d.N, j);
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
    if (!d.dense) load_row_gates(rg, s.ga + bh * 4 * d.r * d.N, d.r, d.N, i);
    const float *gb = s.gb + bh * 4 * d.r * d.N;
    const int64_t vstride = d.BH * (int64_t)d.N * d.LD;
    const float nb = a.beta_not / (float)max(1, d.V - 1);
    float *P = s.P + row * d.LD;
    const RowMask rmk{a.mask, a.mask_sb, a.mask_sh, a.mask_si, a.H, d.N};
    const uint8_t *mk = mask_row(rmk, row);
    float mx = -INFINITY;
    for (int j = lane; j < d.N; j += 64) {
        float sv[MAXV], O, lse, G[4];
        view_stats(s.S, vstride, row * d.LD + j, d.V, sv, O, lse);
        edge_gates(G, d, s, rg, gb, row, j);
        const float cr = s.Cr[row * d.LD + j];
        float sm = sv[0] + G[0] * O + G[1] * (lse - sv[0]) - G[2] * (nb * O) + G[3] * cr;
        if (mk && !mk[j]) sm = -INFINITY;                               // re-mask :549-550 (extension: see MopkEdgewiseArgs.mask)
        P[j] = sm; mx = fmaxf(mx, sm);
    }
    mx = wave_max(mx);
    float den = 0.f;
    for (int j = lane; j < d.N; j += 64) { float e = expf(P[j] - mx); P[j] = e; den += e; }
    den = wave_sum(den);
    const float inv = 1.f / den;
    for (int j = lane; j < d.N; j += 64) P[j] *= inv;
}

// attn_drop on the mixed attention weights (:552): out = in * keep / (1 - p) over one N x N plane per (b,h) -- the counter-based mask
// of the fused kernels (common.h), so a seed means one mask on either path.  In place when out == in.  One wave per row (bh, i).
__global__ void drop_plane_kernel(const float *in, float *out, FaDrop drop, int64_t rows, int N, int LD) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const uint32_t rowh = fa_drop_row(drop, (int)(row / N), (int)(row % N));
    for (int j = lane; j < N; j += 64) out[row * LD + j] = fa_drop_keep(drop, rowh, j) ? in[row * LD + j] * drop.inv_keep : 0.f;
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
    if (!d.dense) load_row_gates(rg, s.ga + bh * 4 * d.r * d.N, d.r, d.N, i);
    const float *gb = s.gb + bh * 4 * d.r * d.N;
    const int64_t vstride = d.BH * (int64_t)d.N * d.LD;
    const float nb = a.beta_not / (float)max(1, d.V - 1);
    float da[4][MAXR];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < MAXR; ++k) da[q][k] = 0.f;
    for (int j = lane; j < d.N; j += 64) {
        const float dsm = P[j] * (dP[j] - dot);
        dP[j] = dsm;
        float sv[MAXV], O, lse, G[4], tg[4];
        view_stats(s.S, vstride, row * d.LD + j, d.V, sv, O, lse);
        edge_gates(G, d, s, rg, gb, row, j);
        gate_terms(tg, sv, O, lse, nb, s.Cr[row * d.LD + j]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float dz = dsm * tg[q] * G[q] * (1.f - G[q]);
            if (d.dense) w.dZ[q * vstride + row * d.LD + j] = dz;      // dense head: the pre-sigmoid gradient map goes to the head's backward
            else {
#pragma unroll
                for (int k = 0; k < MAXR; ++k)
                    if (k < d.r) da[q][k] = fmaf(dz, gb[(q * d.r + k) * d.N + j], da[q][k]);
            }
        }
    }
    if (d.dense) return;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < MAXR; ++k)
            if (k < d.r) {
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
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < MAXR; ++k) { bj[q][k] = k < d.r ? gb[(q * d.r + k) * d.N + j] : 0.f; db[q][k] = 0.f; }
    const int64_t vstride = d.BH * (int64_t)d.N * d.LD;
    const float nb = a.beta_not / (float)max(1, d.V - 1);
    for (int i = 0; i < d.N; ++i) {
        const int64_t off = (bh * d.N + i) * d.LD + j;
        const float dsm = w.dP[off];
        float sv[MAXV], O, lse, tg[4];
        view_stats(s.S, vstride, off, d.V, sv, O, lse);
        gate_terms(tg, sv, O, lse, nb, s.Cr[off]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float z = 0.f, av[MAXR];
#pragma unroll
            for (int k = 0; k < MAXR; ++k) { av[k] = k < d.r ? ga[(q * d.r + k) * d.N + i] : 0.f; z = fmaf(av[k], bj[q][k], z); }
            const float G = sigmoidf_(z);
            const float dz = dsm * tg[q] * G * (1.f - G);
#pragma unroll
            for (int k = 0; k < MAXR; ++k) db[q][k] = fmaf(dz, av[k], db[q][k]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < MAXR; ++k)
            if (k < d.r) w.db[(bh * 4 * d.r + q * d.r + k) * d.N + j] = db[q][k];
}
// thread per (bh,n): d row_feat = Wr^T da, d col_feat = Wc^T db -> gradients of the 2V+2 means
__global__ void gate_ab_bwd_kernel(MopkEdgewiseArgs a, EwDims d, EwWork w) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.BH * d.N) return;
    const int n = idx % d.N;
    const int64_t bh = idx / d.N;
    float drow[MAXC], dcol[MAXC];
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
    for (int lv = 0; lv < d.L * d.V; ++lv) { w.drL[lv * n1 + idx] = drow[2 * d.V + 2 + lv]; w.dcL[lv * n1 + idx] = dcol[2 * d.V + 2 + lv]; }
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
    if (d.dense) {                                   // per-edge feature gradients replace the mean terms
        const int64_t nn = d.BH * (int64_t)d.N * d.LD;
        for (int j = lane; j < d.N; j += 64) {
            const int64_t off = row * d.LD + j;
            w.dCf[off] += (w.dP[off] * s.G[3 * nn + off] + w.dCrf[off]) * expf(-s.Cr[off]);
            w.dCb[off] = w.dClf[off] * expf(-s.Cl[off]);
        }
        return;
    }
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
// wave per (bh,i), all views: dS_v = A_v (dA_v - sum A_v dA_v) + dSmix coef_v + (drS_i + dcS_j)/N   (in place over dA).  The per-edge
// view statistics and gates are evaluated once and shared by the V views (one wave per (v,bh,i) evaluated them V times)
__global__ void ds_final_kernel(MopkEdgewiseArgs a, EwDims d, EwSaved s, EwWork w) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t rows_per_v = d.BH * d.N;
    if (row >= rows_per_v) return;
    const int lane = threadIdx.x & 63;
    const int i = row % d.N;
    const int64_t bh = row / d.N;
    const int64_t vstride = rows_per_v * d.LD;
    float dot[MAXV];
    for (int v = 0; v < d.V; ++v) {
        const float *A = s.A + v * vstride + row * d.LD;
        const float *dA = w.dA + v * vstride + row * d.LD;
        float t = 0.f;
        for (int j = lane; j < d.N; j += 64) t += A[j] * dA[j];
        dot[v] = wave_sum(t);
    }
    RowGates rg;
    if (!d.dense) load_row_gates(rg, s.ga + bh * 4 * d.r * d.N, d.r, d.N, i);
    const float *gb = s.gb + bh * 4 * d.r * d.N;
    const float nb = a.beta_not / (float)max(1, d.V - 1);
    const float invN = 1.f / d.N;
    const bool edge = d.dense || d.L > 0;            // gradients that reach S_v through per-edge features (dense head, lens bank)
    for (int j = lane; j < d.N; j += 64) {
        float sv[MAXV], O, lse, G[4];
        view_stats(s.S, vstride, row * d.LD + j, d.V, sv, O, lse);
        edge_gates(G, d, s, rg, gb, row, j);
        const float dp = w.dP[row * d.LD + j];
        for (int v = 0; v < d.V; ++v) {
            const int64_t o = v * vstride + row * d.LD + j;
            const float pi = expf(sv[v] - lse);
            const float coef = v == 0 ? (1.f - G[1] + G[1] * pi) : (G[0] - nb * G[2] + G[1] * pi);
            float t = s.A[o] * (w.dA[o] - dot[v]) + dp * coef + (d.dense ? 0.f : w.drS[v * rows_per_v + row] * invN);
            if (!d.dense) t += w.dcS[v * rows_per_v + bh * d.N + j] * invN;
            if (edge) t += w.dSf[o];
            w.dA[o] = t;
        }
    }
}
// ---- dense head backward (per pixel) ----
// d last = W2^T dZ ; without k3: dX1 = d last * gelu'(X1)
__global__ void dense_bwd_last_kernel(EwDims d, EwSaved s, EwWork w, EwExt e) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.N * d.N) return;
    const int i = p / d.N, j = p % d.N;
    const int64_t bh = blockIdx.y, pl = (int64_t)d.N * d.LD, nn = d.BH * pl, o = bh * pl + (int64_t)i * d.LD + j;
    float dz[4];
    for (int g = 0; g < 4; ++g) dz[g] = w.dZ[g * nn + o];
    for (int k = 0; k < HID; ++k) {
        float dl = 0.f;
        for (int g = 0; g < 4; ++g) dl = fmaf(e.W2[g * HID + k], dz[g], dl);
        if (d.k3) w.DX3[k * nn + o] = dl;
        else w.DX1[k * nn + o] = dl * gelu_tanh_grad(s.X1[k * nn + o]);
    }
}
// use_k3: dH2 = conv_transpose(dX3, W3) ; dX1 = dH2 * gelu'(gelu(X1)) * gelu'(X1).  Weights in LDS as [(k, tap)][c] (see dense_k3_fwd_kernel)
__global__ __launch_bounds__(256) void dense_k3_bwd_kernel(EwDims d, EwSaved s, EwWork w, EwExt e) {
    __shared__ __attribute__((aligned(16))) float W3s[HID * 9 * HID];
    for (int t = threadIdx.x; t < HID * HID * 9; t += 256) { const int k = t / (HID * 9), c = (t / 9) % HID, tap = t % 9; W3s[(k * 9 + tap) * HID + c] = e.W3[t]; }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.N * d.N) return;
    const int i = p / d.N, j = p % d.N;
    const int64_t bh = blockIdx.y, pl = (int64_t)d.N * d.LD, nn = d.BH * pl, o = bh * pl + (int64_t)i * d.LD + j;
    float dh[HID];
#pragma unroll
    for (int c = 0; c < HID; ++c) dh[c] = 0.f;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            const int ii = i - (a - 1), jj = j - (b - 1);
            if (ii < 0 || ii >= d.N || jj < 0 || jj >= d.N) continue;
            const int64_t q = bh * pl + (int64_t)ii * d.LD + jj;
            for (int k = 0; k < HID; ++k) {
                const float g = w.DX3[k * nn + q];
                const float4 *wr = (const float4 *)&W3s[(k * 9 + a * 3 + b) * HID];
#pragma unroll
                for (int c4 = 0; c4 < HID / 4; ++c4) {
                    const float4 wv = wr[c4];
                    dh[4 * c4] = fmaf(wv.x, g, dh[4 * c4]); dh[4 * c4 + 1] = fmaf(wv.y, g, dh[4 * c4 + 1]);
                    dh[4 * c4 + 2] = fmaf(wv.z, g, dh[4 * c4 + 2]); dh[4 * c4 + 3] = fmaf(wv.w, g, dh[4 * c4 + 3]);
                }
            }
        }
#pragma unroll
    for (int c = 0; c < HID; ++c) {
        const float x1 = s.X1[c * nn + o];
        w.DX1[c * nn + o] = dh[c] * gelu_tanh_grad(gelu_tanh(x1)) * gelu_tanh_grad(x1);
    }
}
// d feat = W1^T dX1 scattered to its sources: S_v(i,j), S_v(j,i) (dX1 of the transposed edge: tile pairs, see above), Cr, Cl, lens planes
__global__ __launch_bounds__(TP * TP) void dense_bwd_feat_kernel(EwDims d, EwWork w, EwExt e) {
    __shared__ float Ta[HID][TP][TP + 1], Tb[HID][TP][TP + 1];
    int ti, tj;
    pair_tiles(blockIdx.x, (d.N + TP - 1) / TP, ti, tj);
    const int64_t bh = blockIdx.y, pl = (int64_t)d.N * d.LD, nn = d.BH * pl;
    stage_tile(Ta, w.DX1, HID, nn, bh * pl, d.LD, d.N, ti * TP, tj * TP);
    if (ti != tj) stage_tile(Tb, w.DX1, HID, nn, bh * pl, d.LD, d.N, tj * TP, ti * TP);
    __syncthreads();
    const int lr = threadIdx.x / TP, lc = threadIdx.x % TP;
    for (int side = 0; side < (ti != tj ? 2 : 1); ++side) {
        const int i = (side ? tj : ti) * TP + lr, j = (side ? ti : tj) * TP + lc;
        if (i >= d.N || j >= d.N) continue;
        float (*Td)[TP][TP + 1] = side ? Tb : Ta, (*Tt)[TP][TP + 1] = (ti == tj) ? Ta : (side ? Ta : Tb);
        const int64_t o = bh * pl + (int64_t)i * d.LD + j;
        float g1[HID], g2[HID];
        for (int k = 0; k < HID; ++k) { g1[k] = Td[k][lr][lc]; g2[k] = Tt[k][lc][lr]; }
        for (int v = 0; v < d.V; ++v) {
            float t = 0.f;
            for (int k = 0; k < HID; ++k) t = fmaf(e.W1[k * d.C + v], g1[k], fmaf(e.W1[k * d.C + d.V + v], g2[k], t));
            w.dSf[v * nn + o] = t;
        }
        float tr = 0.f, tl = 0.f;
        for (int k = 0; k < HID; ++k) { tr = fmaf(e.W1[k * d.C + 2 * d.V], g1[k], tr); tl = fmaf(e.W1[k * d.C + 2 * d.V + 1], g1[k], tl); }
        w.dCrf[o] = tr; w.dClf[o] = tl;
        for (int lv = 0; lv < d.L * d.V; ++lv) {
            float t = 0.f;
            for (int k = 0; k < HID; ++k) t = fmaf(e.W1[k * d.C + 2 * d.V + 2 + lv], g1[k], t);
            w.dLz[lv * nn + o] = t;
        }
    }
}
// low-rank head + lens bank: the head only saw the planes' means -> dLz(i,j) = (drL_i + dcL_j) / N
__global__ void lens_dl_lowrank_kernel(EwDims d, EwWork w) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.N * d.N) return;
    const int i = p / d.N, j = p % d.N, lv = blockIdx.z;
    const int64_t bh = blockIdx.y, pl = (int64_t)d.N * d.LD, n1 = d.BH * d.N;
    w.dLz[(lv * d.BH + bh) * pl + (int64_t)i * d.LD + j] = (w.drL[lv * n1 + bh * d.N + i] + w.dcL[lv * n1 + bh * d.N + j]) / d.N;
}
// dSf_v (+)= sum_l conv_transpose(dLz[l,v], lens_w[l,v])
__global__ void lens_bwd_kernel(EwDims d, EwWork w, EwExt e) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.N * d.N) return;
    const int i = p / d.N, j = p % d.N, v = blockIdx.z;
    const int64_t bh = blockIdx.y, pl = (int64_t)d.N * d.LD, nn = d.BH * pl, o = bh * pl + (int64_t)i * d.LD + j;
    float acc = d.dense ? w.dSf[v * nn + o] : 0.f;
    for (int l = 0; l < d.L; ++l) {
        const int lv = l * d.V + v, dl = d.dil[l];
        const float *g = w.dLz + lv * nn + bh * pl;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                const int ii = i - (a - 1) * dl, jj = j - (b - 1) * dl;
                if (ii >= 0 && ii < d.N && jj >= 0 && jj < d.N) acc = fmaf(e.lens_w[(lv * 3 + a) * 3 + b], g[(int64_t)ii * d.LD + jj], acc);
            }
    }
    w.dSf[v * nn + o] = acc;
}
// dlens_w[lv,a,b] = sum dLz[lv](i,j) S_v(i+(a-1)d, j+(b-1)d): block (lv, chunk) -> 9 partial sums
__global__ void lens_w_bwd_kernel(EwDims d, EwSaved s, EwWork w) {
    __shared__ float red[9][256];
    const int lv = blockIdx.x, l = lv / d.V, v = lv % d.V, dl = d.dil[l];
    const int64_t pl = (int64_t)d.N * d.LD, nn = d.BH * pl, total = d.BH * d.N * d.N;
    float acc[9];
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (int64_t idx = (int64_t)blockIdx.y * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.y * 256) {
        const int j = idx % d.N, i = (idx / d.N) % d.N;
        const int64_t bh = idx / ((int64_t)d.N * d.N);
        const float g = w.dLz[lv * nn + bh * pl + (int64_t)i * d.LD + j];
        const float *Sv = s.S + v * nn + bh * pl;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                const int ii = i + (a - 1) * dl, jj = j + (b - 1) * dl;
                if (ii >= 0 && ii < d.N && jj >= 0 && jj < d.N) acc[a * 3 + b] = fmaf(g, Sv[(int64_t)ii * d.LD + jj], acc[a * 3 + b]);
            }
    }
    for (int t = 0; t < 9; ++t) red[t][threadIdx.x] = acc[t];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) for (int t = 0; t < 9; ++t) red[t][threadIdx.x] += red[t][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 9) w.part[((int64_t)lv * gridDim.y + blockIdx.y) * 9 + threadIdx.x] = red[threadIdx.x][0];
}
__global__ void lens_w_final_kernel(const float *part, float *out, int nlv, int nchunk) {      // fixed-order sum over chunks
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nlv * 9) return;
    const int lv = o / 9, t = o % 9;
    float sm = 0.f;
    for (int c = 0; c < nchunk; ++c) sm += part[((int64_t)lv * nchunk + c) * 9 + t];
    out[o] = sm;
}

// ---- weight-gradient reductions: out[a*nB + b] = sum over all pixels of A_a * B_b (last B channel == 1 -> the bias) ----
struct LdMaps {            // plane a of a (n,BH,N,LD) stack
    const float *base; int64_t nn; int N, LD;
    __device__ float operator()(int a, int64_t bh, int i, int j) const { return base[a * nn + (bh * N + i) * (int64_t)LD + j]; }
};
struct LdFeat1 {           // edge feature channels, then the constant 1
    EwDims d; EwSaved s;
    __device__ float operator()(int b, int64_t bh, int i, int j) const { return b < d.C ? edge_feat(d, s, b, bh, i, j) : 1.f; }
};
struct LdLast1 {           // input of conv2: X3 (use_k3) or gelu(X1); then the constant 1
    const float *X; int64_t nn; int N, LD, k3;
    __device__ float operator()(int b, int64_t bh, int i, int j) const {
        if (b >= HID) return 1.f;
        const float x = X[b * nn + (bh * N + i) * (int64_t)LD + j];
        return k3 ? x : gelu_tanh(x);
    }
};
// One tile = up to PR_PS pixels of one row (b,h,i): the A planes and the B channels go to LDS, then the 16 x nB sums of products
// run on the f32 matrix core (v_mfma_f32_16x16x4_f32: exact f32 fma chains).  Wave w takes the 4-pixel k-steps w, w+4, ... of
// every tile, so the four waves' partials are added once at the end, in wave order (bit-reproducible).  The next tile's values
// are fetched into registers while the matrix core works on the current one.
constexpr int PR_PS = 64, PR_LDS = PR_PS + 4;        // pixels per tile; LDS row stride (fragment reads touch all 64 banks)
struct PrTile { int64_t bh; int i, j0, jn; };
__device__ inline PrTile pr_tile(const EwDims &d, int64_t tile, int pc, int strips) {
    int64_t row, bh;
    if (tile < (1ll << 31)) {                 // the usual case: 32-bit divisions (a 64-bit one is a ~200-instruction routine, per tile)
        const unsigned int r32 = (unsigned int)tile / (unsigned int)strips;
        row = r32; bh = r32 / (unsigned int)d.N;
    } else { row = tile / strips; bh = row / d.N; }
    const int strip = (int)(tile - row * strips), j0 = strip * pc;
    return PrTile{bh, (int)(row - bh * d.N), j0, min(pc, d.N - j0)};
}
// the four waves' [a][b] accumulators -> part[block][a * nB + b]; S is reused as [wave][16][16 * NT]
template <int NT>
__device__ inline void pr_finish(float *S, const f32x4 (&acc)[NT], int nA, int nB, float *part) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lq = l >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
        for (int r = 0; r < 4; ++r) S[(w * HID + 4 * lq + r) * (16 * NT) + 16 * t + lr] = acc[t][r];
    __syncthreads();
    for (int o = tid; o < nA * nB; o += 256) {
        const int ai = o / nB, bi = o % nB;
        float sm = 0.f;
        for (int ww = 0; ww < 4; ++ww) sm += S[(ww * HID + ai) * (16 * NT) + bi];
        part[(int64_t)blockIdx.x * PR_MAXOUT + o] = sm;
    }
}
constexpr int PG_NT = 2, PG_ROWS = HID + 16 * PG_NT, PG_IT = (PG_ROWS * PR_PS + 255) / 256;   // nB <= 32 (conv1: 2V+3, conv2: 17)
static_assert(2 * MAXV + 3 <= 16 * PG_NT && HID + 1 <= 16 * PG_NT, "pair_reduce_kernel holds two 16-channel B tiles");
static_assert(4 * HID * 16 * PG_NT <= PG_ROWS * PR_LDS, "the end-of-kernel wave reduction reuses the staging buffer");
template <typename LA, typename LB>
__global__ __launch_bounds__(256) void pair_reduce_kernel(EwDims d, LA la, LB lb, int nA, int nB, int pc, int strips, float *part) {
    __shared__ float S[PG_ROWS * PR_LDS];            // rows 0..15: A planes; 16..: B channels
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lq = l >> 4;
    const int nT = (nB + 15) / 16;
    const int64_t tiles = d.BH * d.N * strips;
    f32x4 acc[PG_NT];
    float v[PG_IT];
    for (int t = 0; t < PG_NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int idx = tid; idx < PG_ROWS * PR_LDS; idx += 256) S[idx] = 0.f;
    auto fetch = [&](const PrTile &t) {
#pragma unroll
        for (int it = 0; it < PG_IT; ++it) {
            const int idx = tid + 256 * it, ch = idx / PR_PS, pp = idx % PR_PS;
            v[it] = 0.f;
            if (ch < nA + nB && pp < t.jn) v[it] = ch < nA ? la(ch, t.bh, t.i, t.j0 + pp) : lb(ch - nA, t.bh, t.i, t.j0 + pp);
        }
    };
    int64_t tile = blockIdx.x;
    PrTile cur = pr_tile(d, tile < tiles ? tile : 0, pc, strips);
    if (tile < tiles) fetch(cur);
    __syncthreads();
    for (; tile < tiles; tile += gridDim.x) {
#pragma unroll
        for (int it = 0; it < PG_IT; ++it) {
            const int idx = tid + 256 * it, ch = idx / PR_PS, pp = idx % PR_PS;
            if (ch < nA + nB) S[(ch < nA ? ch : HID + ch - nA) * PR_LDS + pp] = v[it];
        }
        __syncthreads();
        const int jn = cur.jn;
        if (tile + gridDim.x < tiles) { cur = pr_tile(d, tile + gridDim.x, pc, strips); fetch(cur); }
        for (int k = w; 4 * k < jn; k += 4) {
            const float av = S[lr * PR_LDS + 4 * k + lq];
#pragma unroll
            for (int t = 0; t < PG_NT; ++t)
                if (t < nT) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, S[(HID + 16 * t + lr) * PR_LDS + 4 * k + lq], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
    pr_finish<PG_NT>(S, acc, nA, nB, part);
}
// dW3 | db3: the 144 tap channels are shifted copies of the 16 H2 planes, so a tile stages the three H2 rows i-1, i, i+1 with a
// one-pixel halo (16 x 3 x 66 loads instead of 144 x 64) and every B fragment is a shifted LDS read.  Same tiling, k-step
// split and end-of-kernel wave reduction as pair_reduce_kernel; output layout part[a * 145 + c * 9 + ta * 3 + tb], 144 -> db3.
constexpr int TR_NT = (HID * 9 + 1 + 15) / 16, TR_HW = PR_PS + 2, TR_A = 0, TR_H = HID * PR_LDS, TR_ONE = TR_H + HID * 3 * PR_LDS,
              TR_ZERO = TR_ONE + PR_LDS, TR_STAGE = TR_ZERO + PR_LDS, TR_RED = 4 * HID * 16 * TR_NT,
              TR_LDS = TR_STAGE > TR_RED ? TR_STAGE : TR_RED, TR_AIT = HID * PR_PS / 256, TR_HIT = (HID * 3 * TR_HW + 255) / 256;
__global__ __launch_bounds__(256) void taps_reduce_kernel(EwDims d, const float *X3, const float *H2, int64_t nn, int pc, int strips, float *part) {
    __shared__ float S[TR_LDS];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lq = l >> 4;
    const int64_t tiles = d.BH * d.N * strips;
    int boff[TR_NT];
    f32x4 acc[TR_NT];
    struct Regs { float va[TR_AIT], vh[TR_HIT]; int jn; };
    Regs R0, R1;                             // two tiles in flight: a tile's loads have two iterations (~2.5 k cycles of MFMAs) to land
#pragma unroll
    for (int t = 0; t < TR_NT; ++t) {
        const int b = 16 * t + lr, c = b / 9, tap = b % 9;
        boff[t] = (b < HID * 9 ? TR_H + (c * 3 + tap / 3) * PR_LDS + tap % 3 : b == HID * 9 ? TR_ONE : TR_ZERO) + lq;
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int idx = tid; idx < TR_STAGE; idx += 256) S[idx] = idx >= TR_ONE && idx < TR_ZERO ? 1.f : 0.f;
    // per-thread constants of the staging pattern: element (channel, row offset, pixel) of slot `it` does not depend on the tile
    int64_t offa[TR_AIT], offh[TR_HIT];
    int ppa[TR_AIT], pph[TR_HIT], drh[TR_HIT];
#pragma unroll
    for (int it = 0; it < TR_AIT; ++it) { const int idx = tid + 256 * it; ppa[it] = idx % PR_PS; offa[it] = (idx / PR_PS) * nn + ppa[it]; }
#pragma unroll
    for (int it = 0; it < TR_HIT; ++it) {
        const int idx = tid + 256 * it, cr = idx / TR_HW;
        pph[it] = cr < HID * 3 ? idx % TR_HW : 1 << 20;              // slots past the 48 rows never pass the bounds test
        drh[it] = cr % 3 - 1;
        offh[it] = (cr / 3) * nn + (int64_t)drh[it] * d.LD + pph[it] - 1;
    }
    auto fetch = [&](Regs &R, int64_t tile) {
        const PrTile t = pr_tile(d, tile, pc, strips);
        R.jn = t.jn;
        const float *x3 = X3 + (t.bh * d.N + t.i) * (int64_t)d.LD + t.j0, *h2 = H2 + (t.bh * d.N + t.i) * (int64_t)d.LD + t.j0;
#pragma unroll
        for (int it = 0; it < TR_AIT; ++it) R.va[it] = ppa[it] < t.jn ? x3[offa[it]] : 0.f;
#pragma unroll
        for (int it = 0; it < TR_HIT; ++it) {
            const int ii = t.i + drh[it], jj = t.j0 + pph[it] - 1;
            R.vh[it] = 0.f;
            if (pph[it] < t.jn + 2 && ii >= 0 && ii < d.N && jj >= 0 && jj < d.N) R.vh[it] = h2[offh[it]];
        }
    };
    // one tile: registers -> LDS, barrier, refill the registers with the tile two steps ahead, MFMAs, barrier
    auto step = [&](Regs &R, int64_t ahead) {
#pragma unroll
        for (int it = 0; it < TR_AIT; ++it) {
            const int idx = tid + 256 * it;
            S[TR_A + (idx / PR_PS) * PR_LDS + idx % PR_PS] = R.va[it];
        }
#pragma unroll
        for (int it = 0; it < TR_HIT; ++it) {
            const int idx = tid + 256 * it, cr = idx / TR_HW;
            if (cr < HID * 3) S[TR_H + cr * PR_LDS + idx % TR_HW] = R.vh[it];
        }
        __syncthreads();
        const int jn = R.jn;
        if (ahead < tiles) fetch(R, ahead);
        // k-steps w, w + 4, w + 8 unconditionally (pixels past jn are staged as zeros), w + 12 when the strip is that long: straight-line
        // code, so the LDS reads of a later step are in flight under the MFMAs of an earlier one
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int k = w + 4 * u;
            const float av = S[TR_A + lr * PR_LDS + 4 * k + lq];
#pragma unroll
            for (int t = 0; t < TR_NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, S[boff[t] + 4 * k], acc[t], 0, 0, 0);
        }
        if (4 * (w + 12) < jn) {
            const int k = w + 12;
            const float av = S[TR_A + lr * PR_LDS + 4 * k + lq];
#pragma unroll
            for (int t = 0; t < TR_NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, S[boff[t] + 4 * k], acc[t], 0, 0, 0);
        }
        __syncthreads();
    };
    const int64_t g = gridDim.x;
    int64_t tile = blockIdx.x;
    if (tile < tiles) fetch(R0, tile);
    if (tile + g < tiles) fetch(R1, tile + g);
    __syncthreads();
    for (; tile < tiles; tile += 2 * g) {
        step(R0, tile + 2 * g);
        if (tile + g < tiles) step(R1, tile + 3 * g);
    }
    pr_finish<TR_NT>(S, acc, HID, HID * 9 + 1, part);
}
// fixed-order sum over the blocks' partials; b < nB-1 -> dW[a*(nB-1)+b], b == nB-1 -> dbias[a]
__global__ void pair_final_kernel(const float *part, int nblk, int nA, int nB, float *dW, float *dbias) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nA * nB) return;
    float sm = 0.f;
    for (int g = 0; g < nblk; ++g) sm += part[(int64_t)g * PR_MAXOUT + o];
    const int ai = o / nB, bi = o % nB;
    if (bi < nB - 1) { if (dW) dW[ai * (nB - 1) + bi] = sm; }
    else if (dbias) dbias[ai] = sm;
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
    if (d.V > MAXV || d.r > MAXR || d.V < 2 || d.r < 1 || d.L < 0 || d.L > MAXL) return MOPK_ERR_UNSUPPORTED;
    const EwSaved s = ew_carve_saved(a->saved, d, nullptr);
    const EwWork w = ew_carve_work(a->workspace, d, nullptr);
    const EwExt e = ew_ext(a);
    const bool mf = a->precision == MOPK_PREC_BF16;
    const int N = d.N, dk = d.dk, LD = d.LD, V = d.V;
    const int64_t nd1 = (int64_t)N * dk, nn1 = (int64_t)N * LD, ndv = d.BH * nd1, nnv = d.BH * nn1;
    const int64_t tot = d.BH * nd1;
    const int BHi = (int)d.BH;
    const dim3 pix((N * N + 255) / 256, BHi);        // thread per edge (i,j) of one (b,h)
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
    const RowMask rm{a->mask, a->mask_sb, a->mask_sh, a->mask_si, a->H, N};
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rowsV + 3) / 4), dim3(256), 0, st, s.S, s.A, s.rS, rowsV, N, LD, rm, rows1);  // :504-507
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
    if (d.L > 0) {                                                                                                    // :523-533
        const int LV = d.L * V;
        hipLaunchKernelGGL(lens_fwd_kernel, dim3(pix.x, BHi, LV), dim3(256), 0, st, d, s, e);
        if (!d.dense) {
            hipLaunchKernelGGL(rowmean_kernel, dim3((LV * rows1 + 3) / 4), dim3(256), 0, st, s.Lz, s.rL, LV * rows1, N, LD);
            hipLaunchKernelGGL(colmean_kernel, dim3((N + 255) / 256, LV * BHi), dim3(256), 0, st, s.Lz, s.cL, N, LD);
        }
    }
    if (d.dense) {                                                                                                    // :312-318
        hipLaunchKernelGGL(dense_gate_fwd_kernel, pair_grid(d.N, (int)d.BH), dim3(TP * TP), 0, st, d, s, e);
        if (d.k3) hipLaunchKernelGGL(dense_k3_fwd_kernel, pix, dim3(256), 0, st, d, s, e);
    } else {
        hipLaunchKernelGGL(gate_ab_kernel, dim3((rows1 + 255) / 256), dim3(256), 0, st, *a, d, s);                    // :325-326
    }
    hipLaunchKernelGGL(mix_fwd_kernel, dim3((rows1 + 3) / 4), dim3(256), 0, st, *a, d, s);                            // :537-551
    MOPK_CHECK_LAUNCH();
    {   // y_base = P V0 ; y_chain = A0(A1(...(A_{V-1} VL))) = Cf VL      :554-560
        GemmDesc g = gd(N, dk, N, 1, BHi);
        const float *Pm = s.P;
        if (a->dropout_p > 0.f) {        // the dropped weights go to a plane of the (backward's) workspace the forward does not use; P itself stays in `saved`
            hipLaunchKernelGGL(drop_plane_kernel, dim3((rows1 + 3) / 4), dim3(256), 0, st, s.P, w.dP, fa_drop(a->dropout_p, a->dropout_seed), rows1, N, LD);
            Pm = w.dP;
        }
        g.A = Pm; g.a_rs = LD; g.a_cs = 1; g.a_b1 = nn1;
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
    if (d.V > MAXV || d.r > MAXR || d.V < 2 || d.r < 1 || d.L < 0 || d.L > MAXL) return MOPK_ERR_UNSUPPORTED;
    const EwSaved s = ew_carve_saved(a->saved, d, nullptr);
    const EwWork w = ew_carve_work(a->workspace, d, nullptr);
    const EwExt e = ew_ext(a);
    const bool mf = a->precision == MOPK_PREC_BF16;
    const int N = d.N, dk = d.dk, LD = d.LD, V = d.V;
    const int64_t nd1 = (int64_t)N * dk, nn1 = (int64_t)N * LD, ndv = d.BH * nd1, nnv = d.BH * nn1;
    const int64_t tot = d.BH * nd1, rows1 = d.BH * N, rowsV = (int64_t)V * rows1;
    const int BHi = (int)d.BH;
    const dim3 pix((N * N + 255) / 256, BHi);
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
        const float *Pm = s.P;
        if (a->dropout_p > 0.f) {        // attn_drop: dP = (dy V0^T) keep / (1 - p), and dV0 sees the dropped weights (rebuilt in a plane the D chain uses later)
            const FaDrop drop = fa_drop(a->dropout_p, a->dropout_seed);
            hipLaunchKernelGGL(drop_plane_kernel, dim3((rows1 + 3) / 4), dim3(256), 0, st, w.dP, w.dP, drop, rows1, N, LD);
            hipLaunchKernelGGL(drop_plane_kernel, dim3((rows1 + 3) / 4), dim3(256), 0, st, s.P, w.D0, drop, rows1, N, LD);
            Pm = w.D0;
        }
        g.B = s.VL; g.C = w.dCf; g.alpha_dev = s.wsig;   // dCf = w dy VL^T
        RET_IF(bgemm(g, mf, st));
        GemmDesc h = gd(N, dk, N, 1, BHi);           // dV0 = P^T dy
        h.A = Pm; h.a_rs = 1; h.a_cs = LD; h.a_b1 = nn1;
        h.B = w.dyc; h.b_rs = dk; h.b_cs = 1; h.b_b1 = nd1;
        h.C = w.dV0; h.c_rs = dk; h.c_b1 = nd1;
        RET_IF(bgemm(h, mf, st));
        h.A = Cf; h.C = w.dVL; h.alpha_dev = s.wsig;  // dVL = w Cf^T dy
        RET_IF(bgemm(h, mf, st));
    }
    hipLaunchKernelGGL(mix_bwd_rows_kernel, dim3((rows1 + 3) / 4), dim3(256), 0, st, *a, d, s, w);
    if (d.dense) {
        hipLaunchKernelGGL(dense_bwd_last_kernel, pix, dim3(256), 0, st, d, s, w, e);
        if (d.k3) hipLaunchKernelGGL(dense_k3_bwd_kernel, pix, dim3(256), 0, st, d, s, w, e);
        hipLaunchKernelGGL(dense_bwd_feat_kernel, pair_grid(N, BHi), dim3(TP * TP), 0, st, d, w, e);
        MOPK_CHECK_LAUNCH();
        // weight gradients: all-pairs pixel reductions, partials per block then a fixed-order sum (bit-reproducible)
        const int strips = (N + PR_PS - 1) / PR_PS, pc = ((N + strips - 1) / strips + 3) & ~3;   // N=197: 4 strips of 52
        const int64_t tiles = d.BH * N * strips;
        const int nblk = (int)(tiles < PR_BLOCKS ? tiles : PR_BLOCKS);
        const LdMaps mZ{w.dZ, nnv, N, LD}, mX1{w.DX1, nnv, N, LD};
        hipLaunchKernelGGL((pair_reduce_kernel<LdMaps, LdLast1>), dim3(nblk), dim3(256), 0, st, d, mZ,
                           LdLast1{d.k3 ? s.X3 : s.X1, nnv, N, LD, d.k3}, 4, HID + 1, pc, strips, w.part);
        hipLaunchKernelGGL(pair_final_kernel, dim3((4 * (HID + 1) + 255) / 256), dim3(256), 0, st, w.part, nblk, 4, HID + 1, e.dW2, e.db2);
        if (d.k3) {
            const int nb3 = nblk < 512 ? nblk : 512;      // 133 VGPRs: two blocks per CU
            hipLaunchKernelGGL(taps_reduce_kernel, dim3(nb3), dim3(256), 0, st, d, w.DX3, s.H2, nnv, pc, strips, w.part);
            hipLaunchKernelGGL(pair_final_kernel, dim3((HID * (HID * 9 + 1) + 255) / 256), dim3(256), 0, st, w.part, nb3, HID, HID * 9 + 1, e.dW3, e.db3);
        }
        hipLaunchKernelGGL((pair_reduce_kernel<LdMaps, LdFeat1>), dim3(nblk), dim3(256), 0, st, d, mX1, LdFeat1{d, s}, HID, d.C + 1, pc, strips, w.part);
        hipLaunchKernelGGL(pair_final_kernel, dim3((HID * (d.C + 1) + 255) / 256), dim3(256), 0, st, w.part, nblk, HID, d.C + 1, e.dW1, e.db1);
    } else {
        hipLaunchKernelGGL(mix_bwd_cols_kernel, dim3((N + 63) / 64, BHi), dim3(64), 0, st, *a, d, s, w);
        hipLaunchKernelGGL(gate_ab_bwd_kernel, dim3((rows1 + 255) / 256), dim3(256), 0, st, *a, d, w);
        hipLaunchKernelGGL(gate_w_bwd_kernel, dim3(4 * d.r, d.C + 1, 2), dim3(256), 0, st, *a, d, s, w);
        if (d.L > 0) hipLaunchKernelGGL(lens_dl_lowrank_kernel, dim3(pix.x, BHi, d.L * V), dim3(256), 0, st, d, w);
    }
    if (d.L > 0) {
        const int LV = d.L * V, nchunk = 64;
        hipLaunchKernelGGL(lens_bwd_kernel, dim3(pix.x, BHi, V), dim3(256), 0, st, d, w, e);
        hipLaunchKernelGGL(lens_w_bwd_kernel, dim3(LV, nchunk), dim3(256), 0, st, d, s, w);
        hipLaunchKernelGGL(lens_w_final_kernel, dim3((LV * 9 + 63) / 64), dim3(64), 0, st, w.part, e.dlens_w, LV, nchunk);
    }
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
    hipLaunchKernelGGL(ds_final_kernel, dim3((rows1 + 3) / 4), dim3(256), 0, st, *a, d, s, w);
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

// ---- batch reduction of the small per-(b) partial gradients: 16 outputs x 16 batch slices per block, slices combined in a
// fixed order; the last block sums dlogit_part
__global__ void __launch_bounds__(256) ew_reduce_parts_kernel(const float *dsqk_p, const float *dvs0_p, const float *dvsL_p,
                                                             const float *dlg_p, float *dsqk, float *dvs0, float *dvsL,
                                                             float *dlogit, int B, int n_sqk, int n_vs, int n_lg) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    if (blockIdx.x == gridDim.x - 1) {
        float s = 0.f;
        for (int i = tid; i < n_lg; i += 256) s += dlg_p[i];
        red[tid] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        if (tid == 0) *dlogit = red[0];
        return;
    }
    const int o = blockIdx.x * 16 + (tid & 15), sl = tid >> 4, total = n_sqk + 2 * n_vs;
    float s = 0.f;
    if (o < total) {
        const float *src; int w, c;
        if (o < n_sqk) { src = dsqk_p; w = n_sqk; c = o; }
        else if (o < n_sqk + n_vs) { src = dvs0_p; w = n_vs; c = o - n_sqk; }
        else { src = dvsL_p; w = n_vs; c = o - n_sqk - n_vs; }
#pragma unroll 4
        for (int b = sl; b < B; b += 16) s += src[(size_t)b * w + c];
    }
    red[tid] = s;
    __syncthreads();
    if (sl == 0 && o < total) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k * 16 + tid];
        if (o < n_sqk) dsqk[o] = t;
        else if (o < n_sqk + n_vs) dvs0[o - n_sqk] = t;
        else dvsL[o - n_sqk - n_vs] = t;
    }
}
int ew_reduce_parts(const MopkEdgewiseArgs *a, float *dsqk, float *dvs0, float *dvsL, float *dlogit, hipStream_t st) {
    const int n_sqk = a->V * a->H * a->dk, n_vs = a->H * a->dk, total = n_sqk + 2 * n_vs;
    hipLaunchKernelGGL(ew_reduce_parts_kernel, dim3((total + 15) / 16 + 1), dim3(256), 0, st, a->dsqk_part, a->dvs0_part,
                       a->dvsL_part, a->dlogit_part, dsqk, dvs0, dvsL, dlogit, a->B, n_sqk, n_vs, a->B * a->H);
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
