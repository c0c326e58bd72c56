// Generic multi-kernel paths for the sibling attention cores (any N / dk):
//   * plain SDPA            (reference attention_variants.py:42-46, components.py:61-64, whisper_mop.py:163-175)
//   * MultiHopMSA dual-path (reference attention_variants.py:200-229)
//   * Quartet causal attn   (reference quartet_attn_patch.py:88-121)
//   * CrossViewMixerMSA     (reference attention_variants.py:90-153)
// Contractions are bgemm (fp32 FMA or bf16 MFMA); maps live in fp32 in `saved`/`workspace`.
// Backward formulas are the hand-derived ones pinned in oracle/{sdpa,multihop,quartet}.py.
#include "bgemm.h"
#include "common.h"

namespace mopk {

#define RET_IF(x) do { int rc_ = (x); if (rc_ != MOPK_OK) return rc_; } while (0)
constexpr float EPS_CH = 1e-6f;

struct Dm { int B, H, N, dk, LD; int64_t BH; };
static Dm mkdm(int B, int H, int N, int dk) { Dm d{B, H, N, dk, (int)round_up(N, 4), (int64_t)B * H}; return d; }

template <typename T>
__global__ void gather_kernel(MopkView4 v, float *out, Dm d) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.BH * d.N * d.dk) return;
    const int dd = idx % d.dk; const int n = (idx / d.dk) % d.N; const int64_t bh = idx / ((int64_t)d.dk * d.N);
    out[idx] = ld_as_f32((const T *)v.ptr + (bh / d.H) * v.sb + (bh % d.H) * v.sh + n * v.sn + dd);
}
template <typename T>
__global__ void scatter_kernel(const float *in, MopkView4 v, Dm d) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.BH * d.N * d.dk) return;
    const int dd = idx % d.dk; const int n = (idx / d.dk) % d.N; const int64_t bh = idx / ((int64_t)d.dk * d.N);
    st_from_f32((T *)v.ptr + (bh / d.H) * v.sb + (bh % d.H) * v.sh + n * v.sn + dd, in[idx]);
}
static int gather(int io, const MopkView4 &v, float *out, const Dm &d, hipStream_t st) {
    const int64_t tot = d.BH * d.N * d.dk;
    if (io == MOPK_BF16) hipLaunchKernelGGL((gather_kernel<unsigned short>), dim3((tot + 255) / 256), dim3(256), 0, st, v, out, d);
    else hipLaunchKernelGGL((gather_kernel<float>), dim3((tot + 255) / 256), dim3(256), 0, st, v, out, d);
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}
static int scatter(int io, const float *in, const MopkView4 &v, const Dm &d, hipStream_t st) {
    const int64_t tot = d.BH * d.N * d.dk;
    if (io == MOPK_BF16) hipLaunchKernelGGL((scatter_kernel<unsigned short>), dim3((tot + 255) / 256), dim3(256), 0, st, in, v, d);
    else hipLaunchKernelGGL((scatter_kernel<float>), dim3((tot + 255) / 256), dim3(256), 0, st, in, v, d);
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}

struct MaskSpec {
    int causal;
    const uint8_t *mask; int64_t msb, msh, msi;     // 1 = keep
    const float *bias; int64_t bsb, bsh, bsi;       // additive
};
__device__ __forceinline__ bool is_blocked(const MaskSpec &m, int b, int h, int i, int j) {
    if (m.causal && j > i) return true;
    if (m.mask && m.mask[b * m.msb + h * m.msh + i * m.msi + j] == 0) return true;
    return false;
}
// in-place-capable: out = softmax_j(in + bias) over non-blocked j ; one wave per row
__global__ void masked_softmax_kernel(const float *in, float *out, Dm d, MaskSpec m) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63, i = row % d.N;
    const int64_t bh = row / d.N;
    const int b = bh / d.H, h = bh % d.H;
    const float *p = in + row * d.LD;
    float *o = out + row * d.LD;
    const float *bp = m.bias ? m.bias + b * m.bsb + h * m.bsh + i * m.bsi : nullptr;
    float mx = -INFINITY;
    for (int j = lane; j < d.N; j += 64) if (!is_blocked(m, b, h, i, j)) mx = fmaxf(mx, p[j] + (bp ? bp[j] : 0.f));
    mx = wave_max(mx);
    float den = 0.f;
    for (int j = lane; j < d.N; j += 64) if (!is_blocked(m, b, h, i, j)) den += expf(p[j] + (bp ? bp[j] : 0.f) - mx);
    den = wave_sum(den);
    const float inv = 1.f / den;
    for (int j = lane; j < d.N; j += 64)
        o[j] = is_blocked(m, b, h, i, j) ? 0.f : expf(p[j] + (bp ? bp[j] : 0.f) - mx) * inv;
}
// dS = P (dP - sum_j P dP) * alpha, in place over dP ; one wave per row
__global__ void softmax_bwd_kernel(const float *P, float *dP, Dm d, float alpha) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63;
    const float *p = P + row * d.LD;
    float *g = dP + row * d.LD;
    float dot = 0.f;
    for (int j = lane; j < d.N; j += 64) dot += p[j] * g[j];
    dot = wave_sum(dot);
    for (int j = lane; j < d.N; j += 64) g[j] = p[j] * (g[j] - dot) * alpha;
}

static inline GemmDesc gd0(int M, int N, int K, int nb) {
    GemmDesc g{}; g.M = M; g.N = N; g.K = K; g.nb0 = 1; g.nb1 = nb; g.alpha = 1.f; g.beta = 0.f; g.alpha_dev = nullptr;
    return g;
}
// C(N,N) = alpha * X(N,dk) Y(N,dk)^T
// attention dropout (`self.attn_drop(A)`): out = in * keep / (1 - p) over one N x N map per (b,h), with the counter-based mask of the
// fused kernels (common.h: a seed means one mask on either path).  In place when out == in.  One wave per row (bh, i).
__global__ void drop_rows_kernel(const float *in, int ld_in, float *out, int ld_out, FaDrop drop, int64_t rows, int N) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const uint32_t rowh = fa_drop_row(drop, (int)(row / N), (int)(row % N));
    for (int j = lane; j < N; j += 64) out[row * ld_out + j] = fa_drop_keep(drop, rowh, j) ? in[row * ld_in + j] * drop.inv_keep : 0.f;
}
static void drop_map(const float *in, float *out, const Dm &d, float p, uint64_t seed, hipStream_t st) {
    const int64_t rows = d.BH * d.N;
    hipLaunchKernelGGL(drop_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, in, d.LD, out, d.LD, fa_drop(p, seed), rows, d.N);
}
// The backward of  y = drop(P) v :  dv = drop(P)^T dy  and  dP = (dy v^T) keep / (1 - p).  The dropped map is rebuilt in the dP plane
// for the dv product BEFORE that plane receives dP (no extra workspace); the softmax backward then sees the undropped P and the
// masked dP, as the reference's autograd does.
static int drop_bwd_pair(const float *P, const float *dy, const float *v, float *dP, float *dv, const Dm &d, float p, uint64_t seed, bool mf,
                         hipStream_t st);

static int gemm_nt_scores(const float *X, const float *Y, float *C, const Dm &d, float alpha, bool mf, hipStream_t st, const float *adev = nullptr) {
    GemmDesc g = gd0(d.N, d.N, d.dk, (int)d.BH);
    g.A = X; g.a_rs = d.dk; g.a_cs = 1; g.a_b1 = (int64_t)d.N * d.dk;
    g.B = Y; g.b_rs = 1; g.b_cs = d.dk; g.b_b1 = (int64_t)d.N * d.dk;
    g.C = C; g.c_rs = d.LD; g.c_b1 = (int64_t)d.N * d.LD; g.alpha = alpha; g.alpha_dev = adev;
    return bgemm(g, mf, st);
}
// C(N,dk) = alpha * op(M)(N,N) X(N,dk) + beta C ; trans: use M^T
static int gemm_map_vec(const float *M, bool trans, const float *X, float *C, const Dm &d, float alpha, float beta, bool mf,
                        hipStream_t st, const float *adev = nullptr) {
    GemmDesc g = gd0(d.N, d.dk, d.N, (int)d.BH);
    g.A = M; g.a_rs = trans ? 1 : d.LD; g.a_cs = trans ? d.LD : 1; g.a_b1 = (int64_t)d.N * d.LD;
    g.B = X; g.b_rs = d.dk; g.b_cs = 1; g.b_b1 = (int64_t)d.N * d.dk;
    g.C = C; g.c_rs = d.dk; g.c_b1 = (int64_t)d.N * d.dk; g.alpha = alpha; g.beta = beta; g.alpha_dev = adev;
    return bgemm(g, mf, st);
}
// C(N,N) = op(A)(N,N) op(B)(N,N) + beta C
static int gemm_map_map(const float *A, bool ta, const float *B, bool tb, float *C, const Dm &d, float beta, bool mf, hipStream_t st) {
    GemmDesc g = gd0(d.N, d.N, d.N, (int)d.BH);
    g.A = A; g.a_rs = ta ? 1 : d.LD; g.a_cs = ta ? d.LD : 1; g.a_b1 = (int64_t)d.N * d.LD;
    g.B = B; g.b_rs = tb ? 1 : d.LD; g.b_cs = tb ? d.LD : 1; g.b_b1 = (int64_t)d.N * d.LD;
    g.C = C; g.c_rs = d.LD; g.c_b1 = (int64_t)d.N * d.LD; g.beta = beta;
    return bgemm(g, mf, st);
}

static int drop_bwd_pair(const float *P, const float *dy, const float *v, float *dP, float *dv, const Dm &d, float p, uint64_t seed, bool mf,
                         hipStream_t st) {
    if (p > 0.f) {
        drop_map(P, dP, d, p, seed, st);
        RET_IF(gemm_map_vec(dP, true, dy, dv, d, 1.f, 0.f, mf, st));
        RET_IF(gemm_nt_scores(dy, v, dP, d, 1.f, mf, st));
        drop_map(dP, dP, d, p, seed, st);
        return MOPK_OK;
    }
    RET_IF(gemm_nt_scores(dy, v, dP, d, 1.f, mf, st));
    return gemm_map_vec(P, true, dy, dv, d, 1.f, 0.f, mf, st);
}

// =====================================================================  SDPA
struct SdpaBuf { float *q, *k, *v, *P, *y, *dy, *dP, *dq, *dk, *dv; };
static SdpaBuf sdpa_carve(void *saved, void *ws, const Dm &d, size_t *ns, size_t *nw) {
    Carver cs(saved), cw(ws);
    const size_t nd = d.BH * d.N * d.dk, nn = d.BH * (size_t)d.N * d.LD;
    SdpaBuf b;
    b.q = cs.take<float>(nd); b.k = cs.take<float>(nd); b.v = cs.take<float>(nd); b.P = cs.take<float>(nn);
    b.y = cw.take<float>(nd); b.dy = cw.take<float>(nd); b.dP = cw.take<float>(nn);
    b.dq = cw.take<float>(nd); b.dk = cw.take<float>(nd); b.dv = cw.take<float>(nd);
    if (ns) *ns = cs.off; if (nw) *nw = cw.off;
    return b;
}
size_t sdpa_saved_bytes(const MopkSdpaArgs *a) { size_t s; sdpa_carve(nullptr, nullptr, mkdm(a->B, a->H, a->N, a->dk), &s, nullptr); return s; }
size_t sdpa_ws_bytes(const MopkSdpaArgs *a) { size_t w; sdpa_carve(nullptr, nullptr, mkdm(a->B, a->H, a->N, a->dk), nullptr, &w); return w; }
int sdpa_fwd(const MopkSdpaArgs *a, hipStream_t st) {
    const Dm d = mkdm(a->B, a->H, a->N, a->dk);
    const SdpaBuf b = sdpa_carve(a->saved, a->workspace, d, nullptr, nullptr);
    const bool mf = a->precision == MOPK_PREC_BF16;
    RET_IF(gather(a->io_dtype, a->q, b.q, d, st)); RET_IF(gather(a->io_dtype, a->k, b.k, d, st)); RET_IF(gather(a->io_dtype, a->v, b.v, d, st));
    RET_IF(gemm_nt_scores(b.q, b.k, b.P, d, 1.f / sqrtf((float)d.dk), mf, st));
    const MaskSpec m{a->causal, a->mask, a->mask_sb, a->mask_sh, a->mask_si, a->bias, a->bias_sb, a->bias_sh, a->bias_si};
    const int64_t rows = d.BH * d.N;
    hipLaunchKernelGGL(masked_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.P, b.P, d, m);
    if (a->dropout_p > 0.f) drop_map(b.P, b.dP, d, a->dropout_p, a->dropout_seed, st);     // dropped weights in a workspace plane; P stays in `saved`
    MOPK_CHECK_LAUNCH();
    RET_IF(gemm_map_vec(a->dropout_p > 0.f ? b.dP : b.P, false, b.v, b.y, d, 1.f, 0.f, mf, st));
    return scatter(a->io_dtype, b.y, a->y, d, st);
}
int sdpa_bwd(const MopkSdpaArgs *a, hipStream_t st) {
    const Dm d = mkdm(a->B, a->H, a->N, a->dk);
    const SdpaBuf b = sdpa_carve(a->saved, a->workspace, d, nullptr, nullptr);
    const bool mf = a->precision == MOPK_PREC_BF16;
    const int64_t rows = d.BH * d.N;
    RET_IF(gather(a->io_dtype, a->dy, b.dy, d, st));
    RET_IF(drop_bwd_pair(b.P, b.dy, b.v, b.dP, b.dv, d, a->dropout_p, a->dropout_seed, mf, st));
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.P, b.dP, d, 1.f / sqrtf((float)d.dk));
    MOPK_CHECK_LAUNCH();
    RET_IF(gemm_map_vec(b.dP, false, b.k, b.dq, d, 1.f, 0.f, mf, st));
    RET_IF(gemm_map_vec(b.dP, true, b.q, b.dk, d, 1.f, 0.f, mf, st));
    RET_IF(scatter(a->io_dtype, b.dq, a->dq, d, st)); RET_IF(scatter(a->io_dtype, b.dk, a->dk_, d, st));
    return scatter(a->io_dtype, b.dv, a->dv, d, st);
}

// =====================================================================  dual path (MultiHopMSA)
struct DpBuf {
    float *q1, *k1, *v1, *q2, *k2, *v2, *S1, *S2, *A1, *A2, *T, *P, *tr, *ychain, *wsig;   // saved
    float *ybase, *dy, *dP, *dS1, *dS2, *dC, *D0, *D1, *dA1, *dA2, *g, *gt, *gt2, *dq, *dk, *dv;  // workspace
};
static DpBuf dp_carve(void *saved, void *ws, const Dm &d, int hops, size_t *ns, size_t *nw) {
    Carver cs(saved), cw(ws);
    const size_t nd = d.BH * d.N * d.dk, nn = d.BH * (size_t)d.N * d.LD;
    DpBuf b;
    b.q1 = cs.take<float>(nd); b.k1 = cs.take<float>(nd); b.v1 = cs.take<float>(nd);
    b.q2 = cs.take<float>(nd); b.k2 = cs.take<float>(nd); b.v2 = cs.take<float>(nd);
    b.S1 = cs.take<float>(nn); b.S2 = cs.take<float>(nn); b.A1 = cs.take<float>(nn); b.A2 = cs.take<float>(nn);
    b.T = cs.take<float>((size_t)(hops - 1) * nn);      // T_1..T_{hops-1}
    b.P = cs.take<float>(nn);
    b.tr = cs.take<float>((size_t)hops * nd);            // tr_0 = v2 .. tr_{hops-1}
    b.ychain = cs.take<float>(nd); b.wsig = cs.take<float>(64);
    b.ybase = cw.take<float>(nd); b.dy = cw.take<float>(nd); b.dP = cw.take<float>(nn);
    b.dS1 = cw.take<float>(nn); b.dS2 = cw.take<float>(nn); b.dC = cw.take<float>(nn);
    b.D0 = cw.take<float>(nn); b.D1 = cw.take<float>(nn); b.dA1 = cw.take<float>(nn); b.dA2 = cw.take<float>(nn);
    b.g = cw.take<float>(nd); b.gt = cw.take<float>(nd); b.gt2 = cw.take<float>(nd);
    b.dq = cw.take<float>(nd); b.dk = cw.take<float>(nd); b.dv = cw.take<float>(nd);
    if (ns) *ns = cs.off; if (nw) *nw = cw.off;
    return b;
}
size_t dp_saved_bytes(const MopkDualPathArgs *a) { size_t s; dp_carve(nullptr, nullptr, mkdm(a->B, a->H, a->N, a->dk), a->hops, &s, nullptr); return s; }
size_t dp_ws_bytes(const MopkDualPathArgs *a) { size_t w; dp_carve(nullptr, nullptr, mkdm(a->B, a->H, a->N, a->dk), a->hops, nullptr, &w); return w; }

__global__ void sigmoid1_kernel(const float *x, float *out) { out[0] = sigmoidf_(x[0]); }
__global__ void axpy2_kernel(float *y, const float *x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i];
}

// Smix (:209-218) over non-blocked entries, then softmax (:219-221); wave per row
__global__ void dp_mix_fwd_kernel(MopkDualPathArgs a, Dm d, DpBuf b, const float *C, MaskSpec m) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63, i = row % d.N;
    const int64_t bh = row / d.N;
    const int bb = bh / d.H, h = bh % d.H;
    float *P = b.P + row * d.LD;
    float mx = -INFINITY;
    for (int j = lane; j < d.N; j += 64) {
        if (is_blocked(m, bb, h, i, j)) { P[j] = -INFINITY; continue; }
        const float s1 = b.S1[row * d.LD + j], s2 = b.S2[row * d.LD + j];
        const float mm = fmaxf(s1, s2);
        const float lse = mm + logf(expf(s1 - mm) + expf(s2 - mm));
        float sm = s1 + a.g_and * s2 + a.g_or * (lse - s1) - a.g_not * (a.beta_not * s2);
        if (a.g_chain != 0.f) sm += a.g_chain * logf(C[row * d.LD + j] + EPS_CH);
        P[j] = sm; mx = fmaxf(mx, sm);
    }
    mx = wave_max(mx);
    float den = 0.f;
    for (int j = lane; j < d.N; j += 64) { const float e = expf(P[j] - mx); P[j] = e; den += e; }
    den = wave_sum(den);
    const float inv = 1.f / den;
    for (int j = lane; j < d.N; j += 64) P[j] *= inv;
}
template <typename T>
__global__ void combine2_kernel(const float *ybase, const float *ychain, const float *wsig, MopkView4 y, Dm d) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.BH * d.N * d.dk) return;
    const int dd = idx % d.dk; const int n = (idx / d.dk) % d.N; const int64_t bh = idx / ((int64_t)d.dk * d.N);
    st_from_f32((T *)y.ptr + (bh / d.H) * y.sb + (bh % d.H) * y.sh + n * y.sn + dd, ybase[idx] + wsig[0] * ychain[idx]);
}
__global__ void dot_part_kernel(const float *x, const float *y, const float *wsig, float *out, int64_t per_bh) {
    __shared__ float red[256];
    float sm = 0.f;
    for (int64_t i = threadIdx.x; i < per_bh; i += 256) sm += x[blockIdx.x * per_bh + i] * y[blockIdx.x * per_bh + i];
    red[threadIdx.x] = sm; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { const float w = wsig[0]; out[blockIdx.x] = red[0] * w * (1.f - w); }
}
// wave per row: dSmix = P(dP - dot) ; direct parts of dS1,dS2 ; dC
__global__ void dp_mix_bwd_kernel(MopkDualPathArgs a, Dm d, DpBuf b, const float *C, MaskSpec m) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63, i = row % d.N;
    const int64_t bh = row / d.N;
    const int bb = bh / d.H, h = bh % d.H;
    const float *P = b.P + row * d.LD;
    const float *dP = b.dP + row * d.LD;
    float dot = 0.f;
    for (int j = lane; j < d.N; j += 64) dot += P[j] * dP[j];
    dot = wave_sum(dot);
    for (int j = lane; j < d.N; j += 64) {
        const int64_t o = row * d.LD + j;
        if (is_blocked(m, bb, h, i, j)) { b.dS1[o] = 0.f; b.dS2[o] = 0.f; b.dC[o] = 0.f; continue; }
        const float dsm = P[j] * (dP[j] - dot);
        const float s1 = b.S1[o], s2 = b.S2[o];
        const float mm = fmaxf(s1, s2);
        const float lse = mm + logf(expf(s1 - mm) + expf(s2 - mm));
        b.dS1[o] = dsm * (1.f - a.g_or + a.g_or * expf(s1 - lse));
        b.dS2[o] = dsm * (a.g_and - a.g_not * a.beta_not + a.g_or * expf(s2 - lse));
        b.dC[o] = a.g_chain != 0.f ? a.g_chain * dsm / (C[o] + EPS_CH) : 0.f;
    }
}
// dS = (dSd + A (dA - sum A dA)) * alpha (0 on blocked: A is 0 there and dSd was zeroed) ; in place over dSd
__global__ void dp_final_kernel(const float *A, const float *dA, float *dS, Dm d, float alpha) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63;
    float dot = 0.f;
    for (int j = lane; j < d.N; j += 64) dot += A[row * d.LD + j] * dA[row * d.LD + j];
    dot = wave_sum(dot);
    for (int j = lane; j < d.N; j += 64) {
        const int64_t o = row * d.LD + j;
        dS[o] = (dS[o] + A[o] * (dA[o] - dot)) * alpha;
    }
}
static MaskSpec dp_mask(const MopkDualPathArgs *a) { return MaskSpec{a->causal, a->mask, a->mask_sb, a->mask_sh, a->mask_si, nullptr, 0, 0, 0}; }

int dp_fwd(const MopkDualPathArgs *a, hipStream_t st) {
    if (a->hops < 2) return MOPK_ERR_BAD_SHAPE;
    const Dm d = mkdm(a->B, a->H, a->N, a->dk);
    const DpBuf b = dp_carve(a->saved, a->workspace, d, a->hops, nullptr, nullptr);
    const bool mf = a->precision == MOPK_PREC_BF16;
    const int64_t rows = d.BH * d.N, nn = d.BH * (int64_t)d.N * d.LD, nd = d.BH * (int64_t)d.N * d.dk;
    const float sc = 1.f / sqrtf((float)d.dk);
    RET_IF(gather(a->io_dtype, a->q1, b.q1, d, st)); RET_IF(gather(a->io_dtype, a->k1, b.k1, d, st)); RET_IF(gather(a->io_dtype, a->v1, b.v1, d, st));
    RET_IF(gather(a->io_dtype, a->q2, b.q2, d, st)); RET_IF(gather(a->io_dtype, a->k2, b.k2, d, st)); RET_IF(gather(a->io_dtype, a->v2, b.tr, d, st));
    hipLaunchKernelGGL(sigmoid1_kernel, dim3(1), dim3(1), 0, st, a->chain_logit, b.wsig);
    RET_IF(gemm_nt_scores(b.q1, b.k1, b.S1, d, sc, mf, st));                         // :200
    RET_IF(gemm_nt_scores(b.q2, b.k2, b.S2, d, sc, mf, st));                         // :201
    const MaskSpec m = dp_mask(a);
    hipLaunchKernelGGL(masked_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.S1, b.A1, d, m);   // :206
    hipLaunchKernelGGL(masked_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.S2, b.A2, d, m);   // :207
    MOPK_CHECK_LAUNCH();
    const float *Tprev = b.A1;                                                        // :214-216
    for (int i = 1; i < a->hops; ++i) {
        RET_IF(gemm_map_map(Tprev, false, b.A2, false, b.T + (size_t)(i - 1) * nn, d, 0.f, mf, st));
        Tprev = b.T + (size_t)(i - 1) * nn;
    }
    hipLaunchKernelGGL(dp_mix_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, *a, d, b, Tprev, m);
    MOPK_CHECK_LAUNCH();
    for (int i = 1; i < a->hops; ++i)                                                 // :224-226
        RET_IF(gemm_map_vec(b.A2, false, b.tr + (size_t)(i - 1) * nd, b.tr + (size_t)i * nd, d, 1.f, 0.f, mf, st));
    RET_IF(gemm_map_vec(b.A1, false, b.tr + (size_t)(a->hops - 1) * nd, b.ychain, d, 1.f, 0.f, mf, st));   // :227
    if (a->dropout_p > 0.f) drop_map(b.P, b.dP, d, a->dropout_p, a->dropout_seed, st);     // :222 (the transport term uses the undropped A1, A2)
    RET_IF(gemm_map_vec(a->dropout_p > 0.f ? b.dP : b.P, false, b.v1, b.ybase, d, 1.f, 0.f, mf, st));
    if (a->io_dtype == MOPK_BF16) hipLaunchKernelGGL((combine2_kernel<unsigned short>), dim3((nd + 255) / 256), dim3(256), 0, st, b.ybase, b.ychain, b.wsig, a->y, d);
    else hipLaunchKernelGGL((combine2_kernel<float>), dim3((nd + 255) / 256), dim3(256), 0, st, b.ybase, b.ychain, b.wsig, a->y, d);
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}
int dp_bwd(const MopkDualPathArgs *a, hipStream_t st) {
    const Dm d = mkdm(a->B, a->H, a->N, a->dk);
    const DpBuf b = dp_carve(a->saved, a->workspace, d, a->hops, nullptr, nullptr);
    const bool mf = a->precision == MOPK_PREC_BF16;
    const int64_t rows = d.BH * d.N, nn = d.BH * (int64_t)d.N * d.LD, nd = d.BH * (int64_t)d.N * d.dk;
    const float sc = 1.f / sqrtf((float)d.dk);
    const int hops = a->hops;
    const MaskSpec m = dp_mask(a);
    RET_IF(gather(a->io_dtype, a->dy, b.dy, d, st));
    hipLaunchKernelGGL(dot_part_kernel, dim3((int)d.BH), dim3(256), 0, st, b.dy, b.ychain, b.wsig, a->dlogit_part, (int64_t)d.N * d.dk);
    RET_IF(drop_bwd_pair(b.P, b.dy, b.v1, b.dP, b.dv, d, a->dropout_p, a->dropout_seed, mf, st));   // dP = dy v1^T, dv1 = P^T dy
    RET_IF(scatter(a->io_dtype, b.dv, a->dv1, d, st));
    // transport: y_chain = A1 tr_{h-1}, tr_i = A2 tr_{i-1}
    RET_IF(gemm_nt_scores(b.dy, b.tr + (size_t)(hops - 1) * nd, b.dA1, d, 1.f, mf, st, b.wsig));    // dA1 = w dy tr^T
    RET_IF(gemm_map_vec(b.A1, true, b.dy, b.gt, d, 1.f, 0.f, mf, st, b.wsig));                      // gt = w A1^T dy
    float *gt = b.gt, *gt2 = b.gt2;
    for (int i = hops - 1; i >= 1; --i) {
        GemmDesc g = gd0(d.N, d.N, d.dk, (int)d.BH);                                 // dA2 (+)= gt tr_{i-1}^T
        g.A = gt; g.a_rs = d.dk; g.a_cs = 1; g.a_b1 = (int64_t)d.N * d.dk;
        g.B = b.tr + (size_t)(i - 1) * nd; g.b_rs = 1; g.b_cs = d.dk; g.b_b1 = (int64_t)d.N * d.dk;
        g.C = b.dA2; g.c_rs = d.LD; g.c_b1 = (int64_t)d.N * d.LD; g.beta = i == hops - 1 ? 0.f : 1.f;
        RET_IF(bgemm(g, mf, st));
        RET_IF(gemm_map_vec(b.A2, true, gt, gt2, d, 1.f, 0.f, mf, st));
        float *t = gt; gt = gt2; gt2 = t;
    }
    RET_IF(scatter(a->io_dtype, gt, a->dv2, d, st));                                 // dv2
    const float *Cf = hops >= 2 ? b.T + (size_t)(hops - 2) * nn : b.A1;
    hipLaunchKernelGGL(dp_mix_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, *a, d, b, Cf, m);
    MOPK_CHECK_LAUNCH();
    if (a->g_chain != 0.f) {                                                          // chain T_i = T_{i-1} A2
        const float *D = b.dC;
        float *pp[2] = {b.D0, b.D1};
        int cur = 0;
        for (int i = hops - 1; i >= 1; --i) {
            const float *Tp = i == 1 ? b.A1 : b.T + (size_t)(i - 2) * nn;
            RET_IF(gemm_map_map(Tp, true, D, false, b.dA2, d, 1.f, mf, st));
            RET_IF(gemm_map_map(D, false, b.A2, true, pp[cur], d, 0.f, mf, st));
            D = pp[cur]; cur ^= 1;
        }
        hipLaunchKernelGGL(axpy2_kernel, dim3((nn + 255) / 256), dim3(256), 0, st, b.dA1, D, nn);   // dA1 += D
        MOPK_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(dp_final_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.A1, b.dA1, b.dS1, d, sc);
    hipLaunchKernelGGL(dp_final_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.A2, b.dA2, b.dS2, d, sc);
    MOPK_CHECK_LAUNCH();
    RET_IF(gemm_map_vec(b.dS1, false, b.k1, b.dq, d, 1.f, 0.f, mf, st)); RET_IF(scatter(a->io_dtype, b.dq, a->dq1, d, st));
    RET_IF(gemm_map_vec(b.dS1, true, b.q1, b.dk, d, 1.f, 0.f, mf, st)); RET_IF(scatter(a->io_dtype, b.dk, a->dk1, d, st));
    RET_IF(gemm_map_vec(b.dS2, false, b.k2, b.dq, d, 1.f, 0.f, mf, st)); RET_IF(scatter(a->io_dtype, b.dq, a->dq2, d, st));
    RET_IF(gemm_map_vec(b.dS2, true, b.q2, b.dk, d, 1.f, 0.f, mf, st)); RET_IF(scatter(a->io_dtype, b.dk, a->dk2, d, st));
    return MOPK_OK;
}

// =====================================================================  Quartet
struct QtBuf {
    float *q, *k, *v, *q2, *k2, *z1, *z2, *sd1, *sd2, *P;                 // saved
    float *y, *dy, *dP, *dz2, *rowdm, *rowdqs, *dq, *dk, *dv;             // workspace
};
static QtBuf qt_carve(void *saved, void *ws, const Dm &d, size_t *ns, size_t *nw) {
    Carver cs(saved), cw(ws);
    const size_t nd = d.BH * d.N * d.dk, nn = d.BH * (size_t)d.N * d.LD, n1 = d.BH * d.N;
    QtBuf b;
    b.q = cs.take<float>(nd); b.k = cs.take<float>(nd); b.v = cs.take<float>(nd); b.q2 = cs.take<float>(nd); b.k2 = cs.take<float>(nd);
    b.z1 = cs.take<float>(nn); b.z2 = cs.take<float>(nn); b.sd1 = cs.take<float>(n1); b.sd2 = cs.take<float>(n1); b.P = cs.take<float>(nn);
    b.y = cw.take<float>(nd); b.dy = cw.take<float>(nd); b.dP = cw.take<float>(nn); b.dz2 = cw.take<float>(nn);
    b.rowdm = cw.take<float>(n1); b.rowdqs = cw.take<float>(n1);
    b.dq = cw.take<float>(nd); b.dk = cw.take<float>(nd); b.dv = cw.take<float>(nd);
    if (ns) *ns = cs.off; if (nw) *nw = cw.off;
    return b;
}
size_t qt_saved_bytes(const MopkQuartetArgs *a) { size_t s; qt_carve(nullptr, nullptr, mkdm(a->B, a->H, a->T, a->dh), &s, nullptr); return s; }
size_t qt_ws_bytes(const MopkQuartetArgs *a) { size_t w; qt_carve(nullptr, nullptr, mkdm(a->B, a->H, a->T, a->dh), nullptr, &w); return w; }

// z = (S - mean)/(std_unbiased + eps) over the FULL row (:95-98); in place; wave per row
__global__ void znorm_kernel(float *S, float *sd, Dm d, float eps) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63;
    float *p = S + row * d.LD;
    float sm = 0.f;
    for (int j = lane; j < d.N; j += 64) sm += p[j];
    const float mu = wave_sum(sm) / d.N;
    float ss = 0.f;
    for (int j = lane; j < d.N; j += 64) { const float t = p[j] - mu; ss += t * t; }
    const float sdv = sqrtf(wave_sum(ss) / (float)max(d.N - 1, 1));
    const float inv = 1.f / (sdv + eps);
    for (int j = lane; j < d.N; j += 64) p[j] = (p[j] - mu) * inv;
    if (lane == 0) sd[row] = sdv;
}
__global__ void qt_mix_fwd_kernel(MopkQuartetArgs a, Dm d, QtBuf b) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63, i = row % d.N;
    const int64_t bh = row / d.N;
    const float *am = a.add_mask ? a.add_mask + (bh / d.H) * a.am_sb + (bh % d.H) * a.am_sh + i * a.am_si : nullptr;
    float mq = 0.f, qs = 0.f;
    if (a.use_quartet) { mq = sigmoidf_(*a.mixture); qs = *a.quartet_scale; }
    float *P = b.P + row * d.LD;
    float mx = -INFINITY;
    for (int j = lane; j < d.N; j += 64) {
        float sc;
        if (j > i) sc = -INFINITY;                                           // :112-113
        else {
            const float z1 = b.z1[row * d.LD + j];
            sc = a.use_quartet ? (1.f - mq) * z1 + mq * (z1 * b.z2[row * d.LD + j]) * qs : z1;   // :104-110
            if (am) sc += am[j];                                             // :115-116
        }
        P[j] = sc; mx = fmaxf(mx, sc);
    }
    mx = wave_max(mx);
    float den = 0.f;
    for (int j = lane; j < d.N; j += 64) { const float e = expf(P[j] - mx); P[j] = e; den += e; }
    den = wave_sum(den);
    const float inv = 1.f / den;
    for (int j = lane; j < d.N; j += 64) {
        P[j] *= inv;
        if (a.attn) a.attn[row * (int64_t)d.N + j] = P[j];
    }
}
// dsc = P(dP - dot); dz1 -> dP (in place), dz2 -> b.dz2 ; per-row partials of dm, dqs
__global__ void qt_mix_bwd_kernel(MopkQuartetArgs a, Dm d, QtBuf b) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63;
    const float *P = b.P + row * d.LD;
    float *dP = b.dP + row * d.LD;
    float dot = 0.f;
    for (int j = lane; j < d.N; j += 64) dot += P[j] * dP[j];
    dot = wave_sum(dot);
    float mq = 0.f, qs = 0.f;
    if (a.use_quartet) { mq = sigmoidf_(*a.mixture); qs = *a.quartet_scale; }
    float sdm = 0.f, sdq = 0.f;
    for (int j = lane; j < d.N; j += 64) {
        const float dsc = P[j] * (dP[j] - dot);
        if (a.use_quartet) {
            const float z1 = b.z1[row * d.LD + j], z2 = b.z2[row * d.LD + j];
            dP[j] = dsc * ((1.f - mq) + mq * qs * z2);
            b.dz2[row * d.LD + j] = dsc * (mq * qs * z1);
            sdm += dsc * (-z1 + z1 * z2 * qs);
            sdq += dsc * (mq * z1 * z2);
        } else dP[j] = dsc;
    }
    sdm = wave_sum(sdm); sdq = wave_sum(sdq);
    if (lane == 0) { b.rowdm[row] = sdm * mq * (1.f - mq); b.rowdqs[row] = sdq; }
}
// gradient through z-norm, times alpha; in place; wave per row
__global__ void znorm_bwd_kernel(float *dz, const float *z, const float *sd, Dm d, float eps, float alpha) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63;
    const float sdv = sd[row], den = sdv + eps;
    float sm = 0.f, sp = 0.f;
    for (int j = lane; j < d.N; j += 64) { const float g = dz[row * d.LD + j]; sm += g; sp += g * z[row * d.LD + j] * den; }
    sm = wave_sum(sm) / d.N; sp = wave_sum(sp);
    const float c2 = sp / ((float)max(d.N - 1, 1) * fmaxf(sdv, 1e-30f) * den * den);
    for (int j = lane; j < d.N; j += 64) {
        const int64_t o = row * d.LD + j;
        dz[o] = ((dz[o] - sm) / den - z[o] * den * c2) * alpha;
    }
}
__global__ void rowsum_part_kernel(const float *x, float *out, int N) {
    __shared__ float red[256];
    float sm = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) sm += x[(int64_t)blockIdx.x * N + i];
    red[threadIdx.x] = sm; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
int qt_fwd(const MopkQuartetArgs *a, hipStream_t st) {
    const Dm d = mkdm(a->B, a->H, a->T, a->dh);
    const QtBuf b = qt_carve(a->saved, a->workspace, d, nullptr, nullptr);
    const bool mf = a->precision == MOPK_PREC_BF16;
    const int64_t rows = d.BH * d.N;
    const float sc = 1.f / sqrtf((float)d.dk);
    RET_IF(gather(a->io_dtype, a->q, b.q, d, st)); RET_IF(gather(a->io_dtype, a->k, b.k, d, st)); RET_IF(gather(a->io_dtype, a->v, b.v, d, st));
    RET_IF(gemm_nt_scores(b.q, b.k, b.z1, d, sc, mf, st));                             // :88
    hipLaunchKernelGGL(znorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.z1, b.sd1, d, a->use_quartet ? a->eps : 1e-5f);
    if (a->use_quartet) {
        RET_IF(gather(a->io_dtype, a->q2, b.q2, d, st)); RET_IF(gather(a->io_dtype, a->k2, b.k2, d, st));
        RET_IF(gemm_nt_scores(b.q2, b.k2, b.z2, d, sc, mf, st));                       // :93
        hipLaunchKernelGGL(znorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.z2, b.sd2, d, a->eps);
    }
    hipLaunchKernelGGL(qt_mix_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, *a, d, b);
    if (a->dropout_p > 0.f) {                                                          // :119; the returned weights are the dropped ones (:125-126)
        drop_map(b.P, b.dP, d, a->dropout_p, a->dropout_seed, st);
        if (a->attn) hipLaunchKernelGGL(drop_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.P, d.LD, a->attn, d.N, fa_drop(a->dropout_p, a->dropout_seed), rows, d.N);
    }
    MOPK_CHECK_LAUNCH();
    RET_IF(gemm_map_vec(a->dropout_p > 0.f ? b.dP : b.P, false, b.v, b.y, d, 1.f, 0.f, mf, st));   // :121
    return scatter(a->io_dtype, b.y, a->y, d, st);
}
int qt_bwd(const MopkQuartetArgs *a, hipStream_t st) {
    const Dm d = mkdm(a->B, a->H, a->T, a->dh);
    const QtBuf b = qt_carve(a->saved, a->workspace, d, nullptr, nullptr);
    const bool mf = a->precision == MOPK_PREC_BF16;
    const int64_t rows = d.BH * d.N;
    const float sc = 1.f / sqrtf((float)d.dk);
    RET_IF(gather(a->io_dtype, a->dy, b.dy, d, st));
    RET_IF(drop_bwd_pair(b.P, b.dy, b.v, b.dP, b.dv, d, a->dropout_p, a->dropout_seed, mf, st));
    RET_IF(scatter(a->io_dtype, b.dv, a->dv, d, st));
    hipLaunchKernelGGL(qt_mix_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, *a, d, b);
    MOPK_CHECK_LAUNCH();
    const float eps1 = a->use_quartet ? a->eps : 1e-5f;
    hipLaunchKernelGGL(znorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.dP, b.z1, b.sd1, d, eps1, sc);
    MOPK_CHECK_LAUNCH();
    RET_IF(gemm_map_vec(b.dP, false, b.k, b.dq, d, 1.f, 0.f, mf, st)); RET_IF(scatter(a->io_dtype, b.dq, a->dq, d, st));
    RET_IF(gemm_map_vec(b.dP, true, b.q, b.dk, d, 1.f, 0.f, mf, st)); RET_IF(scatter(a->io_dtype, b.dk, a->dk_, d, st));
    if (a->use_quartet) {
        hipLaunchKernelGGL(znorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.dz2, b.z2, b.sd2, d, a->eps, sc);
        MOPK_CHECK_LAUNCH();
        RET_IF(gemm_map_vec(b.dz2, false, b.k2, b.dq, d, 1.f, 0.f, mf, st)); RET_IF(scatter(a->io_dtype, b.dq, a->dq2, d, st));
        RET_IF(gemm_map_vec(b.dz2, true, b.q2, b.dk, d, 1.f, 0.f, mf, st)); RET_IF(scatter(a->io_dtype, b.dk, a->dk2, d, st));
        hipLaunchKernelGGL(rowsum_part_kernel, dim3((int)d.BH), dim3(256), 0, st, b.rowdm, a->dmixture_part, d.N);
        hipLaunchKernelGGL(rowsum_part_kernel, dim3((int)d.BH), dim3(256), 0, st, b.rowdqs, a->dqscale_part, d.N);
        MOPK_CHECK_LAUNCH();
    }
    return MOPK_OK;
}

// =====================================================================  CrossViewMixerMSA
struct CvBuf {
    float *q1, *k1, *v1, *q2, *k2, *S1, *S2, *S12, *S21, *P, *A1, *A2, *A, *sden; int *kst;   // saved
    float *y, *dy, *dA, *dS, *g1, *g2, *g12, *g21, *dA1, *danc, *rowt, *dq1, *dk1, *dv1, *dq2, *dk2;   // workspace
};
static CvBuf cv_carve(void *saved, void *ws, const Dm &d, bool prior, size_t *ns, size_t *nw) {
    Carver cs(saved), cw(ws);
    const size_t nd = d.BH * d.N * d.dk, nn = d.BH * (size_t)d.N * d.LD, n1 = d.BH * d.N;
    CvBuf b;
    b.q1 = cs.take<float>(nd); b.k1 = cs.take<float>(nd); b.v1 = cs.take<float>(nd); b.q2 = cs.take<float>(nd); b.k2 = cs.take<float>(nd);
    b.S1 = cs.take<float>(nn); b.S2 = cs.take<float>(nn); b.S12 = cs.take<float>(nn); b.S21 = cs.take<float>(nn); b.P = cs.take<float>(nn);
    b.A1 = cs.take<float>(prior ? nn : 0); b.A2 = cs.take<float>(prior ? nn : 0); b.A = cs.take<float>(prior ? nn : 0);
    b.sden = cs.take<float>(prior ? n1 : 0); b.kst = cs.take<int>(d.BH);
    b.y = cw.take<float>(nd); b.dy = cw.take<float>(nd); b.dA = cw.take<float>(nn); b.dS = cw.take<float>(nn);
    b.g1 = cw.take<float>(nn); b.g2 = cw.take<float>(nn); b.g12 = cw.take<float>(nn); b.g21 = cw.take<float>(nn);
    b.dA1 = cw.take<float>(prior ? nn : 0); b.danc = cw.take<float>(prior ? n1 : 0); b.rowt = cw.take<float>(prior ? n1 : 0);
    b.dq1 = cw.take<float>(nd); b.dk1 = cw.take<float>(nd); b.dv1 = cw.take<float>(nd); b.dq2 = cw.take<float>(nd); b.dk2 = cw.take<float>(nd);
    if (ns) *ns = cs.off; if (nw) *nw = cw.off;
    return b;
}
size_t cv_saved_bytes(const MopkCrossViewArgs *a) { size_t s; cv_carve(nullptr, nullptr, mkdm(a->B, a->H, a->N, a->dk), a->use_prior != 0, &s, nullptr); return s; }
size_t cv_ws_bytes(const MopkCrossViewArgs *a) { size_t w; cv_carve(nullptr, nullptr, mkdm(a->B, a->H, a->N, a->dk), a->use_prior != 0, nullptr, &w); return w; }

// S = m11 S1 + m12 S12 + m21 S21 + m22 S2 + t1 S1^T + t2 S2^T     :105-110
__global__ void cv_combine_kernel(CvBuf b, Dm d, const float *mix, float t1, float t2, float *out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.N * d.N) return;
    const int i = p / d.N, j = p % d.N;
    const int64_t base = (int64_t)blockIdx.y * d.N * d.LD, o = base + (int64_t)i * d.LD + j, ot = base + (int64_t)j * d.LD + i;
    out[o] = mix[0] * b.S1[o] + mix[1] * b.S12[o] + mix[2] * b.S21[o] + mix[3] * b.S2[o] + t1 * b.S1[ot] + t2 * b.S2[ot];
}
// anchor row per (b,h)   :131-145
__global__ void cv_anchor_kernel(CvBuf b, Dm d, int mode, int fixed, int *out_user) {
    __shared__ float bv[256]; __shared__ int bi[256];
    const int64_t bh = blockIdx.x;
    int best = 0;
    if (mode == 0) best = max(0, min(d.N - 1, fixed));
    else if (mode == 1) {
        float bval = -INFINITY; int bidx = 0x7fffffff;
        for (int i = threadIdx.x; i < d.N; i += 256) {
            const float *row = b.A2 + (bh * d.N + i) * (int64_t)d.LD;
            float sm = 0.f;
            for (int j = 0; j < d.N; ++j) sm += row[j];
            if (sm > bval) { bval = sm; bidx = i; }                // strided scan keeps the smallest index among equal values
        }
        bv[threadIdx.x] = bval; bi[threadIdx.x] = bidx; __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) {
                const float ov = bv[threadIdx.x + st]; const int oi = bi[threadIdx.x + st];
                if (ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = ov; bi[threadIdx.x] = oi; }
            }
            __syncthreads();
        }
        best = bi[0];
    }
    if (threadIdx.x == 0) { b.kst[bh] = best; if (out_user) out_user[bh] = best; }
}
// A = (1-w) P + w * normalise_j(A1[i,j] * A2[k*,j])     :147-150 ; wave per row
__global__ void cv_sharp_fwd_kernel(CvBuf b, Dm d, float pw) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63;
    const int64_t bh = row / d.N;
    const float *a1 = b.A1 + row * d.LD, *anc = b.A2 + (bh * d.N + b.kst[bh]) * (int64_t)d.LD, *p = b.P + row * d.LD;
    float sm = 0.f;
    for (int j = lane; j < d.N; j += 64) sm += a1[j] * anc[j];
    sm = wave_sum(sm) + 1e-9f;
    if (lane == 0) b.sden[row] = sm;
    const float inv = 1.f / sm;
    float *o = b.A + row * d.LD;
    for (int j = lane; j < d.N; j += 64) o[j] = (1.f - pw) * p[j] + pw * a1[j] * anc[j] * inv;
}
// prior backward, wave per row: dA1 = du * anc ; g12 <- du * A1 (column-summed into d anc) ; dA <- (1-w) dA
__global__ void cv_sharp_bwd_kernel(CvBuf b, Dm d, float pw) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.BH * d.N) return;
    const int lane = threadIdx.x & 63;
    const int64_t bh = row / d.N;
    const float *a1 = b.A1 + row * d.LD, *anc = b.A2 + (bh * d.N + b.kst[bh]) * (int64_t)d.LD;
    float *dA = b.dA + row * d.LD;
    const float inv = 1.f / b.sden[row];
    float dot = 0.f;
    for (int j = lane; j < d.N; j += 64) dot += pw * dA[j] * a1[j] * anc[j] * inv;
    dot = wave_sum(dot);
    for (int j = lane; j < d.N; j += 64) {
        const float du = (pw * dA[j] - dot) * inv;
        b.dA1[row * d.LD + j] = du * anc[j];
        b.g12[row * d.LD + j] = du * a1[j];
        dA[j] *= (1.f - pw);
    }
}
// thread per column: danc[j] = sum_i g12[i,j] ; then (one block per bh) the softmax backward of row k* of A2
__global__ void cv_anchor_bwd_kernel(CvBuf b, Dm d) {
    __shared__ float red[256];
    const int64_t bh = blockIdx.x;
    const float *anc = b.A2 + (bh * d.N + b.kst[bh]) * (int64_t)d.LD;
    float part = 0.f;
    for (int j = threadIdx.x; j < d.N; j += 256) {
        float sm = 0.f;
        for (int i = 0; i < d.N; ++i) sm += b.g12[(bh * d.N + i) * (int64_t)d.LD + j];
        b.danc[bh * d.N + j] = sm;
        part += anc[j] * sm;
    }
    red[threadIdx.x] = part; __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    const float dot = red[0];
    for (int j = threadIdx.x; j < d.N; j += 256) b.rowt[bh * d.N + j] = anc[j] * (b.danc[bh * d.N + j] - dot);
}
// dmix_part[bh] = (sum dS S1, sum dS S12, sum dS S21, sum dS S2)
__global__ void cv_dmix_kernel(CvBuf b, Dm d, float *out) {
    __shared__ float red[4][256];
    const int64_t bh = blockIdx.x;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = threadIdx.x; p < d.N * d.N; p += 256) {
        const int64_t o = bh * d.N * (int64_t)d.LD + (int64_t)(p / d.N) * d.LD + p % d.N;
        const float g = b.dS[o];
        acc[0] = fmaf(g, b.S1[o], acc[0]); acc[1] = fmaf(g, b.S12[o], acc[1]); acc[2] = fmaf(g, b.S21[o], acc[2]); acc[3] = fmaf(g, b.S2[o], acc[3]);
    }
    for (int q = 0; q < 4; ++q) red[q][threadIdx.x] = acc[q];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 4) out[bh * 4 + threadIdx.x] = red[threadIdx.x][0];
}
// gradients of the four score maps (already multiplied by 1/sqrt(dk)):
// g1 = m11 dS + t1 dS^T [+ dS1 of the prior], g2 = m22 dS + t2 dS^T [+ row k*], g12 = m12 dS, g21 = m21 dS
__global__ void cv_ds_kernel(CvBuf b, Dm d, const float *mix, float t1, float t2, float scale, int prior) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.N * d.N) return;
    const int i = p / d.N, j = p % d.N;
    const int64_t bh = blockIdx.y, base = bh * d.N * (int64_t)d.LD, o = base + (int64_t)i * d.LD + j, ot = base + (int64_t)j * d.LD + i;
    const float g = b.dS[o], gt = b.dS[ot];
    float x1 = mix[0] * g + t1 * gt, x2 = mix[3] * g + t2 * gt;
    if (prior) {
        x1 += b.dA1[o];                                           // dA1 holds the softmax backward of A1 at this point
        if (i == b.kst[bh]) x2 += b.rowt[bh * d.N + j];
    }
    b.g1[o] = x1 * scale; b.g2[o] = x2 * scale; b.g12[o] = mix[1] * g * scale; b.g21[o] = mix[2] * g * scale;
}

int cv_fwd(const MopkCrossViewArgs *a, hipStream_t st) {
    const Dm d = mkdm(a->B, a->H, a->N, a->dk);
    const bool prior = a->use_prior != 0;
    const CvBuf b = cv_carve(a->saved, a->workspace, d, prior, nullptr, nullptr);
    const bool mf = a->precision == MOPK_PREC_BF16;
    const float sc = 1.f / sqrtf((float)d.dk);
    RET_IF(gather(a->io_dtype, a->q1, b.q1, d, st)); RET_IF(gather(a->io_dtype, a->k1, b.k1, d, st)); RET_IF(gather(a->io_dtype, a->v1, b.v1, d, st));
    RET_IF(gather(a->io_dtype, a->q2, b.q2, d, st)); RET_IF(gather(a->io_dtype, a->k2, b.k2, d, st));
    RET_IF(gemm_nt_scores(b.q1, b.k1, b.S1, d, sc, mf, st)); RET_IF(gemm_nt_scores(b.q2, b.k2, b.S2, d, sc, mf, st));     // :99-100
    RET_IF(gemm_nt_scores(b.q1, b.k2, b.S12, d, sc, mf, st)); RET_IF(gemm_nt_scores(b.q2, b.k1, b.S21, d, sc, mf, st));   // :101-102
    const dim3 pix((d.N * d.N + 255) / 256, (unsigned)d.BH);
    const int64_t rows = d.BH * d.N;
    const MaskSpec m{a->causal, a->mask, a->mask_sb, a->mask_sh, a->mask_si, nullptr, 0, 0, 0};
    hipLaunchKernelGGL(cv_combine_kernel, pix, dim3(256), 0, st, b, d, a->mix, a->t1, a->t2, b.P);
    hipLaunchKernelGGL(masked_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.P, b.P, d, m);                    // :124-125
    if (prior) {
        hipLaunchKernelGGL(masked_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.S1, b.A1, d, m);              // :129-130
        hipLaunchKernelGGL(masked_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.S2, b.A2, d, m);
        hipLaunchKernelGGL(cv_anchor_kernel, dim3((unsigned)d.BH), dim3(256), 0, st, b, d, a->anchor_mode, a->fixed_k_star, a->k_star);
        hipLaunchKernelGGL(cv_sharp_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b, d, a->prior_weight);
    }
    const float *Aw = prior ? b.A : b.P;
    if (a->dropout_p > 0.f) { drop_map(Aw, b.dA, d, a->dropout_p, a->dropout_seed, st); Aw = b.dA; }                     // :151 (workspace plane)
    MOPK_CHECK_LAUNCH();
    RET_IF(gemm_map_vec(Aw, false, b.v1, b.y, d, 1.f, 0.f, mf, st));                                                     // :152
    return scatter(a->io_dtype, b.y, a->y, d, st);
}
int cv_bwd(const MopkCrossViewArgs *a, hipStream_t st) {
    const Dm d = mkdm(a->B, a->H, a->N, a->dk);
    const bool prior = a->use_prior != 0;
    const CvBuf b = cv_carve(a->saved, a->workspace, d, prior, nullptr, nullptr);
    const bool mf = a->precision == MOPK_PREC_BF16;
    const float sc = 1.f / sqrtf((float)d.dk);
    const int64_t rows = d.BH * d.N;
    const dim3 pix((d.N * d.N + 255) / 256, (unsigned)d.BH);
    RET_IF(gather(a->io_dtype, a->dy, b.dy, d, st));
    RET_IF(drop_bwd_pair(prior ? b.A : b.P, b.dy, b.v1, b.dA, b.dv1, d, a->dropout_p, a->dropout_seed, mf, st));   // dA = dy v1^T, dv1 = A^T dy
    if (prior) {
        hipLaunchKernelGGL(cv_sharp_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b, d, a->prior_weight);
        hipLaunchKernelGGL(cv_anchor_bwd_kernel, dim3((unsigned)d.BH), dim3(256), 0, st, b, d);
        hipLaunchKernelGGL(softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.A1, b.dA1, d, 1.f);
    }
    hipMemcpyAsync(b.dS, b.dA, sizeof(float) * d.BH * d.N * d.LD, hipMemcpyDeviceToDevice, st);
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, b.P, b.dS, d, 1.f);
    hipLaunchKernelGGL(cv_dmix_kernel, dim3((unsigned)d.BH), dim3(256), 0, st, b, d, a->dmix_part);
    hipLaunchKernelGGL(cv_ds_kernel, pix, dim3(256), 0, st, b, d, a->mix, a->t1, a->t2, sc, prior ? 1 : 0);
    MOPK_CHECK_LAUNCH();
    RET_IF(gemm_map_vec(b.g1, false, b.k1, b.dq1, d, 1.f, 0.f, mf, st)); RET_IF(gemm_map_vec(b.g12, false, b.k2, b.dq1, d, 1.f, 1.f, mf, st));
    RET_IF(gemm_map_vec(b.g1, true, b.q1, b.dk1, d, 1.f, 0.f, mf, st)); RET_IF(gemm_map_vec(b.g21, true, b.q2, b.dk1, d, 1.f, 1.f, mf, st));
    RET_IF(gemm_map_vec(b.g2, false, b.k2, b.dq2, d, 1.f, 0.f, mf, st)); RET_IF(gemm_map_vec(b.g21, false, b.k1, b.dq2, d, 1.f, 1.f, mf, st));
    RET_IF(gemm_map_vec(b.g2, true, b.q2, b.dk2, d, 1.f, 0.f, mf, st)); RET_IF(gemm_map_vec(b.g12, true, b.q1, b.dk2, d, 1.f, 1.f, mf, st));
    RET_IF(scatter(a->io_dtype, b.dq1, a->dq1, d, st)); RET_IF(scatter(a->io_dtype, b.dk1, a->dk1, d, st)); RET_IF(scatter(a->io_dtype, b.dv1, a->dv1, d, st));
    RET_IF(scatter(a->io_dtype, b.dq2, a->dq2, d, st));
    return scatter(a->io_dtype, b.dk2, a->dk2, d, st);
}

}  // namespace mopk
