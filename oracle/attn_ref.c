/*
 * CPU oracle, plain C.  TEST INFRASTRUCTURE ONLY: nothing in the product path
 * (flash_attention_impls_amd/) links, loads or calls this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Restates the reference's forward algorithm for one (batch, head) at a time:
 *   - split-Q tile loop with online softmax:  code/triton_fa2/FA2-triton.py:56-93
 *   - causal rule col > row -> -inf:          code/triton_fa2/FA2-triton.py:70-73
 *   - scale = 1/sqrt(D) supplied by caller:   code/triton_fa2/FA2-triton.py:183
 *   - final 1/l with l==0 guard:              code/cutlass_cuda_fa1/run/flash_attn_cutlass.cu:446-452
 *   - naive 3-pass baseline (fp64):           code/cutlass_cuda_fa1/run/test_flash_attn.cu:548-615
 *   - backward with recomputed softmax:       code/triton_fa2/FA2-triton.py:98-170 (dS in the correct
 *     p*(dP - rowsum(dP*p))*scale form; the reference's :160-161 is defective, see oracle/attn_oracle.py)
 *
 * Parity pin: checked in tests/test_oracle.py against tests/golden/*.npz, which
 * were produced from the reference's own sdpa_reference (oracle/gen_golden.py).
 *
 * Build: gcc -O2 -fPIC -shared -o oracle/_build/liboracle_attn.so oracle/attn_ref.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Tiled online-softmax forward, fp32 arithmetic, deferred normalisation.
 * q,k,v,o: contiguous [B][H][N][D] fp32.  lse (nullable): [B][H][N] fp32,
 * natural-log-sum-exp of scale*q.k over unmasked keys.  Returns 0 on success. */
int oracle_attn_fwd_f32(const float* q, const float* k, const float* v, float* o, float* lse,
                        int B, int H, int N, int D, int causal, float scale,
                        int block_m, int block_n)
{
    if (B < 0 || H < 0 || N < 0 || D <= 0 || block_m <= 0 || block_n <= 0) return -1;
    float* s   = (float*)malloc(sizeof(float) * (size_t)block_m * block_n);
    float* acc = (float*)malloc(sizeof(float) * (size_t)block_m * D);
    float* m_i = (float*)malloc(sizeof(float) * (size_t)block_m);
    float* l_i = (float*)malloc(sizeof(float) * (size_t)block_m);
    if (!s || !acc || !m_i || !l_i) { free(s); free(acc); free(m_i); free(l_i); return -2; }

    for (long bh = 0; bh < (long)B * H; ++bh) {
        const float* qh = q + bh * (long)N * D;
        const float* kh = k + bh * (long)N * D;
        const float* vh = v + bh * (long)N * D;
        float* oh = o + bh * (long)N * D;
        for (int r0 = 0; r0 < N; r0 += block_m) {                 /* FA2-triton.py:41-44 */
            int nr = (N - r0 < block_m) ? N - r0 : block_m;
            for (int i = 0; i < nr; ++i) { m_i[i] = -INFINITY; l_i[i] = 0.f; }     /* :56-57 */
            memset(acc, 0, sizeof(float) * (size_t)nr * D);                        /* :58 */
            for (int c0 = 0; c0 < N; c0 += block_n) {             /* :60 */
                if (causal && c0 > r0 + nr - 1) break;            /* fully masked tile: skip */
                int nc = (N - c0 < block_n) ? N - c0 : block_n;
                for (int i = 0; i < nr; ++i) {                    /* S = Q K^T * scale :68 */
                    const float* qi = qh + (long)(r0 + i) * D;
                    for (int j = 0; j < nc; ++j) {
                        const float* kj = kh + (long)(c0 + j) * D;
                        float dot = 0.f;
                        for (int d = 0; d < D; ++d) dot += qi[d] * kj[d];
                        float sv = dot * scale;
                        if (causal && (c0 + j) > (r0 + i)) sv = -INFINITY;         /* :70-73 */
                        s[i * block_n + j] = sv;
                    }
                }
                for (int i = 0; i < nr; ++i) {
                    float mx = m_i[i];
                    for (int j = 0; j < nc; ++j) mx = fmaxf(mx, s[i * block_n + j]);   /* :75 */
                    float corr = (m_i[i] == -INFINITY) ? 0.f : expf(m_i[i] - mx);
                    float rs = 0.f;
                    float* ai = acc + (long)i * D;
                    for (int d = 0; d < D; ++d) ai[d] *= corr;
                    for (int j = 0; j < nc; ++j) {
                        float p = (mx == -INFINITY) ? 0.f : expf(s[i * block_n + j] - mx); /* :76 */
                        rs += p;
                        const float* vj = vh + (long)(c0 + j) * D;
                        for (int d = 0; d < D; ++d) ai[d] += p * vj[d];             /* :79 */
                    }
                    l_i[i] = l_i[i] * corr + rs;                                     /* :77 */
                    m_i[i] = mx;                                                     /* :84 */
                }
            }
            for (int i = 0; i < nr; ++i) {                        /* flash_attn_cutlass.cu:446-452 */
                float inv = (l_i[i] > 0.f) ? 1.f / l_i[i] : 0.f;
                float* oi = oh + (long)(r0 + i) * D;
                for (int d = 0; d < D; ++d) oi[d] = acc[(long)i * D + d] * inv;
                if (lse) lse[bh * (long)N + r0 + i] = (l_i[i] > 0.f) ? m_i[i] + logf(l_i[i]) : -INFINITY;
            }
        }
    }
    free(s); free(acc); free(m_i); free(l_i);
    return 0;
}

/* Naive 3-pass attention in double precision (test_flash_attn.cu:548-615). */
int oracle_attn_naive_f64(const float* q, const float* k, const float* v, double* o, double* lse,
                          int B, int H, int N, int D, int causal, double scale)
{
    double* sc = (double*)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
    if (!sc) return -2;
    for (long bh = 0; bh < (long)B * H; ++bh) {
        const float* qh = q + bh * (long)N * D;
        const float* kh = k + bh * (long)N * D;
        const float* vh = v + bh * (long)N * D;
        for (int i = 0; i < N; ++i) {
            int lim = causal ? i + 1 : N;
            double mx = -INFINITY;
            for (int j = 0; j < lim; ++j) {                       /* pass 1 :575-586 */
                double dot = 0.0;
                for (int d = 0; d < D; ++d) dot += (double)qh[(long)i * D + d] * (double)kh[(long)j * D + d];
                sc[j] = dot * scale;
                if (sc[j] > mx) mx = sc[j];                       /* :588-592 */
            }
            double sum = 0.0;
            for (int j = 0; j < lim; ++j) { sc[j] = exp(sc[j] - mx); sum += sc[j]; }   /* :594-600 */
            for (int d = 0; d < D; ++d) {                         /* pass 3 :602-613 */
                double a = 0.0;
                for (int j = 0; j < lim; ++j) a += sc[j] * (double)vh[(long)j * D + d];
                o[(bh * (long)N + i) * D + d] = a / sum;
            }
            if (lse) lse[bh * (long)N + i] = mx + log(sum);
        }
    }
    free(sc);
    return 0;
}

/* Backward of o = softmax(scale q k^T [+mask]) v for upstream gradient d_o, float64 accumulation, one query
 * row at a time (p recomputed from the row's own max / sum: FA2-triton.py:147-156; dv :158, dP :159,
 * delta :160 over the whole row, dS :161 corrected, dQ :163, dK :164).  q,k,v,d_o: contiguous [B][H][N][D]
 * fp32; dq,dk,dv: same shape, float64, overwritten. */
int oracle_attn_bwd_f64(const float* q, const float* k, const float* v, const float* d_o,
                        double* dq, double* dk, double* dv,
                        int B, int H, int N, int D, int causal, double scale)
{
    double* p = (double*)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
    double* dp = (double*)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
    if (!p || !dp) { free(p); free(dp); return -2; }
    memset(dq, 0, sizeof(double) * (size_t)B * H * N * D);
    memset(dk, 0, sizeof(double) * (size_t)B * H * N * D);
    memset(dv, 0, sizeof(double) * (size_t)B * H * N * D);
    for (long bh = 0; bh < (long)B * H; ++bh) {
        const float* qh = q + bh * (long)N * D;
        const float* kh = k + bh * (long)N * D;
        const float* vh = v + bh * (long)N * D;
        const float* gh = d_o + bh * (long)N * D;
        double* dqh = dq + bh * (long)N * D;
        double* dkh = dk + bh * (long)N * D;
        double* dvh = dv + bh * (long)N * D;
        for (int i = 0; i < N; ++i) {
            int lim = causal ? i + 1 : N;
            double mx = -INFINITY, sum = 0.0, delta = 0.0;
            for (int j = 0; j < lim; ++j) {
                double dot = 0.0, g = 0.0;
                for (int d = 0; d < D; ++d) {
                    dot += (double)qh[(long)i * D + d] * (double)kh[(long)j * D + d];
                    g += (double)gh[(long)i * D + d] * (double)vh[(long)j * D + d];       /* dP :159 */
                }
                p[j] = dot * scale;
                dp[j] = g;
                if (p[j] > mx) mx = p[j];
            }
            for (int j = 0; j < lim; ++j) { p[j] = exp(p[j] - mx); sum += p[j]; }
            for (int j = 0; j < lim; ++j) { p[j] /= sum; delta += p[j] * dp[j]; }          /* :156,160 */
            for (int j = 0; j < lim; ++j) {
                double ds = p[j] * (dp[j] - delta) * scale;                                /* :161 */
                for (int d = 0; d < D; ++d) {
                    dvh[(long)j * D + d] += p[j] * (double)gh[(long)i * D + d];            /* :158 */
                    dqh[(long)i * D + d] += ds * (double)kh[(long)j * D + d];              /* :163 */
                    dkh[(long)j * D + d] += ds * (double)qh[(long)i * D + d];              /* :164 */
                }
            }
        }
    }
    free(p); free(dp);
    return 0;
}

/* compute_max_relative_error (test_flash_attn.cu:108-143): max |a-b|/(|a|+|b|+1e-5) */
double oracle_sym_rel_err(const float* a, const float* b, long n)
{
    double worst = 0.0;
    for (long i = 0; i < n; ++i) {
        double e = fabs((double)a[i] - (double)b[i]) / (fabs((double)a[i]) + fabs((double)b[i]) + 1e-5);
        if (e > worst) worst = e;
    }
    return worst;
}
