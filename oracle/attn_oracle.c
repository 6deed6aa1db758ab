/*
 * attn_oracle.c — CPU restatement of the reference attention algorithm.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this; the
 * product path (flash-attention-turing_amd/) never does and fails loudly without its HIP build.
 *
 * What is restated (reference = ssiu/flash-attention-turing, paths relative to its root):
 *   forward   csrc/flash_attn/src/flash_fwd_kernel.h:23-789 (compute_attn_1rowblock):
 *             s = q.k / sqrt(d) (:351,562), bottom-right causal + key-bound mask (mask.h:20-72,172),
 *             softmax with fp32 statistics, P rounded to fp16 BEFORE P.V (:485,670),
 *             O = sum(P V) / l (:717-730) rounded to fp16 (:740), LSE = m + ln(l) (:766-785),
 *             rows with no visible key: O = 0, LSE = 0.0 (:720-728,767-771).
 *   preproc.  csrc/flash_attn/src/flash_bwd_preprocess_kernel.h:23-96: D = rowsum(dO * O).
 *   backward  csrc/flash_attn/src/flash_bwd_kernel.h:28-825 (dQ), :842-1676 (dK, dV):
 *             P = exp(s - LSE) (:474), dP = dO.V (:398-407), dS = P (dP - D) (:490),
 *             dQ = scale * sum_j r(dS) K (:512,530,765), dK = scale * sum_i r(dS) Q (:1360,1399,1652),
 *             dV = sum_i r(P) dO (:1359,1380); GQA: K/V head = h / (H/H_k) (flash_fwd_kernel.h:74,91),
 *             dK/dV summed over the group (flash_api.cpp:301-312).
 *   varlen    csrc/flash_attn/src/block_info.h:3-27: packed rows + cu_seqlens offsets, padded LSE.
 * The tiling, the online-softmax rescaling order and fast-math approximations of the reference are
 * NOT restated: this is the exact-arithmetic statement of the same function with the
 * reference's explicit rounding points (r() = round to fp16 / bf16, nearest-even).  Dot
 * products accumulate in double so the oracle is the most accurate member of the family the
 * reference's tolerances (test_flash_attn.py:407-414) are written against.  One deliberate
 * difference: with GQA the reference rounds each q-head's dK/dV to fp16 and then sums them in
 * fp16 (flash_api.cpp:265-272,301-312); the oracle (like our HIP kernel) sums the group in high
 * precision and rounds once.
 *
 * Pinning: oracle/pin_oracle.py checks this file against the reference's own Python oracles
 * (test_flash_attn.py:134-248 vanilla_attention_ref / memory_efficient_attention_ref) imported
 * from /root/reference in the dev container, and tests/golden/ holds vectors generated from
 * those reference functions (tests/golden/make_golden.py).  LSE has no reference test
 * ("parity unpinned" for LSE alone): it is pinned to torch.logsumexp instead.
 *
 * Layout: all tensors fp32 in memory (values already representable in the low precision
 * type when rounding is on), (batch, seqlen, heads, d) contiguous; varlen (total, heads, d).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- low precision rounding (nearest even), value returned as float ---------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float oracle_round_bf16(float f) {
    uint32_t u = f2u(f);
    if ((u & 0x7f800000u) == 0x7f800000u) return f;           /* inf / nan */
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    return u2f(u);
}

float oracle_round_fp16(float f) {
    uint32_t u = f2u(f);
    uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return f;                           /* inf / nan */
    if (a >= 0x477ff000u) return u2f(sign | 0x7f800000u);     /* >= 65520 rounds to inf */
    if (a < 0x38800000u) {                                    /* below 2^-14: fp16 subnormal grid, step 2^-24 */
        float x = u2f(a);
        float r = nearbyintf(x * 16777216.0f) / 16777216.0f;  /* default rounding mode = nearest even */
        return u2f(sign | f2u(r));
    }
    a += 0xfffu + ((a >> 13) & 1u);                           /* 13 dropped mantissa bits */
    a &= 0xffffe000u;
    return u2f(sign | a);
}

static inline float round_lp(float f, int mode) {
    return mode == 1 ? oracle_round_fp16(f) : (mode == 2 ? oracle_round_bf16(f) : f);
}

/* ---- one (batch, head) problem ------------------------------------------------------------- */
typedef struct {
    const float *q, *k, *v;       /* first row of this sequence, this head */
    int64_t q_rs, k_rs, v_rs;     /* row strides in elements */
    int sq, sk, d, causal, mode;
    float scale;
} problem_t;

static inline int visible_end(const problem_t* p, int i) {
    /* number of leading keys visible to query i: j < end (mask.h:52-62,172) */
    if (!p->causal) return p->sk;
    long e = (long)i + (long)(p->sk - p->sq) + 1;
    if (e < 0) e = 0;
    if (e > p->sk) e = p->sk;
    return (int)e;
}

static void scores_row(const problem_t* p, int i, int end, float* s) {
    const float* qi = p->q + (int64_t)i * p->q_rs;
    for (int j = 0; j < end; ++j) {
        const float* kj = p->k + (int64_t)j * p->k_rs;
        double acc = 0.0;
        for (int c = 0; c < p->d; ++c) acc += (double)qi[c] * (double)kj[c];
        s[j] = (float)(acc * (double)p->scale);
    }
}

static void fwd_problem(const problem_t* p, float* o, int64_t o_rs, float* lse) {
    float* s = (float*)malloc(sizeof(float) * (size_t)(p->sk > 0 ? p->sk : 1));
    double* acc = (double*)malloc(sizeof(double) * (size_t)p->d);
    for (int i = 0; i < p->sq; ++i) {
        float* oi = o + (int64_t)i * o_rs;
        const int end = visible_end(p, i);
        if (end == 0) {                                   /* dead row */
            for (int c = 0; c < p->d; ++c) oi[c] = 0.f;
            lse[i] = 0.f;
            continue;
        }
        scores_row(p, i, end, s);
        float m = s[0];
        for (int j = 1; j < end; ++j) m = s[j] > m ? s[j] : m;
        double l = 0.0;
        for (int c = 0; c < p->d; ++c) acc[c] = 0.0;
        for (int j = 0; j < end; ++j) {
            const float e = expf(s[j] - m);
            l += (double)e;
            const float pr = round_lp(e, p->mode);        /* P rounded before P.V */
            const float* vj = p->v + (int64_t)j * p->v_rs;
            for (int c = 0; c < p->d; ++c) acc[c] += (double)pr * (double)vj[c];
        }
        for (int c = 0; c < p->d; ++c) oi[c] = round_lp((float)(acc[c] / l), p->mode);
        lse[i] = (float)((double)m + log(l));
    }
    free(s);
    free(acc);
}

/* dq: (sq, d) rows of this head, overwritten. dk/dv: double accumulators (sk, d), ADDED to. */
static void bwd_problem(const problem_t* p, const float* o, int64_t o_rs, const float* dout, int64_t do_rs,
                        const float* lse, float* dq, int64_t dq_rs, double* dk_acc, double* dv_acc) {
    float* s = (float*)malloc(sizeof(float) * (size_t)(p->sk > 0 ? p->sk : 1));
    double* acc = (double*)malloc(sizeof(double) * (size_t)p->d);
    for (int i = 0; i < p->sq; ++i) {
        const float* qi = p->q + (int64_t)i * p->q_rs;
        const float* oi = o + (int64_t)i * o_rs;
        const float* doi = dout + (int64_t)i * do_rs;
        float* dqi = dq + (int64_t)i * dq_rs;
        const int end = visible_end(p, i);
        for (int c = 0; c < p->d; ++c) acc[c] = 0.0;
        if (end > 0) {
            double Dsum = 0.0;                            /* D_i = sum_d dO * O */
            for (int c = 0; c < p->d; ++c) Dsum += (double)doi[c] * (double)oi[c];
            const float Di = (float)Dsum;
            scores_row(p, i, end, s);
            for (int j = 0; j < end; ++j) {
                const float* kj = p->k + (int64_t)j * p->k_rs;
                const float* vj = p->v + (int64_t)j * p->v_rs;
                const float pij = expf(s[j] - lse[i]);
                double dp = 0.0;
                for (int c = 0; c < p->d; ++c) dp += (double)doi[c] * (double)vj[c];
                const float ds = pij * ((float)dp - Di);
                const float pr = round_lp(pij, p->mode), dsr = round_lp(ds, p->mode);
                double* dkj = dk_acc + (int64_t)j * p->d;
                double* dvj = dv_acc + (int64_t)j * p->d;
                for (int c = 0; c < p->d; ++c) {
                    acc[c] += (double)dsr * (double)kj[c];
                    dkj[c] += (double)dsr * (double)qi[c];
                    dvj[c] += (double)pr * (double)doi[c];
                }
            }
        }
        for (int c = 0; c < p->d; ++c) dqi[c] = round_lp((float)(acc[c] * (double)p->scale), p->mode);
    }
    free(s);
    free(acc);
}

/* ---- public entry points --------------------------------------------------------------------- */
/* cu_q/cu_k NULL: fixed length (b, s, h, d). Otherwise packed (total, h, d) with lse (b, h, max_sq). */
void oracle_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse,
                     const int32_t* cu_q, const int32_t* cu_k,
                     int b, int sq, int sk, int h, int hk, int d, int causal, int round_mode) {
    const int ratio = h / hk;
    const float scale = 1.0f / sqrtf((float)d);
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int bi = 0; bi < b; ++bi)
        for (int hi = 0; hi < h; ++hi) {
            problem_t p;
            int64_t q0, k0;
            if (cu_q) { q0 = cu_q[bi]; k0 = cu_k[bi]; p.sq = cu_q[bi + 1] - cu_q[bi]; p.sk = cu_k[bi + 1] - cu_k[bi]; }
            else { q0 = (int64_t)bi * sq; k0 = (int64_t)bi * sk; p.sq = sq; p.sk = sk; }
            p.q_rs = (int64_t)h * d; p.k_rs = p.v_rs = (int64_t)hk * d;
            p.q = q + q0 * p.q_rs + (int64_t)hi * d;
            p.k = k + k0 * p.k_rs + (int64_t)(hi / ratio) * d;
            p.v = v + k0 * p.v_rs + (int64_t)(hi / ratio) * d;
            p.d = d; p.causal = causal; p.mode = round_mode; p.scale = scale;
            fwd_problem(&p, o + q0 * p.q_rs + (int64_t)hi * d, p.q_rs, lse + ((int64_t)bi * h + hi) * sq);
        }
}

void oracle_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* lse, const float* dout,
                     float* dq, float* dk, float* dv,
                     const int32_t* cu_q, const int32_t* cu_k,
                     int b, int sq, int sk, int h, int hk, int d, int causal, int round_mode) {
    const int ratio = h / hk;
    const float scale = 1.0f / sqrtf((float)d);
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int bi = 0; bi < b; ++bi)
        for (int hki = 0; hki < hk; ++hki) {
            int64_t q0, k0;
            int sqb, skb;
            if (cu_q) { q0 = cu_q[bi]; k0 = cu_k[bi]; sqb = cu_q[bi + 1] - cu_q[bi]; skb = cu_k[bi + 1] - cu_k[bi]; }
            else { q0 = (int64_t)bi * sq; k0 = (int64_t)bi * sk; sqb = sq; skb = sk; }
            const int64_t q_rs = (int64_t)h * d, k_rs = (int64_t)hk * d;
            double* dk_acc = (double*)calloc((size_t)(skb > 0 ? skb : 1) * d, sizeof(double));
            double* dv_acc = (double*)calloc((size_t)(skb > 0 ? skb : 1) * d, sizeof(double));
            for (int g = 0; g < ratio; ++g) {             /* GQA group sharing this kv head */
                const int hi = hki * ratio + g;
                problem_t p;
                p.sq = sqb; p.sk = skb; p.q_rs = q_rs; p.k_rs = p.v_rs = k_rs;
                p.q = q + q0 * q_rs + (int64_t)hi * d;
                p.k = k + k0 * k_rs + (int64_t)hki * d;
                p.v = v + k0 * k_rs + (int64_t)hki * d;
                p.d = d; p.causal = causal; p.mode = round_mode; p.scale = scale;
                bwd_problem(&p, o + q0 * q_rs + (int64_t)hi * d, q_rs, dout + q0 * q_rs + (int64_t)hi * d, q_rs,
                            lse + ((int64_t)bi * h + hi) * sq, dq + q0 * q_rs + (int64_t)hi * d, q_rs, dk_acc, dv_acc);
            }
            for (int j = 0; j < skb; ++j)
                for (int c = 0; c < d; ++c) {
                    dk[(k0 + j) * k_rs + (int64_t)hki * d + c] = round_lp((float)(dk_acc[(int64_t)j * d + c] * (double)scale), round_mode);
                    dv[(k0 + j) * k_rs + (int64_t)hki * d + c] = round_lp((float)dv_acc[(int64_t)j * d + c], round_mode);
                }
            free(dk_acc);
            free(dv_acc);
        }
}

/* D = rowsum(dO * O), layout of lse (flash_bwd_preprocess_kernel.h:81-93) */
void oracle_dot_do_o(const float* o, const float* dout, float* dsum, const int32_t* cu_q,
                     int b, int sq, int h, int d) {
    for (int bi = 0; bi < b; ++bi) {
        const int64_t q0 = cu_q ? cu_q[bi] : (int64_t)bi * sq;
        const int sqb = cu_q ? cu_q[bi + 1] - cu_q[bi] : sq;
        for (int hi = 0; hi < h; ++hi)
            for (int i = 0; i < sqb; ++i) {
                const float* a = o + ((q0 + i) * h + hi) * (int64_t)d;
                const float* g = dout + ((q0 + i) * h + hi) * (int64_t)d;
                double acc = 0.0;
                for (int c = 0; c < d; ++c) acc += (double)a[c] * (double)g[c];
                dsum[((int64_t)bi * h + hi) * sq + i] = (float)acc;
            }
    }
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
