/*
 * tf_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see tf_oracle.h).
 *
 * Plain-C restatement of the reference algorithms on the hot path
 * (Neptune-Crypto/twenty-first v2.0.2; citations relative to twenty-first/src/).
 * It is deliberately the *reference's* algorithm (radix-2 DIT with bit-reversal, scalar
 * Tip5, heap-layout Merkle sweep), not the GPU algorithm, so that parity is a real check.
 */
#include "tf_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

#define P TFO_P
#define EPS 0xffffffffULL /* 2^32 - 1 = 2^64 mod p = 1 + !P */

/* ------------------------------------------------------------------ BFieldElement */

/* math/b_field_element.rs:357-370 */
static inline u64 montyred(u128 x) {
    u64 xl = (u64)x;
    u64 xh = (u64)(x >> 64);
    u64 a = xl + (xl << 32);
    u64 e = a < xl; /* overflowing_add carry */
    u64 b = a - (a >> 32) - e;
    u64 r = xh - b;
    u64 c = xh < b; /* overflowing_sub borrow */
    return r - EPS * c;
}

uint64_t tfo_montyred(uint64_t lo, uint64_t hi) { return montyred(((u128)hi << 64) | lo); }

#define R2 0xfffffffe00000001ULL /* 2^128 mod p, b_field_element.rs:229 */

static inline u64 bfe_new(u64 v) { return montyred((u128)v * R2); }  /* :235-237 */
static inline u64 bfe_value(u64 raw) { return montyred((u128)raw); } /* :334-336 */

/* :711-732 : a + b = a - (p - b), +p on borrow */
static inline u64 bfe_add(u64 a, u64 b) {
    u64 t = P - b;
    u64 x1 = a - t;
    return (a < t) ? x1 + P : x1;
}
/* :773-795 */
static inline u64 bfe_sub(u64 a, u64 b) {
    u64 x1 = a - b;
    u64 c1 = a < b;
    return x1 - EPS * c1;
}
static inline u64 bfe_neg(u64 a) { return bfe_sub(0, a); }                /* :764-771 */
static inline u64 bfe_mul(u64 a, u64 b) { return montyred((u128)a * b); } /* :755-762 */

#define BFE_ONE 0xffffffffULL /* new(1), :707-709 */

/* :340-353 MSB-first square-and-multiply */
static u64 bfe_mod_pow(u64 base, u64 exp) {
    u64 acc = BFE_ONE;
    int bit_length = exp ? 64 - __builtin_clzll(exp) : 0;
    for (int i = 0; i < bit_length; i++) {
        acc = bfe_mul(acc, acc);
        if (exp & (1ULL << (bit_length - 1 - i))) acc = bfe_mul(acc, base);
    }
    return acc;
}

static inline u64 bfe_exp_sq(u64 base, int times) {
    for (int i = 0; i < times; i++) base = bfe_mul(base, base);
    return base;
}

/* :254-284 fixed addition chain for x^(p-2); 0 -> 0 (inverse_or_zero, traits.rs:39-45) */
static u64 bfe_inverse(u64 x) {
    if (x == 0) return 0;
    u64 bin_2_ones = bfe_mul(bfe_mul(x, x), x);
    u64 bin_3_ones = bfe_mul(bfe_mul(bin_2_ones, bin_2_ones), x);
    u64 bin_6_ones = bfe_mul(bfe_exp_sq(bin_3_ones, 3), bin_3_ones);
    u64 bin_12_ones = bfe_mul(bfe_exp_sq(bin_6_ones, 6), bin_6_ones);
    u64 bin_24_ones = bfe_mul(bfe_exp_sq(bin_12_ones, 12), bin_12_ones);
    u64 bin_30_ones = bfe_mul(bfe_exp_sq(bin_24_ones, 6), bin_6_ones);
    u64 bin_31_ones = bfe_mul(bfe_mul(bin_30_ones, bin_30_ones), x);
    u64 bin_31_ones_1_zero = bfe_mul(bin_31_ones, bin_31_ones);
    u64 bin_32_ones = bfe_mul(bfe_mul(bin_31_ones, bin_31_ones), x);
    return bfe_mul(bfe_exp_sq(bin_31_ones_1_zero, 32), bin_32_ones);
}

uint64_t tfo_bfe_new(uint64_t v) { return bfe_new(v); }
uint64_t tfo_bfe_value(uint64_t r) { return bfe_value(r); }
uint64_t tfo_bfe_add(uint64_t a, uint64_t b) { return bfe_add(a, b); }
uint64_t tfo_bfe_sub(uint64_t a, uint64_t b) { return bfe_sub(a, b); }
uint64_t tfo_bfe_neg(uint64_t a) { return bfe_neg(a); }
uint64_t tfo_bfe_mul(uint64_t a, uint64_t b) { return bfe_mul(a, b); }
uint64_t tfo_bfe_mod_pow(uint64_t b, uint64_t e) { return bfe_mod_pow(b, e); }
uint64_t tfo_bfe_inverse(uint64_t a) { return bfe_inverse(a); }

/* math/b_field_element.rs:43-78 : canonical values of the primitive 2^k-th roots, k = 0..32 */
static const u64 PRIMITIVE_ROOTS[33] = {
    1ULL,
    18446744069414584320ULL,
    281474976710656ULL,
    18446744069397807105ULL,
    17293822564807737345ULL,
    70368744161280ULL,
    549755813888ULL,
    17870292113338400769ULL,
    13797081185216407910ULL,
    1803076106186727246ULL,
    11353340290879379826ULL,
    455906449640507599ULL,
    17492915097719143606ULL,
    1532612707718625687ULL,
    16207902636198568418ULL,
    17776499369601055404ULL,
    6115771955107415310ULL,
    12380578893860276750ULL,
    9306717745644682924ULL,
    18146160046829613826ULL,
    3511170319078647661ULL,
    17654865857378133588ULL,
    5416168637041100469ULL,
    16905767614792059275ULL,
    9713644485405565297ULL,
    5456943929260765144ULL,
    17096174751763063430ULL,
    1213594585890690845ULL,
    6414415596519834757ULL,
    16116352524544190054ULL,
    9123114210336311365ULL,
    4614640910117430873ULL,
    1753635133440165772ULL,
};

/* :814-818 ; the phf map also has 0 => 1 (dummy for the empty slice) */
uint64_t tfo_bfe_primitive_root(uint64_t n) {
    if (n == 0) return bfe_new(1);
    if (n & (n - 1)) return 0;
    int k = __builtin_ctzll(n);
    if (k > 32) return 0;
    return bfe_new(PRIMITIVE_ROOTS[k]);
}

/* ------------------------------------------------------------------ XFieldElement */

void tfo_xfe_add(const u64 a[3], const u64 b[3], u64 out[3]) { /* x_field_element.rs:479-489 */
    for (int i = 0; i < 3; i++) out[i] = bfe_add(a[i], b[i]);
}
void tfo_xfe_sub(const u64 a[3], const u64 b[3], u64 out[3]) { /* :570-577 : a + (-b) */
    for (int i = 0; i < 3; i++) out[i] = bfe_add(a[i], bfe_neg(b[i]));
}
/* :512-536, with self = [c, b, a], other = [f, e, d] */
void tfo_xfe_mul(const u64 s[3], const u64 o[3], u64 out[3]) {
    u64 c = s[0], b = s[1], a = s[2];
    u64 f = o[0], e = o[1], d = o[2];
    u64 ae = bfe_mul(a, e), bd = bfe_mul(b, d), ad = bfe_mul(a, d);
    u64 r0 = bfe_sub(bfe_sub(bfe_mul(c, f), ae), bd);
    u64 r1 = bfe_add(bfe_add(bfe_sub(bfe_add(bfe_mul(b, f), bfe_mul(c, e)), ad), ae), bd);
    u64 r2 = bfe_add(bfe_add(bfe_add(bfe_mul(a, f), bfe_mul(b, e)), bfe_mul(c, d)), ad);
    out[0] = r0;
    out[1] = r1;
    out[2] = r2;
}
void tfo_xfe_mul_bfe(const u64 a[3], u64 b, u64 out[3]) { /* :540-548 */
    for (int i = 0; i < 3; i++) out[i] = bfe_mul(a[i], b);
}

/* ------------------------------------------------------------------ NTT (math/ntt.rs) */

/* ntt.rs:241-248 */
static inline uint32_t bitreverse(uint32_t k, uint32_t log2_n) {
    k = ((k & 0x55555555u) << 1) | ((k & 0xaaaaaaaau) >> 1);
    k = ((k & 0x33333333u) << 2) | ((k & 0xccccccccu) >> 2);
    k = ((k & 0x0f0f0f0fu) << 4) | ((k & 0xf0f0f0f0u) >> 4);
    k = ((k & 0x00ff00ffu) << 8) | ((k & 0xff00ff00u) >> 8);
    k = (k >> 16) | (k << 16);
    return k >> ((32 - log2_n) & 0x1f);
}

/* Per-size caches like the reference's OnceLock arrays (ntt.rs:71-72, :113-114, :166-167).
 * The twiddles of all stages are stored back to back: stage i (m = 2^i) at offset m - 1. */
typedef struct {
    u64 *tw[2]; /* [0] forward, [1] inverse; n - 1 words each */
} ntt_cache_t;
static ntt_cache_t g_cache[32];
static pthread_mutex_t g_cache_lock = PTHREAD_MUTEX_INITIALIZER;

/* ntt.rs:309-324 */
static u64 *twiddle_factors(uint32_t n, u64 root) {
    uint32_t log2n = n ? 31 - __builtin_clz(n) : 0;
    u64 *tw = (u64 *)malloc(sizeof(u64) * (n > 1 ? n - 1 : 1));
    for (uint32_t i = 0; i < log2n; i++) {
        uint32_t m = 1u << i;
        uint32_t exponent = n / (2 * m);
        u64 w_m = bfe_mod_pow(root, exponent);
        u64 *w = tw + (m - 1);
        w[0] = BFE_ONE;
        for (uint32_t j = 1; j < m; j++) w[j] = bfe_mul(w[j - 1], w_m);
    }
    return tw;
}

static const u64 *get_twiddles(uint32_t n, int inverse) {
    uint32_t log2n = 31 - __builtin_clz(n);
    pthread_mutex_lock(&g_cache_lock);
    if (!g_cache[log2n].tw[inverse]) {
        u64 omega = tfo_bfe_primitive_root(n);
        if (inverse) omega = bfe_inverse(omega); /* ntt.rs:120 */
        g_cache[log2n].tw[inverse] = twiddle_factors(n, omega);
    }
    const u64 *t = g_cache[log2n].tw[inverse];
    pthread_mutex_unlock(&g_cache_lock);
    return t;
}

/* ntt.rs:153-215, BFE */
static void ntt_unchecked_1(u64 *x, uint32_t n, const u64 *tw) {
    uint32_t log2n = 31 - __builtin_clz(n);
    for (uint32_t k = 0; k < n; k++) { /* :189-193 */
        uint32_t rk = bitreverse(k, log2n);
        if (k < rk) {
            u64 t = x[k];
            x[k] = x[rk];
            x[rk] = t;
        }
    }
    for (uint32_t m = 1; m < n; m *= 2) { /* :195-214 */
        const u64 *w = tw + (m - 1);
        for (uint32_t k = 0; k < n; k += 2 * m) {
            for (uint32_t j = 0; j < m; j++) {
                u64 u = x[k + j];
                u64 v = bfe_mul(x[k + j + m], w[j]);
                x[k + j] = bfe_add(u, v);
                x[k + j + m] = bfe_sub(u, v);
            }
        }
    }
}

/* same, XFE: MulAssign<BFE> on all three coefficients (x_field_element.rs:620-625, :540-548);
 * XFE Sub is a + (-b) (:570-577) which equals the coefficient-wise difference. */
static void ntt_unchecked_3(u64 *x, uint32_t n, const u64 *tw) {
    uint32_t log2n = 31 - __builtin_clz(n);
    for (uint32_t k = 0; k < n; k++) {
        uint32_t rk = bitreverse(k, log2n);
        if (k < rk) {
            for (int c = 0; c < 3; c++) {
                u64 t = x[3 * (size_t)k + c];
                x[3 * (size_t)k + c] = x[3 * (size_t)rk + c];
                x[3 * (size_t)rk + c] = t;
            }
        }
    }
    for (uint32_t m = 1; m < n; m *= 2) {
        const u64 *w = tw + (m - 1);
        for (uint32_t k = 0; k < n; k += 2 * m) {
            for (uint32_t j = 0; j < m; j++) {
                u64 *pu = x + 3 * (size_t)(k + j);
                u64 *pv = x + 3 * (size_t)(k + j + m);
                for (int c = 0; c < 3; c++) {
                    u64 u = pu[c];
                    u64 v = bfe_mul(pv[c], w[j]);
                    pu[c] = bfe_add(u, v);
                    pv[c] = bfe_add(u, bfe_neg(v));
                }
            }
        }
    }
}

/* ntt.rs:135-140 : len must fit u32 and be 0 or a power of two.  (The reference's cache
 * array has 32 slots, so the largest usable length is 2^31.) */
static int check_len(size_t n) {
    if (n > 0xffffffffULL) return 1;
    if (n != 0 && (n & (n - 1))) return 2;
    return 0;
}

static int ntt_any(u64 *x, size_t n, int width, int inverse) {
    int rc = check_len(n);
    if (rc) return rc;
    if (width != 1 && width != 3) return 3;
    if (n == 0) return 0; /* ntt.rs:170-173 */
    if (n > 1) {
        const u64 *tw = get_twiddles((uint32_t)n, inverse);
        if (width == 1)
            ntt_unchecked_1(x, (uint32_t)n, tw);
        else
            ntt_unchecked_3(x, (uint32_t)n, tw);
    }
    if (inverse) { /* unscale, ntt.rs:220-228 */
        u64 n_inv = bfe_inverse(bfe_new((u64)n));
        for (size_t i = 0; i < n * (size_t)width; i++) x[i] = bfe_mul(x[i], n_inv);
    }
    return 0;
}

int tfo_ntt(uint64_t *x, size_t n, int width) { return ntt_any(x, n, width, 0); }
int tfo_intt(uint64_t *x, size_t n, int width) { return ntt_any(x, n, width, 1); }

/* Run `worker(job)` on `threads` threads: threads - 1 created ones and the caller's.  A thread that cannot be created (a
 * cgroup-limited host) is simply not used -- nothing uninitialised is ever joined, and with none at all the caller does the job. */
static void fan_out(void *(*worker)(void *), void *job, int threads) {
    pthread_t *tid = threads > 1 ? (pthread_t *)malloc(sizeof(pthread_t) * (size_t)(threads - 1)) : NULL;
    int made = 0;
    if (tid)
        for (int t = 0; t < threads - 1; t++)
            if (pthread_create(&tid[made], NULL, worker, job) == 0) made++;
    worker(job);
    for (int t = 0; t < made; t++) pthread_join(tid[t], NULL);
    free(tid);
}

typedef struct {
    u64 *x;
    size_t n, batch;
    int width, inverse;
    size_t next; /* shared work counter */
    pthread_mutex_t *lock;
    int rc;
} ntt_job_t;

static void *ntt_worker(void *arg) {
    ntt_job_t *job = (ntt_job_t *)arg;
    for (;;) {
        pthread_mutex_lock(job->lock);
        size_t b = job->next++;
        pthread_mutex_unlock(job->lock);
        if (b >= job->batch) break;
        int rc = ntt_any(job->x + b * job->n * (size_t)job->width, job->n, job->width, job->inverse);
        if (rc) {
            pthread_mutex_lock(job->lock);
            job->rc = rc;
            pthread_mutex_unlock(job->lock);
        }
    }
    return NULL;
}

int tfo_ntt_batch(uint64_t *x, size_t n, size_t batch, int width, int inverse, int threads) {
    int rc = check_len(n);
    if (rc) return rc;
    if (threads <= 1) {
        for (size_t b = 0; b < batch; b++) {
            rc = ntt_any(x + b * n * (size_t)width, n, width, inverse);
            if (rc) return rc;
        }
        return 0;
    }
    if (n > 1) (void)get_twiddles((uint32_t)n, inverse); /* build the cache before fan-out */
    pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
    ntt_job_t job = {x, n, batch, width, inverse, 0, &lock, 0};
    fan_out(ntt_worker, &job, threads);
    return job.rc;
}

/* ------------------------------------------------------------------ Polynomial */

/* polynomial.rs:760-773 : sequential power chain */
void tfo_poly_scale(uint64_t *c, size_t n_coeffs, int width, uint64_t alpha) {
    u64 power = BFE_ONE;
    for (size_t i = 0; i < n_coeffs; i++) {
        for (int k = 0; k < width; k++) c[i * (size_t)width + k] = bfe_mul(c[i * (size_t)width + k], power);
        power = bfe_mul(power, alpha);
    }
}

static size_t poly_len_trimmed(const u64 *c, size_t n_coeffs, int width) {
    /* degree() ignores leading (high-order) zero coefficients */
    while (n_coeffs > 0) {
        int nz = 0;
        for (int k = 0; k < width; k++) nz |= c[(n_coeffs - 1) * (size_t)width + k] != 0;
        if (nz) break;
        n_coeffs--;
    }
    return n_coeffs;
}

/* polynomial.rs:1374-1399 */
int tfo_coset_evaluate(const uint64_t *coeffs, size_t n_coeffs, int width, uint64_t offset, uint64_t *out,
                       size_t order) {
    size_t len = poly_len_trimmed(coeffs, n_coeffs, width);
    /* assert order > degree  (:1388-1392); degree = len - 1 (or -1 for the zero polynomial) */
    if ((long long)order <= (long long)len - 1) return 4;
    int rc = check_len(order);
    if (rc) return rc;
    memset(out, 0, order * (size_t)width * sizeof(u64)); /* resize(order, ZERO) :1395 */
    memcpy(out, coeffs, len * (size_t)width * sizeof(u64));
    tfo_poly_scale(out, len, width, offset); /* :1394 */
    return ntt_any(out, order, width, 0);    /* :1396 */
}

/* `batch` polynomials of n_coeffs coefficients each, contiguous in and out, one polynomial per thread: what a rayon caller of the
 * single-threaded fast_coset_evaluate does (the reference itself is sequential per polynomial, polynomial.rs:1374-1399; bench
 * shape benches/polynomial_coset.rs:15-47).  bench.py's all-core cpu_baseline of BASELINE configs[3]. */
typedef struct {
    const u64 *coeffs;
    u64 *out;
    size_t n_coeffs, order, batch;
    int width;
    u64 offset;
    size_t next;
    pthread_mutex_t *lock;
    int rc;
} coset_job_t;

static void *coset_worker(void *arg) {
    coset_job_t *job = (coset_job_t *)arg;
    for (;;) {
        pthread_mutex_lock(job->lock);
        size_t b = job->next++;
        pthread_mutex_unlock(job->lock);
        if (b >= job->batch) break;
        int rc = tfo_coset_evaluate(job->coeffs + b * job->n_coeffs * (size_t)job->width, job->n_coeffs, job->width, job->offset,
                                    job->out + b * job->order * (size_t)job->width, job->order);
        if (rc) {
            pthread_mutex_lock(job->lock);
            job->rc = rc;
            pthread_mutex_unlock(job->lock);
        }
    }
    return NULL;
}

int tfo_coset_evaluate_batch(const uint64_t *coeffs, size_t n_coeffs, int width, uint64_t offset, uint64_t *out, size_t order,
                             size_t batch, int threads) {
    int rc = check_len(order);
    if (rc) return rc;
    if (order > 1) (void)get_twiddles((uint32_t)order, 0); /* build the cache before fan-out */
    pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
    coset_job_t job = {coeffs, out, n_coeffs, order, batch, width, offset, 0, &lock, 0};
    fan_out(coset_worker, &job, threads);
    return job.rc;
}

/* polynomial.rs:1907-1918 : intt then scale by offset^-1 */
int tfo_coset_interpolate(const uint64_t *values, size_t n, int width, uint64_t offset, uint64_t *out) {
    memcpy(out, values, n * (size_t)width * sizeof(u64));
    int rc = ntt_any(out, n, width, 1);
    if (rc) return rc;
    tfo_poly_scale(out, n, width, bfe_inverse(offset));
    return 0;
}

void tfo_poly_eval(const uint64_t *c, size_t n_coeffs, int width, uint64_t point, uint64_t *out) {
    for (int k = 0; k < width; k++) out[k] = 0;
    for (size_t i = n_coeffs; i-- > 0;) {
        for (int k = 0; k < width; k++) out[k] = bfe_add(bfe_mul(out[k], point), c[i * (size_t)width + k]);
    }
}

/* ------------------------------------------------------------------ Tip5 (tip5/mod.rs) */

#define STATE_SIZE 16
#define NUM_SPLIT_AND_LOOKUP 4
#define RATE 10
#define NUM_ROUNDS 5
#define DIGEST_LEN 5

/* tip5/mod.rs:50-64 ; L(x) = ((x+1)^3 mod 257) - 1  (:1022-1026) -- generated, and checked
 * against the formula by the KAT tests rather than typed in. */
static uint8_t LOOKUP_TABLE[256];

/* tip5/mod.rs:68-149 : canonical values */
static const u64 ROUND_CONSTANTS_CANONICAL[NUM_ROUNDS * STATE_SIZE] = {
    13630775303355457758ULL, 16896927574093233874ULL, 10379449653650130495ULL, 1965408364413093495ULL,
    15232538947090185111ULL, 15892634398091747074ULL, 3989134140024871768ULL,  2851411912127730865ULL,
    8709136439293758776ULL,  3694858669662939734ULL,  12692440244315327141ULL, 10722316166358076749ULL,
    12745429320441639448ULL, 17932424223723990421ULL, 7558102534867937463ULL,  15551047435855531404ULL,
    17532528648579384106ULL, 5216785850422679555ULL,  15418071332095031847ULL, 11921929762955146258ULL,
    9738718993677019874ULL,  3464580399432997147ULL,  13408434769117164050ULL, 264428218649616431ULL,
    4436247869008081381ULL,  4063129435850804221ULL,  2865073155741120117ULL,  5749834437609765994ULL,
    6804196764189408435ULL,  17060469201292988508ULL, 9475383556737206708ULL,  12876344085611465020ULL,
    13835756199368269249ULL, 1648753455944344172ULL,  9836124473569258483ULL,  12867641597107932229ULL,
    11254152636692960595ULL, 16550832737139861108ULL, 11861573970480733262ULL, 1256660473588673495ULL,
    13879506000676455136ULL, 10564103842682358721ULL, 16142842524796397521ULL, 3287098591948630584ULL,
    685911471061284805ULL,   5285298776918878023ULL,  18310953571768047354ULL, 3142266350630002035ULL,
    549990724933663297ULL,   4901984846118077401ULL,  11458643033696775769ULL, 8706785264119212710ULL,
    12521758138015724072ULL, 11877914062416978196ULL, 11333318251134523752ULL, 3933899631278608623ULL,
    16635128972021157924ULL, 10291337173108950450ULL, 4142107155024199350ULL,  16973934533787743537ULL,
    11068111539125175221ULL, 17546769694830203606ULL, 5315217744825068993ULL,  4609594252909613081ULL,
    3350107164315270407ULL,  17715942834299349177ULL, 9600609149219873996ULL,  12894357635820003949ULL,
    4597649658040514631ULL,  7735563950920491847ULL,  1663379455870887181ULL,  13889298103638829706ULL,
    7375530351220884434ULL,  3502022433285269151ULL,  9231805330431056952ULL,  9252272755288523725ULL,
    10014268662326746219ULL, 15565031632950843234ULL, 1209725273521819323ULL,  6024642864597845108ULL,
};
static u64 ROUND_CONSTANTS[NUM_ROUNDS * STATE_SIZE]; /* Montgomery form */

/* tip5/mod.rs:154-157 */
static const u64 MDS_MATRIX_FIRST_COLUMN[STATE_SIZE] = {
    61402, 1108, 28750, 33823, 7454, 43244, 53865, 12034, 56951, 27521, 41351, 40901, 12021, 59689, 26798, 17845,
};

static pthread_once_t g_tip5_once = PTHREAD_ONCE_INIT;
static void tip5_init_tables(void) {
    for (int x = 0; x < 256; x++) {
        u64 xx = (u64)x + 1;
        LOOKUP_TABLE[x] = (uint8_t)(((xx * xx * xx) + 256) % 257); /* offset_fermat_cube_map :1022-1026 */
    }
    for (int i = 0; i < NUM_ROUNDS * STATE_SIZE; i++) ROUND_CONSTANTS[i] = bfe_new(ROUND_CONSTANTS_CANONICAL[i]);
}

/* :197-207 : byte-wise lookup on the raw (Montgomery) little-endian bytes */
static inline u64 split_and_lookup(u64 raw) {
    u64 out = 0;
    for (int i = 0; i < 8; i++) out |= (u64)LOOKUP_TABLE[(raw >> (8 * i)) & 0xff] << (8 * i);
    return out;
}

/* :184-194 */
static inline void sbox_layer(u64 s[STATE_SIZE]) {
    for (int i = 0; i < NUM_SPLIT_AND_LOOKUP; i++) s[i] = split_and_lookup(s[i]);
    for (int i = NUM_SPLIT_AND_LOOKUP; i < STATE_SIZE; i++) {
        u64 sq = bfe_mul(s[i], s[i]);
        u64 qu = bfe_mul(sq, sq);
        s[i] = bfe_mul(s[i], bfe_mul(sq, qu));
    }
}

/* ---- the MDS layer, three ways ---------------------------------------------------------------------------------------
 * (0) mds_plain:     the integer circulant sum  s_r = sum_c M[(r - c) mod 16] * raw[c]  formed directly in 128 bits, then the
 *                    reduction of :244-252 (the round-1 form; it is also the shortcut the GPU kernel takes).
 * (1) mds_cyclomul:  restatement of the reference's own second method, Tip5::mds_cyclomul + fast_cyclomul16/8/4/2,
 *                    complex_negacyclomul8/4/2, complex_karatsuba4/2 (tip5/mod.rs:753-1019): cyclic convolution of the 32-bit
 *                    halves with the MDS column over Z by the CRT split x^n - 1 = (x^(n/2) - 1)(x^(n/2) + 1), the negacyclic
 *                    half as a product in Z[i][x] by Karatsuba; recombination and reduction as :766-777.
 * (2) mds_generated: the shape of Tip5::mds_generated + generated_function (:210-506) -- the same CRT/Karatsuba graph with the
 *                    MDS column as the constant operand, every halving deferred (so the graph returns 16 x the convolution),
 *                    all arithmetic WRAPPING on u64 exactly as the reference's wrapping_add / wrapping_sub / wrapping_mul,
 *                    recombined as s = (lo >> 4) + (hi << 28) and reduced as :244-252 (including the possibly degenerate
 *                    result >= p that the round-constant addition repairs).  The reference's 250-line node list is that graph
 *                    unrolled with its constants folded; the constants it hard-codes for the fully split components,
 *                    524757 = sum M_i and 52427 = sum (-1)^i M_i (node_64 / node_67, :283-284), are asserted in
 *                    tests/test_oracle_kat.py against the graph below.
 * tfo_tip5_permutation runs (2); tests/test_oracle_kat.py restates the reference's differential
 * test_mds_matrix_mul_methods_agree (:1509-1523) over all three on random and on degenerate (>= p) words. */
typedef int64_t i64;

static inline void mds_reduce_store(u128 acc, u64 *out) { /* :244-252 */
    u64 s_hi = (u64)(acc >> 64);
    u64 s_lo = (u64)acc;
    u64 add = s_hi * 0xffffffffULL;
    u64 res = s_lo + add;
    int over = res < s_lo;
    *out = over ? res + 0xffffffffULL : res;
}

static inline void mds_plain(u64 s[STATE_SIZE]) {
    u64 out[STATE_SIZE];
    for (int r = 0; r < STATE_SIZE; r++) {
        u128 acc = 0;
        for (int c = 0; c < STATE_SIZE; c++) acc += (u128)MDS_MATRIX_FIRST_COLUMN[(STATE_SIZE + r - c) % STATE_SIZE] * s[c];
        mds_reduce_store(acc, &out[r]);
    }
    memcpy(s, out, sizeof(out));
}

/* Z[i] pairs (re, im); generic over the word type so that (1) runs on i64 and (2) on wrapping u64 */
#define CYCLO_IMPL(T, SUF)                                                                                                  \
    static void complex_karatsuba_##SUF(int n, const T (*f)[2], const T (*g)[2], T (*out)[2]) { /* :941-997; out: 2n-1 */   \
        if (n == 1) { /* complex_product :999-1002 */                                                                      \
            out[0][0] = f[0][0] * g[0][0] - f[0][1] * g[0][1];                                                              \
            out[0][1] = f[0][0] * g[0][1] + f[0][1] * g[0][0];                                                              \
            return;                                                                                                         \
        }                                                                                                                   \
        const int h = n / 2;                                                                                                \
        T ff[4][2], gg[4][2], lo[7][2], hi[7][2], mid[7][2];                                                                \
        for (int i = 0; i < h; i++)                                                                                         \
            for (int k = 0; k < 2; k++) ff[i][k] = f[i][k] + f[h + i][k], gg[i][k] = g[i][k] + g[h + i][k];                 \
        complex_karatsuba_##SUF(h, f, g, lo);                                                                               \
        complex_karatsuba_##SUF(h, f + h, g + h, hi);                                                                       \
        complex_karatsuba_##SUF(h, ff, gg, mid);                                                                            \
        for (int i = 0; i < 2 * n - 1; i++) out[i][0] = out[i][1] = 0;                                                      \
        for (int i = 0; i < 2 * h - 1; i++)                                                                                 \
            for (int k = 0; k < 2; k++) {                                                                                   \
                out[i][k] += lo[i][k];                                                                                      \
                out[h + i][k] += mid[i][k] - (lo[i][k] + hi[i][k]);                                                         \
                out[2 * h + i][k] += hi[i][k];                                                                              \
            }                                                                                                               \
    }                                                                                                                       \
    /* f * g mod x^n + 1 (n = 2N): x^N acts as i on the pairs (f_j, -f_{N+j}) :871-939 */                                  \
    static void complex_negacyclomul_##SUF(int n, const T *f, const T *g, T *out) {                                        \
        const int N = n / 2;                                                                                                \
        T f0[4][2], g0[4][2], h0[7][2], h[12];                                                                              \
        for (int i = 0; i < N; i++) f0[i][0] = f[i], f0[i][1] = (T)0 - f[N + i], g0[i][0] = g[i], g0[i][1] = (T)0 - g[N + i]; \
        complex_karatsuba_##SUF(N, f0, g0, h0);                                                                             \
        for (int i = 0; i < 3 * N; i++) h[i] = 0;                                                                           \
        for (int i = 0; i < 2 * N - 1; i++) h[i] += h0[i][0], h[i + N] -= h0[i][1];                                         \
        for (int i = 0; i < 2 * N; i++) out[i] = h[i];                                                                      \
        for (int i = 2 * N; i < 3 * N - 1; i++) out[i - 2 * N] -= h[i];                                                     \
    }                                                                                                                       \
    /* f * g mod x^n - 1 :780-869.  halve: the reference's ">> 1" after every recombination (exact: the sums are even).     \
     * !halve: every halving deferred -- the hi branch is scaled by n/2 instead, the result is n x the convolution. */      \
    static void fast_cyclomul_##SUF(int n, const T *f, const T *g, T *out, int halve) {                                     \
        if (n == 1) {                                                                                                       \
            out[0] = f[0] * g[0];                                                                                           \
            return;                                                                                                         \
        }                                                                                                                   \
        const int N = n / 2;                                                                                                \
        T ff_lo[8], gg_lo[8], ff_hi[8], gg_hi[8], hh_lo[8], hh_hi[8];                                                       \
        for (int i = 0; i < N; i++) {                                                                                       \
            ff_lo[i] = f[i] + f[i + N], ff_hi[i] = f[i] - f[i + N];                                                         \
            gg_lo[i] = g[i] + g[i + N], gg_hi[i] = g[i] - g[i + N];                                                         \
        }                                                                                                                   \
        fast_cyclomul_##SUF(N, ff_lo, gg_lo, hh_lo, halve);                                                                 \
        if (N == 1) hh_hi[0] = ff_hi[0] * gg_hi[0]; /* fast_cyclomul2 :855-869 */                                          \
        else complex_negacyclomul_##SUF(N, ff_hi, gg_hi, hh_hi);                                                            \
        for (int i = 0; i < N; i++) {                                                                                       \
            if (halve) {                                                                                                    \
                out[i] = (T)((i64)(hh_lo[i] + hh_hi[i]) >> 1);                                                              \
                out[i + N] = (T)((i64)(hh_lo[i] - hh_hi[i]) >> 1);                                                          \
            } else {                                                                                                        \
                out[i] = hh_lo[i] + (T)N * hh_hi[i];                                                                        \
                out[i + N] = hh_lo[i] - (T)N * hh_hi[i];                                                                    \
            }                                                                                                               \
        }                                                                                                                   \
    }
CYCLO_IMPL(i64, s)
CYCLO_IMPL(u64, w)

static inline void mds_cyclomul(u64 s[STATE_SIZE]) { /* :753-778 */
    i64 lo[STATE_SIZE], hi[STATE_SIZE], m[STATE_SIZE], rl[STATE_SIZE], rh[STATE_SIZE];
    for (int i = 0; i < STATE_SIZE; i++) hi[i] = (i64)(s[i] >> 32), lo[i] = (i64)(s[i] & 0xffffffffULL), m[i] = MDS_MATRIX_FIRST_COLUMN[i];
    fast_cyclomul_s(STATE_SIZE, lo, m, rl, 1);
    fast_cyclomul_s(STATE_SIZE, hi, m, rh, 1);
    for (int r = 0; r < STATE_SIZE; r++) {
        u128 acc = (u128)(u64)rl[r] + ((u128)(u64)rh[r] << 32);
        u64 s_hi = (u64)(acc >> 64), s_lo = (u64)acc;
        u64 z = (s_hi << 32) - s_hi;
        u64 res = s_lo + z;
        int over = res < s_lo;
        s[r] = res + (u64)(uint32_t)(0u - (uint32_t)over);
    }
}

static inline void mds_generated(u64 s[STATE_SIZE]) { /* :210-253 over the graph of :256-506 */
    u64 lo[STATE_SIZE], hi[STATE_SIZE], m[STATE_SIZE], rl[STATE_SIZE], rh[STATE_SIZE];
    for (int i = 0; i < STATE_SIZE; i++) hi[i] = s[i] >> 32, lo[i] = s[i] & 0xffffffffULL, m[i] = (u64)MDS_MATRIX_FIRST_COLUMN[i];
    fast_cyclomul_w(STATE_SIZE, lo, m, rl, 0);  /* = 16 x the convolution, wrapping u64 */
    fast_cyclomul_w(STATE_SIZE, hi, m, rh, 0);
    for (int r = 0; r < STATE_SIZE; r++) mds_reduce_store((u128)(rl[r] >> 4) + ((u128)rh[r] << 28), &s[r]);
}

static inline void mds(u64 s[STATE_SIZE]) { mds_generated(s); }

/* test entry: one MDS layer by method 0 / 1 / 2 (see above) */
void tfo_tip5_mds(uint64_t s[16], int method) {
    if (method == 0) mds_plain(s);
    else if (method == 1) mds_cyclomul(s);
    else mds_generated(s);
}

/* test entry: the graph's fully split constants: out[0] = multiplier of (sum of the inputs) in output 0 of the deferred graph,
 * out[1] = that of the alternating sum -- the reference's node_64 / node_67 literals 524757 and 52427 (:283-284) */
void tfo_tip5_mds_graph_constants(uint64_t out[2]) {
    u64 f[STATE_SIZE], m[STATE_SIZE], r[STATE_SIZE];
    for (int i = 0; i < STATE_SIZE; i++) m[i] = (u64)MDS_MATRIX_FIRST_COLUMN[i];
    for (int i = 0; i < STATE_SIZE; i++) f[i] = 1;  /* only the x - 1 component is non-zero: every output = sum f * sum M */
    fast_cyclomul_w(STATE_SIZE, f, m, r, 0);
    out[0] = r[0] / STATE_SIZE;  /* 16 x (16 * sum M) / 16 ... the graph returns 16 x convolution = 16 * 16 * sum M / 16 */
    for (int i = 0; i < STATE_SIZE; i++) f[i] = (i & 1) ? (u64)0 - 1 : 1;  /* only the x + 1 component */
    fast_cyclomul_w(STATE_SIZE, f, m, r, 0);
    out[1] = r[0] / STATE_SIZE;
}

/* :175-181 */
static inline void tip5_round(u64 s[STATE_SIZE], int round) {
    sbox_layer(s);
    mds(s);
    for (int i = 0; i < STATE_SIZE; i++) s[i] = bfe_add(s[i], ROUND_CONSTANTS[round * STATE_SIZE + i]);
}

void tfo_tip5_permutation(uint64_t s[16]) { /* :529-533 */
    pthread_once(&g_tip5_once, tip5_init_tables);
    for (int r = 0; r < NUM_ROUNDS; r++) tip5_round(s, r);
}

/* Tip5::trace (tip5/mod.rs:538-548): the state before the permutation and after every round; s ends permuted */
void tfo_tip5_trace(uint64_t s[16], uint64_t trace[96]) {
    pthread_once(&g_tip5_once, tip5_init_tables);
    memcpy(trace, s, 16 * sizeof(u64));
    for (int i = 0; i < NUM_ROUNDS; i++) {
        tip5_round(s, i);
        memcpy(trace + 16 * (i + 1), s, 16 * sizeof(u64));
    }
}
/* tip5/naive.rs:26-76 : x^7 by mod_pow, MDS as a field matrix product */
void tfo_tip5_permutation_naive(uint64_t s[16]) {
    pthread_once(&g_tip5_once, tip5_init_tables);
    for (int round = 0; round < NUM_ROUNDS; round++) {
        for (int i = 0; i < NUM_SPLIT_AND_LOOKUP; i++) s[i] = split_and_lookup(s[i]);
        for (int i = NUM_SPLIT_AND_LOOKUP; i < STATE_SIZE; i++) s[i] = bfe_mod_pow(s[i], 7);
        u64 ns[STATE_SIZE];
        for (int r = 0; r < STATE_SIZE; r++) {
            u64 acc = 0;
            for (int c = 0; c < STATE_SIZE; c++) {
                u64 m = bfe_new(MDS_MATRIX_FIRST_COLUMN[(STATE_SIZE + r - c) % STATE_SIZE]);
                acc = bfe_add(acc, bfe_mul(m, s[c]));
            }
            ns[r] = acc;
        }
        for (int i = 0; i < STATE_SIZE; i++) s[i] = bfe_add(ns[i], ROUND_CONSTANTS[round * STATE_SIZE + i]);
    }
}

/* :559-569 ; Tip5::new(FixedLength) :511-526 sets capacity to ONE */
void tfo_tip5_hash_10(const uint64_t in[10], uint64_t out[5]) {
    u64 s[STATE_SIZE];
    memcpy(s, in, RATE * sizeof(u64));
    for (int i = RATE; i < STATE_SIZE; i++) s[i] = BFE_ONE;
    tfo_tip5_permutation(s);
    memcpy(out, s, DIGEST_LEN * sizeof(u64));
}

/* :577-586 */
void tfo_tip5_hash_pair(const uint64_t l[5], const uint64_t r[5], uint64_t out[5]) {
    u64 in[RATE];
    memcpy(in, l, 5 * sizeof(u64));
    memcpy(in + 5, r, 5 * sizeof(u64));
    tfo_tip5_hash_10(in, out);
}

/* :684-691 : overwrite-mode absorb */
void tfo_tip5_absorb(uint64_t s[16], const uint64_t in[10]) {
    memcpy(s, in, RATE * sizeof(u64));
    tfo_tip5_permutation(s);
}

/* :617-623 with sponge.rs:41-55 */
void tfo_tip5_hash_varlen(const uint64_t *in, size_t len, uint64_t out[5]) {
    u64 s[STATE_SIZE] = {0}; /* init() = new(VariableLength), :680-682 */
    size_t full = len / RATE;
    for (size_t c = 0; c < full; c++) tfo_tip5_absorb(s, in + c * RATE);
    size_t rem = len - full * RATE;
    u64 last[RATE] = {0};
    memcpy(last, in + full * RATE, rem * sizeof(u64));
    last[rem] = BFE_ONE;
    tfo_tip5_absorb(s, last);
    memcpy(out, s, DIGEST_LEN * sizeof(u64));
}

void tfo_tip5_hash_pairs(const uint64_t *in, uint64_t *out, size_t count) {
    for (size_t i = 0; i < count; i++) tfo_tip5_hash_10(in + 10 * i, out + 5 * i);
}

void tfo_tip5_hash_varlen_rows(const uint64_t *rows, size_t row_len, size_t n_rows, uint64_t *out) {
    for (size_t i = 0; i < n_rows; i++) tfo_tip5_hash_varlen(rows + i * row_len, row_len, out + 5 * i);
}

/* the same, the rows dealt out to `threads` threads in blocks of 256 (what a rayon caller of hash_varlen does: the rows are
 * independent); used for the all-cores CPU leg of bench.py's commitment pipeline */
typedef struct {
    const u64 *rows;
    u64 *out;
    size_t row_len, n_rows, next;
    pthread_mutex_t *lock;
} rows_job_t;

static void *rows_worker(void *arg) {
    rows_job_t *job = (rows_job_t *)arg;
    for (;;) {
        pthread_mutex_lock(job->lock);
        size_t lo = job->next;
        job->next += 256;
        pthread_mutex_unlock(job->lock);
        if (lo >= job->n_rows) break;
        size_t hi = lo + 256 < job->n_rows ? lo + 256 : job->n_rows;
        for (size_t i = lo; i < hi; i++) tfo_tip5_hash_varlen(job->rows + i * job->row_len, job->row_len, job->out + 5 * i);
    }
    return NULL;
}

void tfo_tip5_hash_varlen_rows_par(const uint64_t *rows, size_t row_len, size_t n_rows, uint64_t *out, int threads) {
    pthread_once(&g_tip5_once, tip5_init_tables); /* before the fan-out */
    pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
    rows_job_t job = {rows, out, row_len, n_rows, 0, &lock};
    fan_out(rows_worker, &job, threads);
}

/* ------------------------------------------------------------------ MerkleTree */

/* merkle_tree.rs:393-429 */
static int merkle_init_nodes(const u64 *leaves, size_t n, u64 *nodes) {
    if (n == 0) return 1;          /* TooFewLeafs */
    if (n & (n - 1)) return 2;     /* IncorrectNumberOfLeafs */
    memset(nodes, 0, n * 5 * sizeof(u64)); /* nodes[0..n) = ALL_ZERO (nodes[0] stays the dummy) */
    memcpy(nodes + 5 * n, leaves, n * 5 * sizeof(u64)); /* :426 */
    return 0;
}

/* :216-222 */
static void merkle_fill_sequential(u64 *nodes, size_t num_remaining) {
    for (size_t i = num_remaining; i-- > 1;) tfo_tip5_hash_pair(nodes + 5 * (2 * i), nodes + 5 * (2 * i + 1), nodes + 5 * i);
}

int tfo_merkle_build(const uint64_t *leaves, size_t n, uint64_t *nodes) { /* :149-153 */
    int rc = merkle_init_nodes(leaves, n, nodes);
    if (rc) return rc;
    merkle_fill_sequential(nodes, n);
    return 0;
}

typedef struct {
    u64 *nodes;
    size_t num_remaining; /* nodes[num_remaining .. 2*num_remaining) is the bottom layer */
    size_t num_trees, tree;
} subtree_job_t;

/* One subtree of merkle_tree.rs:247-275 / :190-200: layer l of subtree t holds
 * nodes[num_trees*2^l + t*2^l .. + 2^l), for l = 0..subtree_height. */
static void *subtree_worker(void *arg) {
    subtree_job_t *j = (subtree_job_t *)arg;
    size_t total_h = 0, tt = j->num_trees, th = 0;
    for (size_t v = j->num_remaining; v > 1; v >>= 1) total_h++;
    for (size_t v = tt; v > 1; v >>= 1) th++;
    size_t sub_h = total_h - th;
    for (size_t l = sub_h; l-- > 0;) {
        size_t base = (j->num_trees << l) + (j->tree << l);
        for (size_t i = 0; i < ((size_t)1 << l); i++) {
            size_t node = base + i;
            tfo_tip5_hash_pair(j->nodes + 5 * (2 * node), j->nodes + 5 * (2 * node + 1), j->nodes + 5 * node);
        }
    }
    return NULL;
}

static size_t prev_pow2(size_t v) {
    size_t p = 1;
    while (p * 2 <= v) p *= 2;
    return p;
}

/* :165-212 */
int tfo_merkle_build_par(const uint64_t *leaves, size_t n, uint64_t *nodes, int threads, size_t cutoff) {
    int rc = merkle_init_nodes(leaves, n, nodes);
    if (rc) return rc;
    if (cutoff < 2) cutoff = 2; /* config.rs MINIMUM */
    size_t num_remaining = n;
    size_t num_threads = prev_pow2(threads < 1 ? 1 : (size_t)threads); /* :376-388 */
    while (num_remaining >= cutoff) {
        while (num_threads > num_remaining / 2) num_threads /= 2; /* :181-183 */
        subtree_job_t *jobs = (subtree_job_t *)malloc(sizeof(subtree_job_t) * num_threads);
        pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * num_threads);
        char *made = (char *)calloc(num_threads, 1);
        /* every subtree is a job of its own (it owns its slice of the node array): a thread that cannot be created -- a cgroup-limited
         * host, or no memory for the bookkeeping -- is not joined, and its subtree is built inline by the caller instead */
        for (size_t t = 0; t < num_threads; t++) {
            if (jobs) jobs[t] = (subtree_job_t){nodes, num_remaining, num_threads, t};
            if (jobs && tid && made && num_threads > 1 && pthread_create(&tid[t], NULL, subtree_worker, &jobs[t]) == 0) {
                made[t] = 1;
            } else {
                subtree_job_t mine = {nodes, num_remaining, num_threads, t};
                subtree_worker(&mine);
            }
        }
        for (size_t t = 0; t < num_threads; t++)
            if (made && made[t]) pthread_join(tid[t], NULL);
        free(jobs);
        free(tid);
        free(made);
        size_t cur_h = 0, th = 0;
        for (size_t v = num_remaining; v > 1; v >>= 1) cur_h++;
        for (size_t v = num_threads; v > 1; v >>= 1) th++;
        num_remaining >>= (cur_h - th); /* :205-207 */
    }
    merkle_fill_sequential(nodes, num_remaining);
    return 0;
}

/* :299-309 via MmrAccumulator::peaks_from_leafs, mmr/mmr_accumulator.rs:96-115 */
int tfo_merkle_frugal_root(const uint64_t *leaves, size_t n, uint64_t root[5]) {
    if (n == 0) return 1;
    u64 peaks[64][5];
    int np = 0;
    for (size_t pair = 0; pair < n / 2; pair++) {
        size_t diagonal_idx = pair + 1;
        u64 right[5];
        tfo_tip5_hash_pair(leaves + 5 * (2 * pair), leaves + 5 * (2 * pair + 1), right);
        int tz = __builtin_ctzll(diagonal_idx);
        for (int k = 0; k < tz; k++) {
            np--;
            tfo_tip5_hash_pair(peaks[np], right, right);
        }
        memcpy(peaks[np++], right, sizeof(right));
    }
    if (n % 2 == 1) memcpy(peaks[np++], leaves + 5 * (n - 1), 5 * sizeof(u64));
    if (np != 1) return 2; /* IncorrectNumberOfLeafs */
    memcpy(root, peaks[0], 5 * sizeof(u64));
    return 0;
}

/* ------------------------------------------------------------------ "next" rows */

/* schoolbook product: ground truth for fast_multiply (polynomial.rs:900-932 computes the same polynomial) */
void tfo_poly_mul_naive(const uint64_t *a, size_t na, const uint64_t *b, size_t nb, int width, uint64_t *out) {
    if (na == 0 || nb == 0) return;
    size_t n_out = na + nb - 1;
    memset(out, 0, n_out * (size_t)width * sizeof(u64));
    for (size_t i = 0; i < na; i++) {
        for (size_t j = 0; j < nb; j++) {
            if (width == 1) {
                out[i + j] = bfe_add(out[i + j], bfe_mul(a[i], b[j]));
            } else {
                u64 prod[3];
                tfo_xfe_mul(a + 3 * i, b + 3 * j, prod);
                for (int c = 0; c < 3; c++) out[3 * (i + j) + c] = bfe_add(out[3 * (i + j) + c], prod[c]);
            }
        }
    }
}

/* Polynomial::fast_multiply restated literally (polynomial.rs:900-932): pad to order, ntt, pointwise, intt, truncate */
int tfo_poly_mul_fast(const uint64_t *a, size_t na, const uint64_t *b, size_t nb, int width, uint64_t *out) {
    if (na == 0 || nb == 0) return 0;
    size_t n_out = na + nb - 1, order = 1;
    while (order < n_out) order <<= 1;
    u64 *l = (u64 *)calloc(order * (size_t)width, sizeof(u64));
    u64 *r = (u64 *)calloc(order * (size_t)width, sizeof(u64));
    memcpy(l, a, na * (size_t)width * sizeof(u64));
    memcpy(r, b, nb * (size_t)width * sizeof(u64));
    int rc = ntt_any(l, order, width, 0);
    if (!rc) rc = ntt_any(r, order, width, 0);
    if (!rc) {
        for (size_t i = 0; i < order; i++) {
            if (width == 1)
                l[i] = bfe_mul(l[i], r[i]);
            else
                tfo_xfe_mul(l + 3 * i, r + 3 * i, l + 3 * i);
        }
        rc = ntt_any(l, order, width, 1);
    }
    if (!rc) memcpy(out, l, n_out * (size_t)width * sizeof(u64));
    free(l);
    free(r);
    return rc;
}

static int cmp_u64_desc(const void *x, const void *y) {
    u64 a = *(const u64 *)x, b = *(const u64 *)y;
    return a < b ? 1 : (a > b ? -1 : 0);
}
static int contains_sorted_desc(const u64 *v, size_t n, u64 key) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (v[mid] == key) return 1;
        if (v[mid] > key) lo = mid + 1; else hi = mid;
    }
    return 0;
}

/* MerkleTree::authentication_structure_node_indices (merkle_tree.rs:449-504).
 * rc: 2 IncorrectNumberOfLeafs, 11 LeafIndexInvalid.  out gets *count indices, descending. */
int tfo_auth_structure_indices(size_t num_leafs, const uint64_t *leaf_indices, size_t k, uint64_t *out, size_t cap, size_t *count) {
    if (num_leafs == 0 || ((num_leafs - 1) & num_leafs)) return 2;
    size_t height = 0;
    for (size_t v = num_leafs; v > 1; v >>= 1) height++;
    size_t maxn = k * (height + 1) + 1;
    u64 *needed = (u64 *)malloc(maxn * sizeof(u64)), *comp = (u64 *)malloc(maxn * sizeof(u64));
    size_t nn = 0, nc = 0;
    for (size_t i = 0; i < k; i++) {
        if (leaf_indices[i] >= num_leafs) { free(needed); free(comp); return 11; }
        u64 node = leaf_indices[i] + num_leafs;
        while (node > 1) {
            comp[nc++] = node;
            needed[nn++] = node ^ 1;
            node /= 2;
        }
    }
    qsort(needed, nn, sizeof(u64), cmp_u64_desc);
    qsort(comp, nc, sizeof(u64), cmp_u64_desc);
    size_t w = 0;
    for (size_t i = 0; i < nn; i++) {
        if (i && needed[i] == needed[i - 1]) continue;          /* set semantics */
        if (contains_sorted_desc(comp, nc, needed[i])) continue; /* difference */
        if (w < cap) out[w] = needed[i];
        w++;
    }
    *count = w;
    free(needed);
    free(comp);
    return 0;
}

/* Horner at an XFieldElement point (Polynomial::evaluate with Ind = Eval = XFieldElement, polynomial.rs:309-325) */
void tfo_poly_eval_xfe_point(const uint64_t *c, size_t n_coeffs, const uint64_t point[3], uint64_t out[3]) {
    u64 acc[3] = {0, 0, 0};
    for (size_t i = n_coeffs; i-- > 0;) {
        u64 t[3];
        tfo_xfe_mul(acc, point, t);
        for (int k = 0; k < 3; k++) acc[k] = bfe_add(t[k], c[3 * i + k]);
    }
    memcpy(out, acc, sizeof(acc));
}

/* ------------------------------------------------------------------ zerofier / interpolation (math/polynomial.rs) */

static void fe_mul_w(const u64 *a, const u64 *b, u64 *out, int w) {
    if (w == 1) out[0] = bfe_mul(a[0], b[0]);
    else tfo_xfe_mul(a, b, out);
}
static void fe_add_w(const u64 *a, const u64 *b, u64 *out, int w) {
    for (int k = 0; k < w; k++) out[k] = bfe_add(a[k], b[k]);
}
static void fe_sub_w(const u64 *a, const u64 *b, u64 *out, int w) {
    for (int k = 0; k < w; k++) out[k] = bfe_sub(a[k], b[k]);
}

/* XFieldElement::inverse (x_field_element.rs:371-379) runs xgcd(self, x^3 - x + 1) over Polynomial<BFieldElement>; the inverse
 * is unique, and here it is found as the solution b of the linear system (a * x^j)_j b = 1: the columns come from the
 * reference's own product (tfo_xfe_mul), the system is solved by Gauss-Jordan elimination over BFieldElement.
 * Returns 1 for a = 0 ("Cannot invert the zero element in the extension field."). */
int tfo_xfe_inverse(const u64 a[3], u64 out[3]) {
    u64 m[3][4];
    for (int j = 0; j < 3; j++) {
        u64 e[3] = {0, 0, 0}, col[3];
        e[j] = bfe_new(1);
        tfo_xfe_mul(a, e, col);
        for (int r = 0; r < 3; r++) m[r][j] = col[r];
    }
    m[0][3] = bfe_new(1);
    m[1][3] = 0;
    m[2][3] = 0;
    for (int c = 0; c < 3; c++) {
        int piv = -1;
        for (int r = c; r < 3; r++)
            if (m[r][c]) { piv = r; break; }
        if (piv < 0) return 1;
        for (int k = 0; k < 4; k++) { u64 t = m[c][k]; m[c][k] = m[piv][k]; m[piv][k] = t; }
        u64 inv = bfe_inverse(m[c][c]);
        for (int k = 0; k < 4; k++) m[c][k] = bfe_mul(m[c][k], inv);
        for (int r = 0; r < 3; r++) {
            if (r == c || !m[r][c]) continue;
            u64 f = m[r][c];
            for (int k = 0; k < 4; k++) m[r][k] = bfe_sub(m[r][k], bfe_mul(f, m[c][k]));
        }
    }
    for (int r = 0; r < 3; r++) out[r] = m[r][3];
    return 0;
}

/* Polynomial::smart_zerofier (polynomial.rs:1462-1475): prod (x - root), n + 1 coefficients, one root at a time */
void tfo_poly_zerofier(const uint64_t *roots, size_t n, int width, uint64_t *out) {
    const size_t w = (size_t)width;
    memset(out, 0, (n + 1) * w * sizeof(u64));
    out[0] = bfe_new(1);
    size_t num_coeffs = 1;
    for (size_t i = 0; i < n; i++) {
        const u64 *root = roots + i * w;
        for (size_t k = num_coeffs; k >= 1; k--) { /* zerofier[k] = zerofier[k - 1] - root * zerofier[k] */
            u64 t[3];
            fe_mul_w(root, out + k * w, t, width);
            fe_sub_w(out + (k - 1) * w, t, out + k * w, width);
        }
        u64 t[3], z[3] = {0, 0, 0};
        fe_mul_w(root, out, t, width); /* zerofier[0] = -root * zerofier[0] */
        fe_sub_w(z, t, out, width);
        num_coeffs++;
    }
}

/* Polynomial::lagrange_interpolate (polynomial.rs:1565-1606): n coefficients; returns 1 where the reference panics dividing by
 * a zero summand_eval (a repeated domain point), 2 for an empty domain */
int tfo_poly_lagrange_interpolate(const uint64_t *domain, const uint64_t *values, size_t n, int width, uint64_t *out) {
    if (n == 0) return 2;
    const size_t w = (size_t)width;
    u64 *zerofier = (u64 *)malloc((n + 1) * w * sizeof(u64));
    u64 *summand = (u64 *)malloc(n * w * sizeof(u64));
    tfo_poly_zerofier(domain, n, width, zerofier);
    memset(out, 0, n * w * sizeof(u64));
    int rc = 0;
    for (size_t i = 0; i < n && !rc; i++) {
        const u64 *x = domain + i * w;
        u64 lead[3], supp[3], eval[3] = {0, 0, 0}, t[3];
        memcpy(lead, zerofier + n * w, w * sizeof(u64));
        memcpy(supp, zerofier + (n - 1) * w, w * sizeof(u64));
        for (size_t j = n - 1; j >= 1; j--) {
            memcpy(summand + j * w, lead, w * sizeof(u64));
            fe_mul_w(eval, x, t, width);
            fe_add_w(t, lead, eval, width);
            fe_mul_w(lead, x, t, width);
            fe_add_w(supp, t, lead, width);
            memcpy(supp, zerofier + (j - 1) * w, w * sizeof(u64));
        }
        memcpy(summand, lead, w * sizeof(u64));
        fe_mul_w(eval, x, t, width);
        fe_add_w(t, lead, eval, width);
        u64 inv[3] = {0, 0, 0}, corrected[3];
        if (width == 1) {
            if (!eval[0]) rc = 1;
            else inv[0] = bfe_inverse(eval[0]);
        } else if (tfo_xfe_inverse(eval, inv)) {
            rc = 1;
        }
        if (rc) break;
        fe_mul_w(values + i * w, inv, corrected, width);
        for (size_t j = 0; j < n; j++) {
            fe_mul_w(corrected, summand + j * w, t, width);
            fe_add_w(out + j * w, t, out + j * w, width);
        }
    }
    free(zerofier);
    free(summand);
    return rc;
}

/* Polynomial::scale with an XFieldElement alpha on XFieldElement coefficients (polynomial.rs:760-773): c_i <- c_i * alpha^i,
 * the power chain carried sequentially as the reference does */
void tfo_poly_scale_xfe(uint64_t *c, size_t n_coeffs, const uint64_t alpha[3]) {
    u64 pw[3] = {bfe_new(1), 0, 0}, t[3];
    for (size_t i = 0; i < n_coeffs; i++) {
        tfo_xfe_mul(c + 3 * i, pw, t);
        memcpy(c + 3 * i, t, sizeof(t));
        tfo_xfe_mul(pw, alpha, t);
        memcpy(pw, t, sizeof(t));
    }
}

/* barycentric_evaluate (polynomial.rs:2609-2637): the interpolant of `codeword` (on the subgroup of order n, natural order) at
 * an indeterminate that is not in the subgroup:  sum_i c_i d_i / (x - d_i)  over  sum_i d_i / (x - d_i),  d_i = w_n^i.
 * The indeterminate is given as an XFieldElement (a BFieldElement is its lift [x, 0, 0]: the arithmetic then never leaves the
 * base field and limb 0 is the reference's BFieldElement result); width = 1 / 3 is the codeword's field.
 * Returns 1 for a length that is not a power of two <= 2^32 (primitive_root_of_unity(..).unwrap() panics), 2 where
 * batch_inversion / inverse panic on zero (the indeterminate is in the subgroup, or n = 0). */
int tfo_barycentric_evaluate(const uint64_t *codeword, size_t n, int width, const uint64_t x[3], uint64_t out[3]) {
    if (n & (n - 1)) return 1;
    if (n == 0) return 2; /* denominator = 0: inverse() panics */
    const u64 gen = tfo_bfe_primitive_root(n);
    u64 d = bfe_new(1), den[3] = {0, 0, 0}, num[3] = {0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        u64 shift[3] = {bfe_sub(x[0], d), x[1], x[2]}, inv[3], w[3], t[3];
        if (tfo_xfe_inverse(shift, inv)) return 2;
        tfo_xfe_mul_bfe(inv, d, w); /* domain_over_domain_shift */
        tfo_xfe_add(den, w, t);
        memcpy(den, t, sizeof(t));
        if (width == 1) tfo_xfe_mul_bfe(w, codeword[i], t);
        else tfo_xfe_mul(codeword + 3 * i, w, t);
        u64 acc[3];
        tfo_xfe_add(num, t, acc);
        memcpy(num, acc, sizeof(acc));
        d = bfe_mul(d, gen);
    }
    u64 dinv[3];
    if (tfo_xfe_inverse(den, dinv)) return 2;
    tfo_xfe_mul(num, dinv, out);
    return 0;
}

/* ------------------------------------------------------------------ division (math/polynomial.rs) */

/* Polynomial::naive_divide (polynomial.rs:552-600) over BFieldElement: quotient (max(na - nb + 1, 0) coefficients, untrimmed) and
 * remainder (na coefficients, the top ones zeroed as the long division consumes them).  na / nb count NORMALISED coefficients
 * (non-zero leading coefficient), as Polynomial::degree sees them.  Returns 1 for a zero divisor ("divisor should be non-zero"). */
int tfo_poly_naive_divide_bfe(const uint64_t *a, size_t na, const uint64_t *b, size_t nb, uint64_t *quot, uint64_t *rem) {
    if (nb == 0) return 1;
    memcpy(rem, a, na * sizeof(u64));
    if (na < nb) return 0;
    const u64 lc_inv = bfe_inverse(b[nb - 1]);
    const size_t qdeg = na - nb;
    size_t len = na; /* remainder_coefficients.len() */
    for (size_t step = 0; step <= qdeg; step++) {
        const u64 lc = rem[len - 1];
        rem[len - 1] = 0;
        len--; /* pop */
        const u64 qc = bfe_mul(lc, lc_inv);
        quot[qdeg - step] = qc;
        if (!qc) continue;
        const size_t rdeg = len ? len - 1 : 0;
        for (size_t i = 0; i + 1 < nb; i++) /* divisor back to front, leading coefficient skipped */
            rem[rdeg - i] = bfe_sub(rem[rdeg - i], bfe_mul(qc, b[nb - 2 - i]));
    }
    return 0;
}

/* Polynomial::<BFieldElement>::clean_divide (polynomial.rs:2358-2411).  cutoff = CLEAN_DIVIDE_CUTOFF_THRESHOLD (:2339: 1 << 9,
 * 0 under cfg(test)): divisors of lower degree go through naive_divide.  na / nb count normalised coefficients.
 * out receives na - nb + 1 coefficients.  Returns 0, or the reference's panics: 1 zero divisor (naive_divide :556-559),
 * 2 a zero among the divisor's values on the coset (batch_inversion, traits.rs:106), 3 division not clean (:2374
 * assert / :2410 unlift().unwrap(); also a quotient of higher degree than deg a - deg b, which the reference would return as a
 * wrong result), 4 dividend of lower degree than the divisor and not zero. */
int tfo_poly_clean_divide_bfe(const uint64_t *a, size_t na, const uint64_t *b, size_t nb, size_t cutoff, uint64_t *out) {
    if (nb == 0) return 1;
    if (na < nb) return na ? 4 : 0;
    if (nb - 1 < cutoff) { /* divisor.degree() < threshold */
        u64 *rem = (u64 *)malloc((na ? na : 1) * sizeof(u64));
        int rc = tfo_poly_naive_divide_bfe(a, na, b, nb, out, rem);
        free(rem); /* debug_assert!(remainder.is_zero()) only */
        return rc;
    }
    /* :2368-2378 one factor x off both when the divisor's constant term is zero */
    if (b[0] == 0) {
        if (a[0] != 0) return 3;
        a++, b++, na--, nb--;
    }
    const u64 X[3] = {0, bfe_new(1), 0};            /* offset = XFieldElement::from([0, 1, 0]) :2383 */
    size_t order = 1;
    while (order < na) order <<= 1;                 /* (dividend.degree() + 1).next_power_of_two() :2388-2389 */
    u64 *da = (u64 *)calloc(3 * order, sizeof(u64)), *db = (u64 *)calloc(3 * order, sizeof(u64));
    u64 pw[3] = {bfe_new(1), 0, 0}, t[3];
    for (size_t i = 0; i < order; i++) {            /* scale(offset) :2384-2385: c_i * offset^i */
        if (i < na) tfo_xfe_mul_bfe(pw, a[i], da + 3 * i);
        if (i < nb) tfo_xfe_mul_bfe(pw, b[i], db + 3 * i);
        tfo_xfe_mul(pw, X, t);
        memcpy(pw, t, sizeof(t));
    }
    int rc = 0;
    ntt_any(da, order, 3, 0);
    ntt_any(db, order, 3, 0);
    for (size_t i = 0; i < order && !rc; i++) {     /* batch_inversion + pointwise product :2397-2402 */
        u64 inv[3];
        if (tfo_xfe_inverse(db + 3 * i, inv)) { rc = 2; break; }
        tfo_xfe_mul(da + 3 * i, inv, t);
        memcpy(da + 3 * i, t, sizeof(t));
    }
    if (!rc) {
        ntt_any(da, order, 3, 1);
        u64 Xinv[3];
        tfo_xfe_inverse(X, Xinv);
        u64 q[3] = {bfe_new(1), 0, 0};
        const size_t nq = na - nb + 1;
        for (size_t i = 0; i < order; i++) {        /* scale(offset.inverse()) and unlift :2408-2410 */
            tfo_xfe_mul(da + 3 * i, q, t);
            if (t[1] || t[2] || (i >= nq && t[0])) { rc = 3; break; }
            if (i < nq) out[i] = t[0];
            u64 n2[3];
            tfo_xfe_mul(q, Xinv, n2);
            memcpy(q, n2, sizeof(n2));
        }
    }
    free(da);
    free(db);
    return rc;
}

/* ------------------------------------------------------------------ helpers */

uint64_t tfo_splitmix64(uint64_t *state) {
    u64 z = (*state += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

/* element i = new(mix(seed ^ i) mod p), counter-based so any slice can be regenerated */
void tfo_fill_random_from(uint64_t *out, size_t count, uint64_t seed, uint64_t first_index) {
    for (size_t i = 0; i < count; i++) {
        u64 st = seed ^ (first_index + (u64)i);
        u64 v = tfo_splitmix64(&st);
        out[i] = bfe_new(v % P);
    }
}

void tfo_fill_random(uint64_t *out, size_t count, uint64_t seed) { tfo_fill_random_from(out, count, seed, 0); }

void tfo_digest_to_hex(const uint64_t d[5], char out[81]) {
    static const char *hx = "0123456789abcdef";
    for (int i = 0; i < 5; i++) {
        u64 v = bfe_value(d[i]);
        for (int b = 0; b < 8; b++) {
            unsigned byte = (unsigned)((v >> (8 * b)) & 0xff);
            out[16 * i + 2 * b] = hx[byte >> 4];
            out[16 * i + 2 * b + 1] = hx[byte & 15];
        }
    }
    out[80] = 0;
}
