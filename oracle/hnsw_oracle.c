/*
 * oracle/hnsw_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the HNSWIndex.Net v1.6.0 hot path: the three float32
 * metrics in the exact lane/rounding order of the reference's AVX(+FMA) branch, and
 * the host-side traversal/link logic that consumes them (heaps, visited list, level
 * RNG, relative-neighbour pruning, Add, KnnQuery).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (hnswindex.net_amd/) never links, imports or calls it.
 *
 * PARITY STATUS
 *   - metric arithmetic: follows the reference source line by line (citations below);
 *     pinned by the reference's own tolerances (MetricsTests.cs:7-92, 1e-6 vs scalar;
 *     bindings/__tests__/metric_test.py:34-96, atol 1e-5 vs float64) in
 *     tests/test_oracle_metrics.py.
 *   - System.Random(seed): restated from the public .NET runtime algorithm (Knuth
 *     subtractive generator, "Net5CompatSeedImpl"); pinned by publicly known outputs
 *     (new Random(0).Next()==1559595546, new Random(42).Next()==1434747710,
 *     new Random(1).Next()==534011718) in tests/test_oracle_dotnet.py.  The reference
 *     itself holds no level-sequence fixture.
 *   - Span.Sort (introsort) tie order, and therefore bit-exact neighbour ids on inputs
 *     with equal distances: PARITY UNPINNED -- restated from knowledge of the .NET BCL;
 *     the reference cannot be compiled or run here (no dotnet toolchain) and ships no
 *     golden vectors.  See DESIGN.md "Oracle".
 *
 * All citations are relative to /root/reference/.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#define ORC_HAVE_AVX2 1
#else
#define ORC_HAVE_AVX2 0
#endif

#define ORC_API __attribute__((visibility("default")))

enum { ORC_SQ_EUCLID = 0, ORC_COSINE = 1, ORC_UCOSINE = 2, ORC_SQ_EUCLID_I8 = 3 };

typedef struct { int id; float dist; } nd_t; /* src/HNSWIndex/NodeDistance.cs:5-14 */

/* ------------------------------------------------------------------------------------
 * Metrics, "spec" form: eight scalar partial sums standing for the eight AVX lanes.
 * Compiled with -ffp-contract=off so that a*b+c is never fused unless fmaf() is written.
 * ---------------------------------------------------------------------------------- */

/* src/HNSWIndex/Metrics/EuclideanMetric.cs:19-60 (AVX+FMA branch). */
static float sq_euclid_spec(const float *a, const float *b, int n)
{
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int stop = n & ~7; /* :23 */
    int i = 0;
    for (; i < stop; i += 8) /* :25-43, one accumulator, lane j <- elements j, j+8, ... */
        for (int j = 0; j < 8; j++) {
            float d = a[i + j] - b[i + j]; /* Avx.Subtract :29 */
            acc[j] = fmaf(d, d, acc[j]);   /* Fma.MultiplyAdd :30 */
        }
    /* :45-50  lower+upper, hadd, hadd  => ((p0+p4)+(p1+p5)) + ((p2+p6)+(p3+p7)) */
    float t0 = acc[0] + acc[4], t1 = acc[1] + acc[5], t2 = acc[2] + acc[6], t3 = acc[3] + acc[7];
    float s = (t0 + t1) + (t2 + t3);
    for (; i < n; i++) { /* :53-57 scalar tail: separate multiply and add */
        float d = a[i] - b[i];
        float m = d * d;
        s = s + m;
    }
    return s;
}

/* CosineMetric.HorizontalSum256/128, src/HNSWIndex/Metrics/CosineMetric.cs:145-171:
 * u_j = p_j + p_{j+4}; movehl add; shuffle-0x55 add  => (u0+u2) + (u1+u3). */
static float hsum_cos(const float *p)
{
    float u0 = p[0] + p[4], u1 = p[1] + p[5], u2 = p[2] + p[6], u3 = p[3] + p[7];
    return (u0 + u2) + (u1 + u3);
}

/* src/HNSWIndex/Metrics/CosineMetric.cs:95-142 (UnitCompute, AVX branch): mul then add. */
static float ucosine_spec(const float *a, const float *b, int n)
{
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int stop = n - 8 + 1; /* :108 */
    int i = 0;
    for (; i < stop; i += 8)
        for (int j = 0; j < 8; j++) {
            float p = a[i + j] * b[i + j]; /* Avx.Multiply :114 */
            acc[j] = acc[j] + p;           /* Avx.Add :115 */
        }
    float dot = hsum_cos(acc); /* :117 */
    for (; i < n; i++) {       /* :135-138 */
        float p = a[i] * b[i];
        dot = dot + p;
    }
    return 1.0f - dot; /* :141 */
}

/* src/HNSWIndex/Metrics/CosineMetric.cs:10-92 (Compute, AVX branch). */
static float cosine_spec(const float *a, const float *b, int n)
{
    float dA[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nA[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nB[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int stop = n - 8 + 1; /* :30 */
    int i = 0;
    for (; i < stop; i += 8)
        for (int j = 0; j < 8; j++) {
            float va = a[i + j], vb = b[i + j];
            float p = va * vb;  dA[j] = dA[j] + p; /* :37-38 */
            float sa = va * va; nA[j] = nA[j] + sa; /* :40-41 */
            float sb = vb * vb; nB[j] = nB[j] + sb; /* :43-44 */
        }
    float dot = hsum_cos(dA), na = hsum_cos(nA), nb = hsum_cos(nB); /* :46-48 */
    for (; i < n; i++) { /* :78-85 */
        float va = a[i], vb = b[i];
        float p = va * vb;  dot = dot + p;
        float sa = va * va; na = na + sa;
        float sb = vb * vb; nb = nb + sb;
    }
    float denom = (float)(sqrt((double)na) * sqrt((double)nb)); /* :88 */
    if (denom < 1e-30f) return 1.0f;                            /* :89-90 */
    return 1.0f - dot / denom;                                  /* :91 */
}

#if ORC_HAVE_AVX2
/* The same three functions with the x86 instructions the reference's intrinsics map to
 * 1:1 (System.Runtime.Intrinsics.X86 -> <immintrin.h>).  Must agree bit for bit with the
 * spec forms (tests/test_oracle_metrics.py); used for the cpu_baseline timing. */
static float sq_euclid_avx(const float *a, const float *b, int n)
{
    __m256 acc = _mm256_setzero_ps();
    int stop = n & ~7, i = 0;
    for (; i + 16 <= stop; i += 16) { /* EuclideanMetric.cs:25-36 */
        __m256 d0 = _mm256_sub_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i));
        acc = _mm256_fmadd_ps(d0, d0, acc);
        __m256 d1 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 8), _mm256_loadu_ps(b + i + 8));
        acc = _mm256_fmadd_ps(d1, d1, acc);
    }
    for (; i < stop; i += 8) { /* :37-43 */
        __m256 d0 = _mm256_sub_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i));
        acc = _mm256_fmadd_ps(d0, d0, acc);
    }
    __m128 s = _mm_add_ps(_mm256_castps256_ps128(acc), _mm256_extractf128_ps(acc, 1)); /* :45-47 */
    s = _mm_hadd_ps(s, s); /* :48 */
    s = _mm_hadd_ps(s, s); /* :49 */
    float r = _mm_cvtss_f32(s);
    for (; i < n; i++) {
        float d = a[i] - b[i];
        float m = d * d;
        r = r + m;
    }
    return r;
}

static inline float hsum_cos_avx(__m256 acc)
{
    __m128 s = _mm_add_ps(_mm256_extractf128_ps(acc, 0), _mm256_extractf128_ps(acc, 1)); /* CosineMetric.cs:147-150 */
    __m128 t = _mm_add_ps(s, _mm_movehl_ps(s, s));                                        /* :158 */
    __m128 t2 = _mm_add_ps(t, _mm_shuffle_ps(t, t, 0x55));                                /* :159 */
    return _mm_cvtss_f32(t2);
}

static float ucosine_avx(const float *a, const float *b, int n)
{
    __m256 acc = _mm256_setzero_ps();
    int stop = n - 8 + 1, i = 0;
    for (; i < stop; i += 8)
        acc = _mm256_add_ps(acc, _mm256_mul_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i)));
    float dot = hsum_cos_avx(acc);
    for (; i < n; i++) {
        float p = a[i] * b[i];
        dot = dot + p;
    }
    return 1.0f - dot;
}

static float cosine_avx(const float *a, const float *b, int n)
{
    __m256 d = _mm256_setzero_ps(), na = _mm256_setzero_ps(), nb = _mm256_setzero_ps();
    int stop = n - 8 + 1, i = 0;
    for (; i < stop; i += 8) {
        __m256 va = _mm256_loadu_ps(a + i), vb = _mm256_loadu_ps(b + i);
        d = _mm256_add_ps(d, _mm256_mul_ps(va, vb));
        na = _mm256_add_ps(na, _mm256_mul_ps(va, va));
        nb = _mm256_add_ps(nb, _mm256_mul_ps(vb, vb));
    }
    float dot = hsum_cos_avx(d), sa = hsum_cos_avx(na), sb = hsum_cos_avx(nb);
    for (; i < n; i++) {
        float va = a[i], vb = b[i];
        float p = va * vb;  dot = dot + p;
        float x = va * va;  sa = sa + x;
        float y = vb * vb;  sb = sb + y;
    }
    float denom = (float)(sqrt((double)sa) * sqrt((double)sb));
    if (denom < 1e-30f) return 1.0f;
    return 1.0f - dot / denom;
}
#endif

typedef float (*metric_fn)(const float *, const float *, int);

/* ------------------------------------------------------------------------------------
 * int8 rows with one float scale per row (BASELINE config 5).  NO reference counterpart: the
 * reference is generic over TDistance (src/HNSWIndex/HNSWIndex.cs:6) and ships float metrics only,
 * so this metric is the BUILDER'S OWN definition, stated here and implemented identically by the
 * HIP kernels (hnswindex.net_amd/csrc/device_kernels.h "int8 rows"); what the tests hold is
 * product == this restatement, bit for bit.
 *   record of `pitch` 32-bit words (a multiple of 16): [pitch-2 words: the elements, four int8 per
 *   word, little endian, zero padded | scale (float) | sumsq (int32)]
 *   scale = max|x| / 127 (float division); q_i = clamp(rintf(x_i / scale), -127, 127), 0 when the
 *   scale is not positive; sumsq = sum q_i^2
 *   distance(a, b) = (float)((A + B) - 2*C),  A = (sa*sa)*na, B = (sb*sb)*nb, C = (sa*sb)*dot in
 *   double, dot = sum q_a q_b in int32 -- the squared Euclidean distance of the dequantised vectors.
 * Inside the oracle an int8 index keeps ix->dim = pitch (records are addressed like float rows);
 * vectors arrive at the API as `udim` floats and are quantised on entry.
 * ---------------------------------------------------------------------------------- */
static int i8_pitch(int udim) { return (((udim + 3) / 4 + 2) + 15) & ~15; }
static void i8_quantize(const float *x, int udim, float *rec, int pitch)
{
    float m = 0.0f;
    for (int i = 0; i < udim; i++) m = fmaxf(m, fabsf(x[i]));
    const float scale = m / 127.0f;
    int32_t *w = (int32_t *)rec;
    int sumsq = 0;
    for (int k = 0; k < pitch - 2; k++) {
        uint32_t packed = 0;
        for (int t = 0; t < 4; t++) {
            int i = 4 * k + t, q = 0;
            if (i < udim && scale > 0.0f) {
                float v = rintf(x[i] / scale);
                v = fminf(fmaxf(v, -127.0f), 127.0f);
                q = (int)v;
            }
            sumsq += q * q;
            packed |= (uint32_t)(q & 0xff) << (8 * t);
        }
        w[k] = (int32_t)packed;
    }
    memcpy(&w[pitch - 2], &scale, 4);
    w[pitch - 1] = sumsq;
}
static float sq_euclid_i8(const float *a, const float *b, int pitch)
{
    const int8_t *qa = (const int8_t *)a, *qb = (const int8_t *)b;
    const int32_t *wa = (const int32_t *)a, *wb = (const int32_t *)b;
    int32_t dot = 0;
    for (int i = 0; i < 4 * (pitch - 2); i++) dot += (int32_t)qa[i] * (int32_t)qb[i];
    float sa, sb;
    memcpy(&sa, &wa[pitch - 2], 4);
    memcpy(&sb, &wb[pitch - 2], 4);
    const double A = ((double)sa * (double)sa) * (double)wa[pitch - 1];
    const double B = ((double)sb * (double)sb) * (double)wb[pitch - 1];
    const double C = ((double)sa * (double)sb) * (double)dot;
    return (float)((A + B) - 2.0 * C);
}

#if ORC_HAVE_AVX2
/* The same distance with the integer dot product on AVX2: sign-extend 16 bytes to 16-bit lanes, multiply-add pairs into 32-bit
 * lanes (|q| <= 127: a pair sums to at most 32 258), accumulate -- exact integer arithmetic, so the result is the scalar form's
 * bit for bit (tests/test_int8.py holds the two together).  This is the form the CPU baselines time (use_avx): the scalar byte
 * loop above is the SPEC, and a baseline measured on it would flatter the GPU. */
static float sq_euclid_i8_avx(const float *a, const float *b, int pitch)
{
    const int8_t *qa = (const int8_t *)a, *qb = (const int8_t *)b;
    const int32_t *wa = (const int32_t *)a, *wb = (const int32_t *)b;
    const int nbytes = 4 * (pitch - 2);
    __m256i acc = _mm256_setzero_si256();
    int i = 0;
    for (; i + 16 <= nbytes; i += 16) {
        const __m256i xa = _mm256_cvtepi8_epi16(_mm_loadu_si128((const __m128i *)(qa + i)));
        const __m256i xb = _mm256_cvtepi8_epi16(_mm_loadu_si128((const __m128i *)(qb + i)));
        acc = _mm256_add_epi32(acc, _mm256_madd_epi16(xa, xb));
    }
    __m128i s4 = _mm_add_epi32(_mm256_castsi256_si128(acc), _mm256_extracti128_si256(acc, 1));
    s4 = _mm_add_epi32(s4, _mm_shuffle_epi32(s4, 0x4e));
    s4 = _mm_add_epi32(s4, _mm_shuffle_epi32(s4, 0xb1));
    int32_t dot = _mm_cvtsi128_si32(s4);
    for (; i < nbytes; i++) dot += (int32_t)qa[i] * (int32_t)qb[i];
    float sa, sb;
    memcpy(&sa, &wa[pitch - 2], 4);
    memcpy(&sb, &wb[pitch - 2], 4);
    const double A = ((double)sa * (double)sa) * (double)wa[pitch - 1];
    const double B = ((double)sb * (double)sb) * (double)wb[pitch - 1];
    const double C = ((double)sa * (double)sb) * (double)dot;
    return (float)((A + B) - 2.0 * C);
}
#endif

static int cpu_has_avx2_fma(void)
{
#if ORC_HAVE_AVX2
    return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
#else
    return 0;
#endif
}

static metric_fn pick_metric(int metric, int want_avx)
{
#if ORC_HAVE_AVX2
    if (metric == ORC_SQ_EUCLID_I8) return (want_avx && cpu_has_avx2_fma()) ? sq_euclid_i8_avx : sq_euclid_i8; /* exact integers either way */
#endif
    if (metric == ORC_SQ_EUCLID_I8) return sq_euclid_i8;
#if ORC_HAVE_AVX2
    if (want_avx && cpu_has_avx2_fma()) {
        if (metric == ORC_SQ_EUCLID) return sq_euclid_avx;
        if (metric == ORC_COSINE) return cosine_avx;
        return ucosine_avx;
    }
#endif
    (void)want_avx;
    if (metric == ORC_SQ_EUCLID) return sq_euclid_spec;
    if (metric == ORC_COSINE) return cosine_spec;
    return ucosine_spec;
}

/* ------------------------------------------------------------------------------------
 * System.Random(int seed) -- .NET's seeded-compatibility generator.  NOT in /root/reference
 * (BCL); restated from the public dotnet/runtime algorithm.  Used by
 * src/HNSWIndex/GraphData.cs:42 (new Random(seed)) and :216 (NextSingle()).
 * ---------------------------------------------------------------------------------- */
typedef struct { int sa[56]; int inext, inextp; } dotnet_rng;

static void rng_init(dotnet_rng *r, int seed)
{
    int subtraction = (seed == INT32_MIN) ? INT32_MAX : abs(seed);
    int mj = 161803398 - subtraction;
    memset(r->sa, 0, sizeof r->sa);
    r->sa[55] = mj;
    int mk = 1, ii = 0;
    for (int i = 1; i < 55; i++) {
        if ((ii += 21) >= 55) ii -= 55;
        r->sa[ii] = mk;
        mk = mj - mk;
        if (mk < 0) mk += INT32_MAX;
        mj = r->sa[ii];
    }
    for (int k = 1; k < 5; k++)
        for (int i = 1; i < 56; i++) {
            int n = i + 30;
            if (n >= 55) n -= 55;
            /* wrap-around subtraction exactly as C# unchecked int arithmetic */
            r->sa[i] = (int)((uint32_t)r->sa[i] - (uint32_t)r->sa[1 + n]);
            if (r->sa[i] < 0) r->sa[i] += INT32_MAX;
        }
    r->inext = 0;
    r->inextp = 21;
}

static int rng_internal_sample(dotnet_rng *r)
{
    int li = r->inext, lp = r->inextp;
    if (++li >= 56) li = 1;
    if (++lp >= 56) lp = 1;
    int ret = (int)((uint32_t)r->sa[li] - (uint32_t)r->sa[lp]);
    if (ret == INT32_MAX) ret--;
    if (ret < 0) ret += INT32_MAX;
    r->sa[li] = ret;
    r->inext = li;
    r->inextp = lp;
    return ret;
}

static double rng_sample(dotnet_rng *r) { return rng_internal_sample(r) * (1.0 / INT32_MAX); }
/* NextSingle() of the seeded generator: (float)Sample(), drawn again while the cast rounds up to
 * 1.0f, so the result stays in [0, 1) (samples >= 2147483583 round up).  BCL behaviour, restated. */
static float single_of_sample(int internal) { return (float)(internal * (1.0 / INT32_MAX)); }
static float rng_next_single(dotnet_rng *r)
{
    for (;;) {
        float f = single_of_sample(rng_internal_sample(r));
        if (f < 1.0f) return f;
    }
}

/* src/HNSWIndex/GraphData.cs:211-219: (int)(-Math.Log(random) * distRate).
 * A non-finite or out-of-range product (random == 0) is returned as -1, which is what
 * the reference's "topLayer < 0 => return -1" path (GraphData.cs:82) sees on x64
 * runtimes that convert +inf to int.MinValue. */
static int level_from_uniform(float random, double dist_rate)
{
    double v = -log((double)random) * dist_rate;
    if (!(v < 2147483648.0) || !(v > -2147483649.0)) return -1;
    return (int)v;
}

/* ------------------------------------------------------------------------------------
 * Comparers -- src/HNSWIndex/DistanceComparer.cs:6-25.  float.CompareTo: NaN sorts
 * below every number, NaN == NaN.
 * ---------------------------------------------------------------------------------- */
static inline int float_compare_to(float x, float y)
{
    if (x < y) return -1;
    if (x > y) return 1;
    if (x == y) return 0;
    if (isnan(x)) return isnan(y) ? 0 : -1;
    return 1;
}
static inline int cmp_far(nd_t x, nd_t y) /* DistanceComparer :9-14 */
{
    if (x.dist < y.dist) return -1;
    if (x.dist > y.dist) return 1;
    return float_compare_to(x.dist, y.dist);
}
static inline int cmp_close(nd_t x, nd_t y) /* ReverseDistanceComparer :20-25 */
{
    if (x.dist > y.dist) return -1;
    if (x.dist < y.dist) return 1;
    return float_compare_to(y.dist, x.dist);
}

/* ------------------------------------------------------------------------------------
 * BinaryHeap -- src/HNSWIndex/BinaryHeap.cs:30-107.  `rev` selects the comparer.
 * ---------------------------------------------------------------------------------- */
typedef struct { nd_t *buf; int count, cap, rev; } heap_t;

static inline int heap_cmp(const heap_t *h, nd_t x, nd_t y) { return h->rev ? cmp_close(x, y) : cmp_far(x, y); }

static void heap_init(heap_t *h, int capacity, int rev)
{
    h->cap = capacity > 0 ? capacity : 0;
    h->buf = h->cap ? (nd_t *)malloc(sizeof(nd_t) * (size_t)h->cap) : NULL;
    h->count = 0;
    h->rev = rev;
}
static void heap_free(heap_t *h) { free(h->buf); h->buf = NULL; }

static void heap_sift_up(heap_t *h, int i, nd_t item) /* :89-107 */
{
    nd_t *b = h->buf;
    while (i > 0) {
        int p = (i - 1) >> 1;
        nd_t parent = b[p];
        if (heap_cmp(h, item, parent) <= 0) break;
        b[i] = parent;
        i = p;
    }
    b[i] = item;
}
static void heap_sift_down(heap_t *h, int i, nd_t item, int count) /* :67-87 */
{
    nd_t *b = h->buf;
    int half = count >> 1;
    while (i < half) {
        int left = (i << 1) + 1, right = left + 1;
        int mc = (right < count && heap_cmp(h, b[left], b[right]) < 0) ? right : left;
        if (heap_cmp(h, b[mc], item) <= 0) break;
        b[i] = b[mc];
        i = mc;
    }
    b[i] = item;
}
static void heap_push(heap_t *h, nd_t item) /* :30-34, growth :109-113 */
{
    if (h->count == h->cap) {
        h->cap = h->cap == 0 ? 16 : h->cap * 2;
        h->buf = (nd_t *)realloc(h->buf, sizeof(nd_t) * (size_t)h->cap);
    }
    heap_sift_up(h, h->count++, item);
}
static nd_t heap_pop(heap_t *h) /* :53-65 */
{
    nd_t result = h->buf[0];
    int jc = --h->count;
    nd_t last = h->buf[jc];
    if (jc != 0) heap_sift_down(h, 0, last, jc);
    return result;
}

/* ------------------------------------------------------------------------------------
 * MemoryExtensions.Sort(Span<NodeDistance>, DistanceComparer) -- BCL introsort, called at
 * src/HNSWIndex/Heuristic.cs:22.  NOT in /root/reference; restated from the public
 * dotnet/runtime ArraySortHelper<T> (threshold 16, median-of-three, heapsort fallback,
 * depth limit 2*(floor(log2 n)+1)).  Tie order: PARITY UNPINNED.
 * ---------------------------------------------------------------------------------- */
static inline void nd_swap(nd_t *k, int i, int j) { nd_t t = k[i]; k[i] = k[j]; k[j] = t; }
static inline void swap_if_greater(nd_t *k, int i, int j)
{
    if (cmp_far(k[i], k[j]) > 0) nd_swap(k, i, j);
}
static void insertion_sort(nd_t *k, int n)
{
    for (int i = 0; i < n - 1; i++) {
        nd_t t = k[i + 1];
        int j = i;
        while (j >= 0 && cmp_far(t, k[j]) < 0) {
            k[j + 1] = k[j];
            j--;
        }
        k[j + 1] = t;
    }
}
static void down_heap(nd_t *k, int i, int n)
{
    nd_t d = k[i - 1];
    while (i <= (n >> 1)) {
        int child = 2 * i;
        if (child < n && cmp_far(k[child - 1], k[child]) < 0) child++;
        if (!(cmp_far(d, k[child - 1]) < 0)) break;
        k[i - 1] = k[child - 1];
        i = child;
    }
    k[i - 1] = d;
}
static void heap_sort(nd_t *k, int n)
{
    for (int i = n >> 1; i >= 1; i--) down_heap(k, i, n);
    for (int i = n; i > 1; i--) {
        nd_swap(k, 0, i - 1);
        down_heap(k, 1, i - 1);
    }
}
static int pick_pivot_and_partition(nd_t *k, int n)
{
    int hi = n - 1, middle = hi >> 1;
    swap_if_greater(k, 0, middle);
    swap_if_greater(k, 0, hi);
    swap_if_greater(k, middle, hi);
    nd_t pivot = k[middle];
    nd_swap(k, middle, hi - 1);
    int left = 0, right = hi - 1;
    while (left < right) {
        while (cmp_far(k[++left], pivot) < 0) {}
        while (cmp_far(pivot, k[--right]) < 0) {}
        if (left >= right) break;
        nd_swap(k, left, right);
    }
    if (left != hi - 1) nd_swap(k, left, hi - 1);
    return left;
}
static void intro_sort(nd_t *k, int n, int depth_limit)
{
    int ps = n;
    while (ps > 1) {
        if (ps <= 16) {
            if (ps == 2) { swap_if_greater(k, 0, 1); return; }
            if (ps == 3) { swap_if_greater(k, 0, 1); swap_if_greater(k, 0, 2); swap_if_greater(k, 1, 2); return; }
            insertion_sort(k, ps);
            return;
        }
        if (depth_limit == 0) { heap_sort(k, ps); return; }
        depth_limit--;
        int p = pick_pivot_and_partition(k, ps);
        intro_sort(k + p + 1, ps - (p + 1), depth_limit);
        ps = p;
    }
}
static void dotnet_sort_nd(nd_t *k, int n)
{
    if (n > 1) {
        int lg = 31 - __builtin_clz((unsigned)n);
        intro_sort(k, n, 2 * (lg + 1));
    }
}

/* ------------------------------------------------------------------------------------
 * EdgeList / Node -- src/HNSWIndex/Node.cs:31-107 (append; swap-with-last removal).
 * ---------------------------------------------------------------------------------- */
typedef struct { int *buf; int count, cap; } edges_t;
typedef struct { int max_layer; edges_t *out, *in; int is_removed; } node_t;

static edges_t edges_new(int cap)
{
    edges_t e;
    e.cap = cap > 0 ? cap : 0;
    e.buf = e.cap ? (int *)malloc(sizeof(int) * (size_t)e.cap) : NULL;
    e.count = 0;
    return e;
}
static void edges_add(edges_t *e, int v) /* :66-76, growth :96-106 */
{
    if (e->cap < e->count + 1) {
        int nc = e->cap < 16 ? 16 : e->cap * 2;
        if (nc < e->count + 1) nc = e->count + 1;
        e->buf = (int *)realloc(e->buf, sizeof(int) * (size_t)nc);
        e->cap = nc;
    }
    e->buf[e->count++] = v;
}
static int edges_remove(edges_t *e, int v) /* :79-93 */
{
    for (int i = 0; i < e->count; i++)
        if (e->buf[i] == v) {
            int last = --e->count;
            if (i != last) e->buf[i] = e->buf[last];
            return 1;
        }
    return 0;
}
static edges_t edges_copy(const edges_t *o) /* EdgeList(EdgeList other) :42-47 */
{
    edges_t e = edges_new(o->count);
    memcpy(e.buf, o->buf, sizeof(int) * (size_t)o->count);
    e.count = o->count;
    return e;
}

/* ------------------------------------------------------------------------------------
 * Index
 * ---------------------------------------------------------------------------------- */
typedef struct {
    uint16_t *ver; int len; uint16_t cur; /* src/HNSWIndex/VisitedListPool.cs:10-67 (set semantics only) */
} visited_t;

typedef struct {
    int dim, metric; /* dim: floats per stored row (int8: the record pitch) */
    int udim;        /* elements per vector as the caller sees them */
    int max_edges, min_nn, max_candidates, seed, allow_removals, remove_max_candidates;
    double dist_rate;
    int capacity, length, count, entry;
    float *items; /* row-major length x dim: Items[id] (GraphData.cs:18) */
    node_t *nodes;
    dotnet_rng rng;
    metric_fn dist;
    /* ActiveSet (src/HNSWIndex/ActiveSet.cs:10-97) + RemovedIndexes (GraphData.cs:19, LIFO) */
    int *dense, *sparse, *removed_stack;
    int n_removed_stack;
    visited_t vis;          /* used by the single-threaded paths */
    visited_t *vis_pool;    /* per-thread lists of the threaded batched Add, kept across calls */
    int n_vis_pool;
    uint64_t n_eval;        /* distance evaluations (SURVEY 8d N_eval) */
    /* optional access log of the single-threaded Add (tools/window_sim.py): which adjacency lists an insert's
     * searches read and which its links write.  Entry = kind << 60 | layer << 40 | node; kind 0 read, 1 write,
     * 2 "item `node` starts". */
    int64_t *alog;
    size_t alog_n, alog_cap;
} index_t;

typedef struct { index_t *ix; visited_t *vis; uint64_t n_eval; } sctx_t;

static inline void alog_put(index_t *ix, int kind, int layer, int node)
{
    if (!ix->alog || ix->alog_n >= ix->alog_cap) return;
    ix->alog[ix->alog_n++] = ((int64_t)kind << 60) | ((int64_t)layer << 40) | (int64_t)(uint32_t)node;
}

/* Vectors as they arrive at the API (n x udim floats) -> what the index stores and measures (n x dim):
 * themselves, or their int8 records.  *tmp must be freed by the caller (NULL when nothing was made). */
static const float *incoming(const index_t *ix, const float *v, int n, float **tmp)
{
    *tmp = NULL;
    if (ix->metric != ORC_SQ_EUCLID_I8 || !v || n <= 0) return v;
    float *r = (float *)malloc(sizeof(float) * (size_t)n * (size_t)ix->dim);
    for (int i = 0; i < n; i++) i8_quantize(v + (size_t)i * (size_t)ix->udim, ix->udim, r + (size_t)i * (size_t)ix->dim, ix->dim);
    *tmp = r;
    return r;
}


static void visited_init(visited_t *v, int n) { v->ver = (uint16_t *)calloc((size_t)(n > 0 ? n : 1), 2); v->len = n > 0 ? n : 1; v->cur = 0; }
static void visited_free(visited_t *v) { free(v->ver); v->ver = NULL; }
static void visited_next(visited_t *v, int need)
{
    if (v->len < need) {
        free(v->ver);
        v->ver = (uint16_t *)calloc((size_t)need, 2);
        v->len = need;
        v->cur = 0;
    }
    v->cur++;
    if (v->cur == 0) { memset(v->ver, 0, (size_t)v->len * 2); v->cur++; }
}
static inline int visited_has(const visited_t *v, int id) { return id < v->len && v->ver[id] == v->cur; }
static inline void visited_add(visited_t *v, int id) { v->ver[id] = v->cur; }

static inline int max_edges_at(const index_t *ix, int layer) { return layer == 0 ? ix->max_edges * 2 : ix->max_edges; } /* GraphData.cs:247-250 */
static inline const float *item(const index_t *ix, int id) { return ix->items + (size_t)id * (size_t)ix->dim; }

/* GraphData.Distance(int, TVector) :274 */
static inline float dist_iq(sctx_t *c, int id, const float *q)
{
    c->n_eval++;
    return c->ix->dist(item(c->ix, id), q, c->ix->dim);
}
/* GraphData.Distance(int, int) :256 */
static inline float dist_ii(sctx_t *c, int a, int b)
{
    c->n_eval++;
    return c->ix->dist(item(c->ix, a), item(c->ix, b), c->ix->dim);
}

/* src/HNSWIndex/GraphNavigator.cs:51-82 (FindEntryAtLayer; no filter). */
static int find_entry_at_layer(sctx_t *c, int layer, int start, const float *q)
{
    index_t *ix = c->ix;
    int best = start;
    float cur = dist_iq(c, best, q); /* :57 */
    int changed = 1;
    while (changed) {
        changed = 0;
        /* :65 the span is taken once per pass: `best` may move mid-scan, the span does not */
        const edges_t *e = &ix->nodes[best].out[layer];
        const int *conn = e->buf;
        int n = e->count;
        alog_put(ix, 0, layer, best);
        for (int i = 0; i < n; i++) {
            int cand = conn[i];
            float d = dist_iq(c, cand, q); /* :70 */
            if (d < cur) {                 /* :71 */
                cur = d;
                best = cand;
                changed = 1;
            }
        }
    }
    return best;
}

/* GraphNavigator.cs:27-33 / :39-45 */
static int find_entry_point(sctx_t *c, int dst_layer, const float *q)
{
    index_t *ix = c->ix;
    int best = ix->entry;
    for (int layer = ix->nodes[best].max_layer; layer > dst_layer; layer--)
        best = find_entry_at_layer(c, layer, best, q);
    return best;
}

/* GraphNavigator.cs:123-189 (SearchLayer) == :194-256 (SearchLayerQuery) minus locks.
 * Returns the top-candidate heap's buffer prefix (heap order) in *out (malloc'd). */
static int search_layer_f(sctx_t *c, int entry_id, int layer, int k, const float *q, nd_t **out, int exclude_id);
static int search_layer(sctx_t *c, int entry_id, int layer, int k, const float *q, nd_t **out)
{
    return search_layer_f(c, entry_id, layer, k, q, out, -1);
}
/* exclude_id >= 0: the filter `id => id != removedNode.Id` of GraphConnector.cs:96 */
static int search_layer_f(sctx_t *c, int entry_id, int layer, int k, const float *q, nd_t **out, int exclude_id)
{
    index_t *ix = c->ix;
    heap_t top, cand;
    heap_init(&top, k, 0);      /* :126 fartherFirst */
    heap_init(&cand, k * 2, 1); /* :127 closerFirst */
    nd_t entry = {entry_id, dist_iq(c, entry_id, q)}; /* :129 */
    float farthest = 3.402823466e+38f;   /* TDistance.MaxValue :130 */
    if (entry_id != exclude_id) {        /* filterFnc(entryPointId) :132 */
        heap_push(&top, entry);          /* :134 */
        farthest = entry.dist;           /* :135 */
    }
    heap_push(&cand, entry);    /* :138 */
    visited_next(c->vis, ix->capacity);
    visited_add(c->vis, entry_id); /* :140 */
    while (cand.count > 0) {
        nd_t closest = heap_pop(&cand);                         /* :146 */
        if (closest.dist > farthest && top.count >= k) break;   /* :147 */
        const edges_t *e = &ix->nodes[closest.id].out[layer];
        alog_put(ix, 0, layer, closest.id);
        if (ix->alog) { union { float f; uint32_t u; } fu; fu.f = farthest; alog_put(ix, 4, layer, (int)(top.count >= k ? fu.u : 0xFFFFFFFFu)); }
        for (int i = 0; i < e->count; ++i) {
            int nb = e->buf[i];
            if (visited_has(c->vis, nb)) continue;              /* :161 */
            float d = dist_iq(c, nb, q);                        /* :163 */
            if (top.count < k || d < farthest) {                /* :165 */
                nd_t sel = {nb, d};
                heap_push(&cand, sel);                          /* :168 */
                if (sel.id != exclude_id) heap_push(&top, sel); /* :170-171 */
                if (top.count > k) heap_pop(&top);              /* :173-174 */
                if (top.count > 0) farthest = top.buf[0].dist;  /* :176-177 */
            }
            visited_add(c->vis, nb);                            /* :181 */
        }
    }
    int n = top.count;
    *out = top.buf; /* ToArray(): buffer prefix, BinaryHeap.cs:41-44 */
    heap_free(&cand);
    return n;
}

/* GraphNavigator.SearchLayerRange (src/HNSWIndex/GraphNavigator.cs:262-325), no filter. */
static int search_layer_range(sctx_t *c, int entry_id, int layer, float range, const float *q, nd_t **out)
{
    index_t *ix = c->ix;
    heap_t top, cand;
    heap_init(&top, max_edges_at(ix, layer), 0);      /* :265 */
    heap_init(&cand, max_edges_at(ix, layer) * 2, 1); /* :266 */
    nd_t entry = {entry_id, dist_iq(c, entry_id, q)}; /* :268 */
    float farthest = 3.402823466e+38f;                /* TDistance.MaxValue :269 */
    if (entry.dist <= range) { heap_push(&top, entry); farthest = entry.dist; } /* :271-275 */
    heap_push(&cand, entry);                          /* :277 */
    visited_next(c->vis, ix->capacity);
    visited_add(c->vis, entry_id);                    /* :279 */
    while (cand.count > 0) {
        nd_t closest = cand.buf[0];                                   /* Peek :285 */
        if (closest.dist > farthest && closest.dist > range) break;  /* :286-289 */
        heap_pop(&cand);                                              /* :290 */
        const edges_t *e = &ix->nodes[closest.id].out[layer];
        for (int i = 0; i < e->count; ++i) {
            int nb = e->buf[i];
            if (visited_has(c->vis, nb)) continue;   /* :297 */
            float d = dist_iq(c, nb, q);             /* :299 */
            if (d <= range) {                        /* :302 */
                nd_t sel = {nb, d};
                heap_push(&cand, sel);               /* :305 */
                heap_push(&top, sel);                /* :308 */
                if (top.buf[0].dist > range) heap_pop(&top);       /* :310-311 */
                if (top.count > 0) farthest = top.buf[0].dist;     /* :313-314 */
            }
            visited_add(c->vis, nb);                 /* :318 */
        }
    }
    int n = top.count;
    *out = top.buf;
    heap_free(&cand);
    return n;
}

/* src/HNSWIndex/Heuristic.cs:11-46.  `cands` is sorted in place (as the reference's span). */
static edges_t relative_neighbor_pruning(sctx_t *c, nd_t *cands, int n, int max_edges)
{
    if (n < max_edges) { /* :13-18: ids in input order, unsorted */
        edges_t ids = edges_new(n);
        for (int i = 0; i < n; i++) edges_add(&ids, cands[i].id);
        return ids;
    }
    int rc = 0;
    nd_t *res = (nd_t *)malloc(sizeof(nd_t) * (size_t)(max_edges + 1));
    dotnet_sort_nd(cands, n); /* :22 */
    for (int i = 0; i < n && rc < max_edges; i++) {
        nd_t cand = cands[i];
        int ok = 1;
        for (int j = 0; j < rc; j++)
            if (dist_ii(c, res[j].id, cand.id) < cand.dist) { ok = 0; break; } /* :34 */
        if (ok) res[rc++] = cand;
    }
    edges_t out = edges_new(max_edges + 1);
    for (int k = 0; k < rc; k++) edges_add(&out, res[k].id);
    free(res);
    return out;
}

/* src/HNSWIndex/GraphConnector.cs:222-262 */
static void prune_overflow(sctx_t *c, int node_id, int layer)
{
    index_t *ix = c->ix;
    node_t *node = &ix->nodes[node_id];
    edges_t old = node->out[layer];
    nd_t *cd = (nd_t *)malloc(sizeof(nd_t) * (size_t)old.count);
    for (int i = 0; i < old.count; i++) {
        cd[i].id = old.buf[i];
        cd[i].dist = dist_ii(c, old.buf[i], node_id); /* :233 */
    }
    edges_t nw = relative_neighbor_pruning(c, cd, old.count, max_edges_at(ix, layer)); /* :235 */
    node->out[layer] = nw;
    free(cd);
    if (ix->allow_removals) { /* :239-261 */
        for (int i = 0; i < old.count; i++) {
            int id = old.buf[i], keep = 0;
            for (int j = 0; j < nw.count; j++)
                if (nw.buf[j] == id) { keep = 1; break; }
            if (!keep) edges_remove(&ix->nodes[id].in[layer], node_id);
        }
    }
    free(old.buf);
}

/* src/HNSWIndex/GraphConnector.cs:187-217 */
static int connect_at_layer(sctx_t *c, int cur_id, int best_peer, int layer)
{
    index_t *ix = c->ix;
    nd_t *topc;
    int n = search_layer(c, best_peer, layer, ix->max_candidates, item(ix, cur_id), &topc); /* :189 */
    edges_t best = relative_neighbor_pruning(c, topc, n, max_edges_at(ix, layer));          /* :190 */
    free(topc);
    node_t *cur = &ix->nodes[cur_id];
    free(cur->out[layer].buf);
    cur->out[layer] = best; /* :192 */
    if (ix->allow_removals) { /* :193 */
        free(cur->in[layer].buf);
        cur->in[layer] = edges_copy(&best);
    }
    int first = best.buf[0];
    int cnt = best.count;
    for (int i = 0; i < cnt; ++i) {
        /* re-read through the node: `best` aliases cur->out[layer], which a prune of a
         * neighbour never touches (cur is not yet anyone's overflow victim) */
        int nb_id = ix->nodes[cur_id].out[layer].buf[i];
        node_t *nb = &ix->nodes[nb_id];
        if (ix->allow_removals) edges_add(&nb->in[layer], cur_id); /* :204 */
        edges_add(&nb->out[layer], cur_id);                        /* :207 */
        if (nb->out[layer].count > max_edges_at(ix, layer)) {      /* :209-212 */
            int alog_old[160], alog_old_n = 0;
            if (ix->alog) for (int t = 0; t + 1 < nb->out[layer].count && t < 160; t++) alog_old[alog_old_n++] = nb->out[layer].buf[t]; /* the list before the append */
            prune_overflow(c, nb_id, layer);
            if (ix->alog) { /* kind 3: the prune turned the new item away and kept everything else (the list is the same set) */
                const edges_t *e = &ix->nodes[nb_id].out[layer];
                int has = 0;
                for (int t = 0; t < e->count; t++) has |= e->buf[t] == cur_id;
                const int same = !has && e->count == max_edges_at(ix, layer);
                alog_put(ix, same ? 3 : 1, layer, nb_id);
                if (!same) { /* kind 5: an id the list gained, kind 6: an id it lost (old = saved copy below) */
                    if (has) alog_put(ix, 5, layer, cur_id);
                    for (int t = 0; t < alog_old_n; t++) {
                        int keep = 0;
                        for (int u = 0; u < e->count; u++) keep |= e->buf[u] == alog_old[t];
                        if (!keep) alog_put(ix, 6, layer, alog_old[t]);
                    }
                }
            }
        } else { alog_put(ix, 1, layer, nb_id); alog_put(ix, 5, layer, cur_id); }
    }
    return first; /* :216 */
}

/* GraphData.NewNode :224-242 */
static void node_init(index_t *ix, node_t *nd, int top_layer)
{
    nd->max_layer = top_layer;
    nd->out = (edges_t *)malloc(sizeof(edges_t) * (size_t)(top_layer + 1));
    nd->in = ix->allow_removals ? (edges_t *)malloc(sizeof(edges_t) * (size_t)(top_layer + 1)) : NULL;
    for (int l = 0; l <= top_layer; l++) {
        nd->out[l] = edges_new(max_edges_at(ix, l) + 1);
        if (ix->allow_removals) nd->in[l] = edges_new(max_edges_at(ix, l) + 1);
    }
}

static void grow(index_t *ix) /* GraphData.cs:98-111 */
{
    int nc = ix->capacity * 2;
    if (nc < 1) nc = 1;
    ix->items = (float *)realloc(ix->items, sizeof(float) * (size_t)nc * (size_t)ix->dim);
    ix->nodes = (node_t *)realloc(ix->nodes, sizeof(node_t) * (size_t)nc);
    memset(ix->nodes + ix->capacity, 0, sizeof(node_t) * (size_t)(nc - ix->capacity));
    ix->dense = (int *)realloc(ix->dense, sizeof(int) * (size_t)nc);
    ix->sparse = (int *)realloc(ix->sparse, sizeof(int) * (size_t)nc);
    ix->removed_stack = (int *)realloc(ix->removed_stack, sizeof(int) * (size_t)nc);
    ix->capacity = nc;
}

static void node_free_lists(node_t *nd)
{
    if (!nd->out) return;
    for (int l = 0; l <= nd->max_layer; l++) {
        free(nd->out[l].buf);
        if (nd->in) free(nd->in[l].buf);
    }
    free(nd->out);
    free(nd->in);
    nd->out = nd->in = NULL;
}

/* GraphData.AddItem (src/HNSWIndex/GraphData.cs:79-118): level draw, then either the most
 * recently vacated slot (RemovedIndexes.TryPop, :85-91) or a fresh one; ActiveSet.Add. */
static int alloc_node(index_t *ix, const float *v)
{
    int top_layer = level_from_uniform(rng_next_single(&ix->rng), ix->dist_rate); /* :81 */
    if (top_layer < 0) return -1;                                                 /* :82 */
    int id;
    if (ix->allow_removals && ix->n_removed_stack > 0) {
        id = ix->removed_stack[--ix->n_removed_stack];
        node_free_lists(&ix->nodes[id]);
    } else {
        id = ix->length++;
        if (ix->length > ix->capacity) grow(ix);
    }
    node_init(ix, &ix->nodes[id], top_layer);
    ix->nodes[id].is_removed = 0;
    memcpy(ix->items + (size_t)id * (size_t)ix->dim, v, sizeof(float) * (size_t)ix->dim);
    ix->dense[ix->count] = id; /* ActiveSet.Add :72-80 */
    ix->sparse[id] = ix->count;
    ix->count++;
    return id;
}

/* HNSWIndex.Add(item) src/HNSWIndex/HNSWIndex.cs:55-65 -> GraphData.AddItem :79-118 ->
 * GraphConnector.ConnectNewNode :24-47 -> AddNewConnections :172-181. */
static int add_one(sctx_t *c, const float *v)
{
    index_t *ix = c->ix;
    int id = alloc_node(ix, v);
    if (id < 0) return -1;
    if (ix->entry < 0) { ix->entry = id; return id; } /* GraphConnector.cs:28-33 */
    alog_put(ix, 2, 0, id);
    node_t *cur = &ix->nodes[id];
    int top = ix->nodes[ix->entry].max_layer; /* GetTopLayer :195-198 */
    int new_ep = cur->max_layer > top;        /* :36 */
    /* AddNewConnections */
    int best = find_entry_point(c, cur->max_layer, item(ix, id)); /* :174 */
    int start = cur->max_layer < top ? cur->max_layer : top;      /* :176 */
    for (int layer = start; layer >= 0; --layer)
        best = connect_at_layer(c, id, best, layer); /* :178-179 */
    if (new_ep) ix->entry = id; /* :39 */
    return id;
}

/* ------------------------------------------------------------------------------------
 * Snapshot-batched Add -- NOT a reference code path.  It is the deterministic schedule the
 * product's hnsw_add uses for count > 1 (DESIGN.md "Add"), restated here on the CPU so that
 * the GPU result can be checked bit for bit.  It is one legal outcome of the reference's
 * Parallel.For Add(List) (HNSWIndex.cs:70-78) in which the B items of a batch all finish
 * their searches before any of them links:
 *   search half, per item, on the graph as it stands before the batch:
 *       FindEntryPoint + for every layer SearchLayer -> RelativeNeighborPruning
 *       (GraphConnector.cs:174-179, :189-190; next entry = selected[0], :216)
 *   link half, items in id order: OutEdges = selected (:192), back-edges and
 *       PruneOverflow (:196-214).
 * A batch is at most max(1, linked/4) items while fewer than min(65 536, count after the call / 16) nodes
 * are linked, max(1, linked/16) afterwards (capped by max_batch); an item whose level
 * exceeds the current top layer is inserted alone (the reference holds the entry-point lock
 * for it, GraphConnector.cs:27-41).  max_batch == 1 is exactly orc_add.
 * ---------------------------------------------------------------------------------- */
static void batch_search(sctx_t *c, int id, edges_t *sel /* [max_layer+1] */)
{
    index_t *ix = c->ix;
    node_t *cur = &ix->nodes[id];
    int top = ix->nodes[ix->entry].max_layer;
    int best = find_entry_point(c, cur->max_layer, item(ix, id));
    int start = cur->max_layer < top ? cur->max_layer : top;
    for (int l = 0; l <= cur->max_layer; l++) { sel[l].buf = NULL; sel[l].count = 0; sel[l].cap = 0; }
    for (int layer = start; layer >= 0; --layer) {
        nd_t *topc;
        int n = search_layer(c, best, layer, ix->max_candidates, item(ix, id), &topc);
        sel[layer] = relative_neighbor_pruning(c, topc, n, max_edges_at(ix, layer));
        free(topc);
        best = sel[layer].buf[0];
    }
}

static void batch_link(sctx_t *c, int id, edges_t *sel)
{
    index_t *ix = c->ix;
    int top = ix->nodes[ix->entry].max_layer;
    int lvl = ix->nodes[id].max_layer;
    int start = lvl < top ? lvl : top;
    for (int layer = start; layer >= 0; --layer) {
        node_t *cur = &ix->nodes[id];
        free(cur->out[layer].buf);
        cur->out[layer] = sel[layer];
        if (ix->allow_removals) { free(cur->in[layer].buf); cur->in[layer] = edges_copy(&sel[layer]); }
        int cnt = sel[layer].count;
        for (int i = 0; i < cnt; ++i) {
            int nb_id = ix->nodes[id].out[layer].buf[i];
            node_t *nb = &ix->nodes[nb_id];
            if (ix->allow_removals) edges_add(&nb->in[layer], id);
            edges_add(&nb->out[layer], id);
            if (nb->out[layer].count > max_edges_at(ix, layer)) prune_overflow(c, nb_id, layer);
        }
    }
}

ORC_API int orc_add_batched(void *h, const float *v, int n, int *out_ids, int max_batch)
{
    index_t *ix = (index_t *)h;
    if (!ix || !v || n <= 0) return 0;
    if (max_batch < 1) max_batch = 1;
    sctx_t c = {ix, &ix->vis, 0};
    int *ids = (int *)malloc(sizeof(int) * (size_t)n);
    int m = 0;
    float *tmp;
    v = incoming(ix, v, n, &tmp);
    for (int i = 0; i < n; i++) {
        int id = alloc_node(ix, v + (size_t)i * (size_t)ix->dim);
        if (out_ids) out_ids[i] = id;
        if (id >= 0) ids[m++] = id;
    }
    free(tmp);
    int p = 0;
    while (p < m) {
        if (ix->entry < 0) { ix->entry = ids[p++]; continue; }
        int top = ix->nodes[ix->entry].max_layer;
        int nb = 1, new_ep = 0;
        if (ix->nodes[ids[p]].max_layer > top) {
            new_ep = 1;
        } else {
            int linked = ix->count - (m - p); /* nodes already linked (== the id when nothing was ever removed) */
            int early = ix->count / 16 < 65536 ? ix->count / 16 : 65536; /* the product's growth rule (hnsw_index.cpp) */
            int b = linked / (linked < early ? 4 : 16);
            if (b < 1) b = 1;
            if (b > max_batch) b = max_batch;
            while (nb < b && p + nb < m && ix->nodes[ids[p + nb]].max_layer <= top) nb++;
        }
        edges_t **sels = (edges_t **)malloc(sizeof(edges_t *) * (size_t)nb);
        for (int i = 0; i < nb; i++) {
            sels[i] = (edges_t *)calloc((size_t)ix->nodes[ids[p + i]].max_layer + 1, sizeof(edges_t));
            batch_search(&c, ids[p + i], sels[i]);
        }
        for (int i = 0; i < nb; i++) {
            batch_link(&c, ids[p + i], sels[i]);
            free(sels[i]);
        }
        free(sels);
        if (new_ep) ix->entry = ids[p];
        p += nb;
    }
    free(ids);
    ix->n_eval += c.n_eval;
    return n;
}

/* The same schedule with the search half of every batch spread over `threads` host threads (one
 * visited list each) -- what a T-core host running this schedule does; the link half stays
 * sequential.  Same graph as orc_add_batched whatever the thread count (the searches of a batch
 * read the graph as it stands before the batch and write nothing).  bench.py's cpu_baseline leg
 * times it beside the GPU's batched Add. */
typedef struct {
    index_t *ix; visited_t *vis; const int *ids; edges_t **sels; int nb;
    volatile int *next; uint64_t n_eval;
} bjob_t;
static void *batch_search_worker(void *arg)
{
    bjob_t *j = (bjob_t *)arg;
    sctx_t c = {j->ix, j->vis, 0};
    for (;;) {
        int i = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED);
        if (i >= j->nb) break;
        batch_search(&c, j->ids[i], j->sels[i]);
    }
    j->n_eval = c.n_eval;
    return NULL;
}
typedef struct { index_t *ix; const int *ids; int nb, top, t, nthreads; uint64_t n_eval; } ljob_t;
static void *batch_link_worker(void *arg)
{
    ljob_t *j = (ljob_t *)arg;
    index_t *ix = j->ix;
    sctx_t c = {ix, NULL, 0};
    for (int i = 0; i < j->nb; i++) {
        int id = j->ids[i], lvl = ix->nodes[id].max_layer;
        for (int layer = lvl < j->top ? lvl : j->top; layer >= 0; --layer) {
            const edges_t *sel = &ix->nodes[id].out[layer]; /* this item's own list: nobody appends to it during the batch */
            for (int e = 0; e < sel->count; ++e) {
                int nb_id = sel->buf[e];
                if (nb_id % j->nthreads != j->t) continue;
                node_t *nbn = &ix->nodes[nb_id];
                edges_add(&nbn->out[layer], id);
                if (nbn->out[layer].count > max_edges_at(ix, layer)) prune_overflow(&c, nb_id, layer);
            }
        }
    }
    j->n_eval = c.n_eval;
    return NULL;
}
ORC_API int orc_add_batched_mt(void *h, const float *v, int n, int *out_ids, int max_batch, int threads)
{
    index_t *ix = (index_t *)h;
    if (!ix || !v || n <= 0) return 0;
    if (threads <= 1) return orc_add_batched(h, v, n, out_ids, max_batch);
    if (threads > 256) threads = 256;
    if (max_batch < 1) max_batch = 1;
    sctx_t c = {ix, &ix->vis, 0};
    int *ids = (int *)malloc(sizeof(int) * (size_t)n);
    int m = 0;
    float *tmp;
    v = incoming(ix, v, n, &tmp);
    for (int i = 0; i < n; i++) {
        int id = alloc_node(ix, v + (size_t)i * (size_t)ix->dim);
        if (out_ids) out_ids[i] = id;
        if (id >= 0) ids[m++] = id;
    }
    free(tmp);
    if (ix->n_vis_pool < threads) { /* allocated once: a 16-item call must not pay for 16 fresh lists of `capacity` entries */
        ix->vis_pool = (visited_t *)realloc(ix->vis_pool, sizeof(visited_t) * (size_t)threads);
        for (int t = ix->n_vis_pool; t < threads; t++) visited_init(&ix->vis_pool[t], ix->capacity);
        ix->n_vis_pool = threads;
    }
    visited_t *vis = ix->vis_pool;
    int p = 0;
    while (p < m) {
        if (ix->entry < 0) { ix->entry = ids[p++]; continue; }
        int top = ix->nodes[ix->entry].max_layer;
        int nb = 1, new_ep = 0;
        if (ix->nodes[ids[p]].max_layer > top) {
            new_ep = 1;
        } else {
            int linked = ix->count - (m - p);
            int early = ix->count / 16 < 65536 ? ix->count / 16 : 65536; /* the product's growth rule (hnsw_index.cpp) */
            int b = linked / (linked < early ? 4 : 16);
            if (b < 1) b = 1;
            if (b > max_batch) b = max_batch;
            while (nb < b && p + nb < m && ix->nodes[ids[p + nb]].max_layer <= top) nb++;
        }
        edges_t **sels = (edges_t **)malloc(sizeof(edges_t *) * (size_t)nb);
        for (int i = 0; i < nb; i++) sels[i] = (edges_t *)calloc((size_t)ix->nodes[ids[p + i]].max_layer + 1, sizeof(edges_t));
        int used = nb < threads ? nb : threads;
        if (used <= 1) {
            for (int i = 0; i < nb; i++) batch_search(&c, ids[p + i], sels[i]);
        } else {
            pthread_t th[256];
            bjob_t jobs[256];
            volatile int next = 0;
            for (int t = 0; t < used; t++) {
                jobs[t] = (bjob_t){ix, &vis[t], ids + p, sels, nb, &next, 0};
                pthread_create(&th[t], NULL, batch_search_worker, &jobs[t]);
            }
            for (int t = 0; t < used; t++) { pthread_join(th[t], NULL); c.n_eval += jobs[t].n_eval; }
        }
        if (used <= 1 || ix->allow_removals) { /* in-edge upkeep touches other nodes' lists: sequential */
            for (int i = 0; i < nb; i++) batch_link(&c, ids[p + i], sels[i]);
        } else {
            /* link half by target list: OutEdges = selected for every item, then each thread applies, in
             * item order, the back-edge appends (and overflow prunes) of the lists it owns (nb % threads).
             * Lists are independent, so this is the sequential outcome. */
            for (int i = 0; i < nb; i++) {
                int id = ids[p + i], lvl = ix->nodes[id].max_layer;
                for (int layer = lvl < top ? lvl : top; layer >= 0; --layer) {
                    free(ix->nodes[id].out[layer].buf);
                    ix->nodes[id].out[layer] = sels[i][layer];
                }
            }
            pthread_t th[256];
            ljob_t jobs[256];
            for (int t = 0; t < used; t++) {
                jobs[t] = (ljob_t){ix, ids + p, nb, top, t, used, 0};
                pthread_create(&th[t], NULL, batch_link_worker, &jobs[t]);
            }
            for (int t = 0; t < used; t++) { pthread_join(th[t], NULL); c.n_eval += jobs[t].n_eval; }
        }
        for (int i = 0; i < nb; i++) free(sels[i]);
        free(sels);
        if (new_ep) ix->entry = ids[p];
        p += nb;
    }
    free(ids);
    ix->n_eval += c.n_eval;
    return n;
}

/* ------------------------------------------------------------------------------------
 * Tick schedule -- NOT a reference code path and NOT a product path: the CPU model behind DESIGN.md 4.3's analysis.
 * One more deterministic member of the outcome set of HNSWIndex.Add(List)'s Parallel.For (HNSWIndex.cs:70-78) on a host that
 * holds `slots` items in flight: ids are handed out at AddItem in start order (:57) and an item holds its thread -- and its own
 * OutEdgesLock (:60-63) -- from there to its last ConnectAtLayer, so any execution that STARTS items in id order with at most
 * `slots` of them between start and finish is an interleaving real threads can produce; the slots need not turn over together.
 * Time runs in ticks.  In a tick
 *   1. free slots are filled in id order: a starting item runs FindEntryPoint (GraphConnector.cs:174) on the graph as it stands;
 *   2. every item in flight searches on that same graph (SearchLayer + RelativeNeighborPruning, :189-190; next entry =
 *      selected[0], :216): an item on one layer that layer; an item on more its TOP layer in its first tick and all the layers
 *      below, one after the other, in its second;
 *   3. in id order, the items that have searched layer 0 link all their layers, top down (:192-214), and leave their slots.
 * So every multi-layer item splits its top layer off into a tick of its own: one in sixteen items takes two ticks, and a tick is
 * as long as ONE traversal unless it holds the second half of an item on three or more layers (one in 256).
 * How far a link may wait, and why not further: ConnectAtLayer(l) writes lists of layer l only, the search of layer l - 1 reads none
 * of them, and reads do not conflict with reads -- so the real execution "every layer-l read of the tick, then the item's link of
 * layer l, then its search of layer l - 1, ..., then the tick's links of layer 0" respects every thread's program order and gives
 * exactly this schedule's graph (links of different layers commute).  Hence the link of layer l may wait for the tick in which
 * layer l - 1 is searched, and NO longer (the descents of the items starting a tick later read layer l and would have to see it).
 * Two variants that take ALL layers apart were measured here first and dropped: linked tick by tick, a node is visible on an
 * upper layer while its lists below are still empty -- the reference's own race; an item whose descent ends on such a node finds
 * nothing below and links to that node alone: recall@10 0.647 against 0.692 sequential at 6 000 x 32, 0.470 against 0.478 at
 * 100 000 x 32 even when only items on three or more layers are exposed; linked all at the end, recall is on par but an item on
 * three layers links its top layer two ticks late, which is not an interleaving.  With the rule above every node becomes visible
 * on all its layers at once AND every link is within its bound.
 * An item whose level exceeds the top layer starts only when nothing is in flight and is inserted alone (entry-point lock,
 * GraphConnector.cs:27-41).  The number of slots grows with the graph by the batched schedule's rule.  slots == 1 is exactly
 * orc_add.  stats (may be NULL): [0] ticks, [1] item-ticks, [2] most items in flight in a tick, [3] items inserted alone,
 * [4] ticks that hold the second half of an item on three or more layers (two traversals long or more).
 * ---------------------------------------------------------------------------------- */
typedef struct { int id, layer, best; edges_t *sel; /* [max_layer + 1] */ } tick_item_t;
static void link_layer(sctx_t *c, int id, int layer, edges_t sel) /* batch_link's body for one layer */
{
    index_t *ix = c->ix;
    node_t *cur = &ix->nodes[id];
    free(cur->out[layer].buf);
    cur->out[layer] = sel;
    if (ix->allow_removals) { free(cur->in[layer].buf); cur->in[layer] = edges_copy(&sel); }
    int cnt = sel.count;
    for (int i = 0; i < cnt; ++i) {
        int nb_id = ix->nodes[id].out[layer].buf[i];
        node_t *nb = &ix->nodes[nb_id];
        if (ix->allow_removals) edges_add(&nb->in[layer], id);
        edges_add(&nb->out[layer], id);
        if (nb->out[layer].count > max_edges_at(ix, layer)) prune_overflow(c, nb_id, layer);
    }
}
ORC_API int orc_add_ticks(void *h, const float *v, int n, int *out_ids, int slots, uint64_t *stats)
{
    index_t *ix = (index_t *)h;
    if (!ix || !v || n <= 0) return 0;
    if (slots < 1) slots = 1;
    sctx_t c = {ix, &ix->vis, 0};
    int *ids = (int *)malloc(sizeof(int) * (size_t)n);
    int m = 0;
    float *tmp;
    v = incoming(ix, v, n, &tmp);
    for (int i = 0; i < n; i++) {
        int id = alloc_node(ix, v + (size_t)i * (size_t)ix->dim);
        if (out_ids) out_ids[i] = id;
        if (id >= 0) ids[m++] = id;
    }
    free(tmp);
    tick_item_t *act = (tick_item_t *)malloc(sizeof(tick_item_t) * (size_t)slots);
    int na = 0, p = 0, finished = 0;
    const int before = ix->count - m; /* nodes linked before this call */
    uint64_t st[5] = {0, 0, 0, 0, 0};
    while (p < m || na > 0) {
        /* 1. starts, in id order */
        while (p < m) {
            if (ix->entry < 0) { if (na) break; ix->entry = ids[p++]; finished++; continue; }
            const int id = ids[p], lvl = ix->nodes[id].max_layer, top = ix->nodes[ix->entry].max_layer;
            if (lvl > top) { /* a new entry point: alone, exactly add_one's body */
                if (na) break;
                int best = find_entry_point(&c, lvl, item(ix, id));
                for (int layer = top; layer >= 0; --layer) best = connect_at_layer(&c, id, best, layer);
                ix->entry = id;
                p++; finished++; st[3]++;
                continue;
            }
            int linked = before + finished;
            int early = ix->count / 16 < 65536 ? ix->count / 16 : 65536;
            int cap = linked / (linked < early ? 4 : 16);
            if (cap < 1) cap = 1;
            if (cap > slots) cap = slots;
            if (na >= cap) break;
            act[na].id = id;
            act[na].layer = lvl; /* lvl <= top here */
            act[na].best = find_entry_point(&c, lvl, item(ix, id));
            act[na].sel = (edges_t *)calloc((size_t)lvl + 1, sizeof(edges_t));
            na++; p++;
        }
        if (na == 0) continue;
        st[0]++;
        st[1] += (uint64_t)na;
        if ((uint64_t)na > st[2]) st[2] = (uint64_t)na;
        /* 2. searches: the graph is only read.  First tick of a multi-layer item: its top layer; otherwise everything below */
        int long_job = 0;
        for (int i = 0; i < na; i++) {
            const int id = act[i].id, lvl = ix->nodes[id].max_layer;
            const int first_tick = act[i].layer == lvl && lvl > 0;
            const int stop = first_tick ? lvl : 0; /* lowest layer searched in this tick */
            if (!first_tick && act[i].layer >= 1) long_job = 1; /* layers 1 and 0 (at least) in one job */
            for (int layer = act[i].layer; layer >= stop; --layer) {
                nd_t *topc;
                int cnt = search_layer(&c, act[i].best, layer, ix->max_candidates, item(ix, id), &topc);
                act[i].sel[layer] = relative_neighbor_pruning(&c, topc, cnt, max_edges_at(ix, layer));
                free(topc);
                act[i].best = act[i].sel[layer].buf[0];
            }
            act[i].layer = stop;
        }
        st[4] += (uint64_t)long_job;
        /* 3. in id order (act is kept in id order): the items that have searched layer 0 link every layer, top down */
        int keep = 0;
        for (int i = 0; i < na; i++) {
            const int id = act[i].id;
            if (act[i].layer == 0) {
                for (int layer = ix->nodes[id].max_layer; layer >= 0; --layer) link_layer(&c, id, layer, act[i].sel[layer]);
                free(act[i].sel);
                finished++;
                continue;
            }
            act[keep] = act[i];
            act[keep].layer--; /* `best` already is the entry of the next layer */
            keep++;
        }
        na = keep;
    }
    free(act);
    free(ids);
    ix->n_eval += c.n_eval;
    if (stats) memcpy(stats, st, sizeof st);
    return n;
}

/* Advances the level generator by n draws without inserting anything: after orc_import_nodes the
 * generator stands at its seed, while the index the graph came from has drawn one level per node
 * (GraphData.cs:211-219).  With the draws skipped, Adds on the imported graph continue exactly as
 * they would on the original. */
ORC_API void orc_rng_skip(void *h, int n)
{
    index_t *ix = (index_t *)h;
    for (int i = 0; ix && i < n; i++) (void)rng_next_single(&ix->rng);
}

/* ------------------------------------------------------------------------------------
 * Removal: HNSWIndex.Remove (src/HNSWIndex/HNSWIndex.cs:83-102) ->
 * GraphConnector.RemoveNodeConnections (GraphConnector.cs:53-167), single-threaded (the region
 * locker, GraphLocker.cs, only serialises concurrent removals).
 * ---------------------------------------------------------------------------------- */
static int try_replace_entry_point(index_t *ix, int layer) /* GraphData.cs:146-167 */
{
    const edges_t *e = &ix->nodes[ix->entry].out[layer];
    if (e->count <= 0) return 0;
    int repl = -1, maxc = -1;
    for (int i = 0; i < e->count; i++) {
        int nb = e->buf[i];
        int cnt = ix->nodes[nb].out[layer].count;
        if (cnt > maxc) { maxc = cnt; repl = nb; }
    }
    ix->entry = repl;
    return 1;
}
static void force_replace_entry_point(index_t *ix) /* GraphData.cs:173-190 */
{
    if (ix->count == 0) return;
    int best_layer = -1, best_id = -1;
    for (int i = 0; i < ix->count; i++) {
        int id = ix->dense[i];
        if (ix->nodes[id].max_layer > best_layer) { best_layer = ix->nodes[id].max_layer; best_id = id; }
    }
    ix->entry = best_id;
}
static void replace_entry_point_if_needed(index_t *ix, int removed, int layer) /* GraphConnector.cs:72-85 */
{
    if (removed != ix->entry) return;
    if (try_replace_entry_point(ix, layer)) return;
    if (layer > 0) return;
    if (ix->count == 1) { ix->entry = -1; return; }
    force_replace_entry_point(ix);
}
static int in_list(const int *a, int n, int v)
{
    for (int i = 0; i < n; i++) if (a[i] == v) return 1;
    return 0;
}
/* pre / n_pre: the search result to use instead of searching now (the snapshot-batched schedule below), or NULL */
static void remove_connections_at_layer_pre(sctx_t *c, int removed, int layer, const nd_t *pre, int n_pre) /* GraphConnector.cs:90-167 */
{
    index_t *ix = c->ix;
    const int max_edges = max_edges_at(ix, layer);
    node_t *rn = &ix->nodes[removed];
    for (int i = 0; i < rn->out[layer].count; i++) /* DetachOutgoingReferences :277-288 */
        edges_remove(&ix->nodes[rn->out[layer].buf[i]].in[layer], removed);
    int n_aff = rn->in[layer].count; /* :95 */
    int *affected = (int *)malloc(sizeof(int) * (size_t)(n_aff > 0 ? n_aff : 1));
    memcpy(affected, rn->in[layer].buf, sizeof(int) * (size_t)n_aff);
    nd_t *sc;
    int n_sc;
    if (pre) {
        n_sc = n_pre;
        sc = (nd_t *)malloc(sizeof(nd_t) * (size_t)(n_pre > 0 ? n_pre : 1));
        memcpy(sc, pre, sizeof(nd_t) * (size_t)n_pre);
    } else n_sc = search_layer_f(c, removed, layer, ix->remove_max_candidates, item(ix, removed), &sc, removed); /* :96 */
    nd_t *cd = (nd_t *)malloc(sizeof(nd_t) * (size_t)(n_sc + max_edges + 2));
    int *old_ids = (int *)malloc(sizeof(int) * (size_t)(max_edges + 2));
    for (int a = 0; a < n_aff; a++) {
        const int aid = affected[a];
        node_t *an = &ix->nodes[aid];
        edges_remove(&an->out[layer], removed); /* RemoveOutEdge :104 */
        const int old_count = an->out[layer].count;
        memcpy(old_ids, an->out[layer].buf, sizeof(int) * (size_t)old_count); /* :110-111 */
        int cc = 0;
        for (int j = 0; j < old_count; j++) { cd[cc].id = old_ids[j]; cd[cc].dist = dist_ii(c, old_ids[j], aid); cc++; } /* :115-120 */
        for (int j = 0; j < n_sc; j++) { /* :123-129 */
            int cid = sc[j].id;
            if (cid == aid) continue;
            if (in_list(old_ids, old_count, cid)) continue;
            cd[cc].id = cid; cd[cc].dist = dist_ii(c, cid, aid); cc++;
        }
        edges_t nw = relative_neighbor_pruning(c, cd, cc, max_edges); /* :131 */
        for (int j = 0; j < old_count; j++) { /* :135-143 old neighbours no longer selected */
            int o = old_ids[j];
            if (in_list(nw.buf, nw.count, o)) continue;
            edges_remove(&an->out[layer], o);
            edges_remove(&ix->nodes[o].in[layer], aid);
        }
        for (int j = 0; j < nw.count; j++) { /* :146-164 newly selected neighbours */
            int w = nw.buf[j];
            if (in_list(old_ids, old_count, w)) continue;
            if (ix->nodes[w].is_removed) continue; /* :155 */
            edges_add(&an->out[layer], w);
            edges_add(&ix->nodes[w].in[layer], aid);
        }
        free(nw.buf);
    }
    free(cd); free(old_ids); free(sc); free(affected);
}

static void remove_connections_at_layer(sctx_t *c, int removed, int layer) { remove_connections_at_layer_pre(c, removed, layer, NULL, 0); }
static void retire_item(index_t *ix, int id) /* GraphData.RemoveItem :124-128, ActiveSet.Remove :85-97 */
{
    ix->removed_stack[ix->n_removed_stack++] = id;
    int idx = ix->sparse[id], last = --ix->count, last_id = ix->dense[last];
    ix->dense[idx] = last_id;
    ix->sparse[last_id] = idx;
}

/* Snapshot-batched removal -- NOT a reference code path: the deterministic counterpart of the reference's
 * Remove(List) = Parallel.For under region locks (HNSWIndex.cs:95-101, GraphLocker.cs:28-72), which lets removals
 * with disjoint neighbourhoods run concurrently in a scheduler-dependent order.  The builder's schedule
 * (hnsw_mi355x_set_remove_batch(B), B > 1):
 *   - the ids are taken in order; the current entry point is removed alone, by the sequential path;
 *   - otherwise a batch is formed from the first 8 B remaining ids: an id joins (up to B) if its region -- itself,
 *     its out-neighbours and its in-neighbours on every layer -- is disjoint from the regions already in the batch
 *     and it is not the entry point; the others stay, in order, for later batches;
 *   - every member is marked removed; every (member, layer) search runs on the graph as it stands before the
 *     batch; then the members are unlinked in order with those search results.
 * Disjoint regions make the unlinking steps independent of each other (no step reads a list another one writes),
 * so the outcome is one the reference's locking admits. */
ORC_API int orc_remove(void *h, const int *ids, int n);
ORC_API int orc_remove_batched(void *h, const int *ids_in, int n, int bmax)
{
    index_t *ix = (index_t *)h;
    if (!ix) return 0;
    if (!ix->allow_removals) return -1;
    if (bmax < 1) bmax = 1;
    for (int t = 0; t < n; t++) {
        const int id = ids_in[t];
        if (id < 0 || id >= ix->length || ix->nodes[id].is_removed || !ix->nodes[id].out) return -1;
        for (int u = 0; u < t; u++) if (ids_in[u] == id) return -1;
    }
    sctx_t c = {ix, &ix->vis, 0};
    int *rem = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)), *next = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int *batch = (int *)malloc(sizeof(int) * (size_t)bmax);
    unsigned char *marked = (unsigned char *)calloc((size_t)ix->length + 1, 1);
    memcpy(rem, ids_in, sizeof(int) * (size_t)n);
    int nrem = n;
    while (nrem > 0) {
        if (rem[0] == ix->entry) { /* alone, sequentially (the entry point moves) */
            int one = rem[0];
            ix->n_eval += c.n_eval; c.n_eval = 0;
            if (orc_remove(h, &one, 1) != 0) { free(rem); free(next); free(batch); free(marked); return -1; }
            memmove(rem, rem + 1, sizeof(int) * (size_t)(nrem - 1));
            nrem--;
            continue;
        }
        const int window = nrem < 8 * bmax ? nrem : 8 * bmax;
        int nb = 0, nn = 0;
        for (int t = 0; t < window; t++) {
            const int id = rem[t];
            const node_t *nd = &ix->nodes[id];
            int ok = id != ix->entry && nb < bmax && !marked[id];
            for (int layer = 0; ok && layer <= nd->max_layer; layer++) {
                for (int i = 0; ok && i < nd->out[layer].count; i++) ok = !marked[nd->out[layer].buf[i]];
                for (int i = 0; ok && i < nd->in[layer].count; i++) ok = !marked[nd->in[layer].buf[i]];
            }
            if (!ok) { next[nn++] = id; continue; }
            batch[nb++] = id;
            marked[id] = 1;
            for (int layer = 0; layer <= nd->max_layer; layer++) {
                for (int i = 0; i < nd->out[layer].count; i++) marked[nd->out[layer].buf[i]] = 1;
                for (int i = 0; i < nd->in[layer].count; i++) marked[nd->in[layer].buf[i]] = 1;
            }
        }
        for (int t = window; t < nrem; t++) next[nn++] = rem[t];
        /* clear the marks of this batch */
        for (int b = 0; b < nb; b++) {
            const node_t *nd = &ix->nodes[batch[b]];
            marked[batch[b]] = 0;
            for (int layer = 0; layer <= nd->max_layer; layer++) {
                for (int i = 0; i < nd->out[layer].count; i++) marked[nd->out[layer].buf[i]] = 0;
                for (int i = 0; i < nd->in[layer].count; i++) marked[nd->in[layer].buf[i]] = 0;
            }
        }
        /* searches on the snapshot */
        int nsteps = 0;
        for (int b = 0; b < nb; b++) nsteps += ix->nodes[batch[b]].max_layer + 1;
        nd_t **res = (nd_t **)malloc(sizeof(nd_t *) * (size_t)nsteps);
        int *nres = (int *)malloc(sizeof(int) * (size_t)nsteps);
        int sidx = 0;
        for (int b = 0; b < nb; b++) {
            const int id = batch[b];
            for (int layer = ix->nodes[id].max_layer; layer >= 0; layer--, sidx++)
                nres[sidx] = search_layer_f(&c, id, layer, ix->remove_max_candidates, item(ix, id), &res[sidx], id);
        }
        for (int b = 0; b < nb; b++) ix->nodes[batch[b]].is_removed = 1;
        sidx = 0;
        for (int b = 0; b < nb; b++) {
            const int id = batch[b];
            for (int layer = ix->nodes[id].max_layer; layer >= 0; layer--, sidx++) {
                remove_connections_at_layer_pre(&c, id, layer, res[sidx], nres[sidx]);
                free(res[sidx]);
                if (layer == 0) retire_item(ix, id);
            }
        }
        free(res); free(nres);
        int *tmp = rem; rem = next; next = tmp;
        nrem = nn;
    }
    free(rem); free(next); free(batch); free(marked);
    ix->n_eval += c.n_eval;
    return 0;
}

ORC_API int orc_remove(void *h, const int *ids, int n)
{
    index_t *ix = (index_t *)h;
    if (!ix) return 0;
    if (!ix->allow_removals) return -1; /* InvalidOperationException, HNSWIndex.cs:85-86 */
    sctx_t c = {ix, &ix->vis, 0};
    for (int t = 0; t < n; t++) {
        const int id = ids[t];
        if (id < 0 || id >= ix->length || ix->nodes[id].is_removed || !ix->nodes[id].out) return -1;
        ix->nodes[id].is_removed = 1; /* :55-57 */
        for (int layer = ix->nodes[id].max_layer; layer >= 0; layer--) { /* :59-66 */
            replace_entry_point_if_needed(ix, id, layer);
            remove_connections_at_layer(&c, id, layer);
            if (layer == 0) { /* GraphData.RemoveItem :124-128 */
                ix->removed_stack[ix->n_removed_stack++] = id;
                int idx = ix->sparse[id], last = --ix->count, last_id = ix->dense[last]; /* ActiveSet.Remove :85-97 */
                ix->dense[idx] = last_id;
                ix->sparse[last_id] = idx;
            }
        }
    }
    ix->n_eval += c.n_eval;
    return 0;
}

ORC_API int orc_active_ids(void *h, int *out, int cap)
{
    index_t *ix = (index_t *)h;
    int n = ix->count < cap ? ix->count : cap;
    memcpy(out, ix->dense, sizeof(int) * (size_t)n);
    return ix->count;
}
ORC_API int orc_length(void *h) { return ((index_t *)h)->length; }

/* ------------------------------------------------------------------------------------
 * C API (ctypes)
 * ---------------------------------------------------------------------------------- */
ORC_API void *orc_create(int dim, int metric, int max_edges, double dist_rate, int min_nn, int max_candidates,
                         int collection_size, int seed, int allow_removals, int use_avx)
{
    if (dim <= 0 || metric < 0 || metric > 3) return NULL;
    index_t *ix = (index_t *)calloc(1, sizeof(index_t));
    ix->udim = dim;
    if (metric == ORC_SQ_EUCLID_I8) dim = i8_pitch(dim);
    ix->dim = dim; ix->metric = metric; ix->max_edges = max_edges; ix->dist_rate = dist_rate;
    ix->min_nn = min_nn; ix->max_candidates = max_candidates; ix->seed = seed; ix->allow_removals = allow_removals;
    ix->capacity = collection_size > 0 ? collection_size : 1;
    ix->items = (float *)malloc(sizeof(float) * (size_t)ix->capacity * (size_t)dim);
    ix->nodes = (node_t *)calloc((size_t)ix->capacity, sizeof(node_t));
    ix->remove_max_candidates = 100; /* HNSWParameters.cs:37 */
    ix->dense = (int *)malloc(sizeof(int) * (size_t)ix->capacity);
    ix->sparse = (int *)malloc(sizeof(int) * (size_t)ix->capacity);
    ix->removed_stack = (int *)malloc(sizeof(int) * (size_t)ix->capacity);
    ix->entry = -1;
    /* RandomSeed < 0 => unseeded Random() (GraphData.cs:42): not reproducible by design;
     * the oracle seeds with |seed| so that runs stay deterministic. */
    rng_init(&ix->rng, seed < 0 ? -seed : seed);
    ix->dist = pick_metric(metric, use_avx);
    visited_init(&ix->vis, ix->capacity);
    return ix;
}

ORC_API void orc_free(void *h)
{
    index_t *ix = (index_t *)h;
    if (!ix) return;
    for (int i = 0; i < ix->length; i++) node_free_lists(&ix->nodes[i]);
    free(ix->nodes);
    free(ix->dense); free(ix->sparse); free(ix->removed_stack);
    free(ix->items);
    visited_free(&ix->vis);
    for (int t = 0; t < ix->n_vis_pool; t++) visited_free(&ix->vis_pool[t]);
    free(ix->vis_pool);
    free(ix->alog);
    free(ix);
}

/* Sequential, in input order: the deterministic schedule the reference's own determinism
 * test uses (bindings/__tests__/parameters_test.py:65-68, one add() per vector). */
ORC_API int orc_add(void *h, const float *v, int n, int *out_ids)
{
    index_t *ix = (index_t *)h;
    if (!ix || !v || n <= 0) return 0;
    float *tmp;
    v = incoming(ix, v, n, &tmp);
    sctx_t c = {ix, &ix->vis, 0};
    for (int i = 0; i < n; i++) {
        int id = add_one(&c, v + (size_t)i * (size_t)ix->dim);
        if (out_ids) out_ids[i] = id;
    }
    ix->n_eval += c.n_eval;
    free(tmp);
    return n;
}

/* src/HNSWIndex/HNSWIndex.cs:107-124 (layer 0, no filter) + export padding
 * bindings/HNSWIndex.Native/HNSWIndexExports.cs:135-145. */
static void knn_one(sctx_t *c, const float *q, int k, int *out_ids, float *out_d)
{
    index_t *ix = c->ix;
    int n = 0;
    nd_t *res = NULL;
    if (ix->count > 0 && k >= 1) {
        int ef = ix->min_nn > k ? ix->min_nn : k;     /* :115 */
        int ep = find_entry_point(c, 0, q);          /* :116 */
        n = search_layer(c, ep, 0, ef, q, &res);     /* :117 */
        /* LINQ OrderBy(c => c.Dist) (:121) is a STABLE sort, key order float.CompareTo:
         * insertion sort (stable) over the heap-order array; n <= ef. */
        for (int i = 1; i < n; i++) {
            nd_t t = res[i];
            int j = i - 1;
            while (j >= 0 && float_compare_to(t.dist, res[j].dist) < 0) { res[j + 1] = res[j]; j--; }
            res[j + 1] = t;
        }
    }
    int m = n < k ? n : k;
    for (int j = 0; j < m; j++) { out_ids[j] = res[j].id; out_d[j] = res[j].dist; }
    for (int j = m; j < k; j++) { out_ids[j] = -1; out_d[j] = NAN; }
    free(res);
}

typedef struct {
    index_t *ix; const float *q; int n, k; int *ids; float *d;
    volatile int *next; uint64_t n_eval;
} qjob_t;

static void *query_worker(void *arg)
{
    qjob_t *j = (qjob_t *)arg;
    visited_t vis;
    visited_init(&vis, j->ix->capacity);
    sctx_t c = {j->ix, &vis, 0};
    for (;;) {
        int i0 = __atomic_fetch_add(j->next, 16, __ATOMIC_RELAXED);
        if (i0 >= j->n) break;
        int i1 = i0 + 16 < j->n ? i0 + 16 : j->n;
        for (int i = i0; i < i1; i++)
            knn_one(&c, j->q + (size_t)i * (size_t)j->ix->dim, j->k, j->ids + (size_t)i * (size_t)j->k,
                    j->d + (size_t)i * (size_t)j->k);
    }
    j->n_eval = c.n_eval;
    visited_free(&vis);
    return NULL;
}

/* hnsw_knn_query -> BatchKnnQuery (Parallel.For over queries, HNSWIndex.cs:129-137).
 * threads <= 1: plain loop. */
ORC_API int orc_knn_query(void *h, const float *q, int n, int k, int *out_ids, float *out_d, int threads)
{
    index_t *ix = (index_t *)h;
    if (!ix) return 0;
    if (n <= 0 || k <= 0) return 0;
    float *tmp;
    q = incoming(ix, q, n, &tmp);
    if (threads <= 1) {
        sctx_t c = {ix, &ix->vis, 0};
        for (int i = 0; i < n; i++)
            knn_one(&c, q + (size_t)i * (size_t)ix->dim, k, out_ids + (size_t)i * (size_t)k, out_d + (size_t)i * (size_t)k);
        ix->n_eval += c.n_eval;
        free(tmp);
        return 0;
    }
    if (threads > 256) threads = 256;
    pthread_t th[256];
    qjob_t jobs[256];
    volatile int next = 0;
    for (int t = 0; t < threads; t++) {
        jobs[t] = (qjob_t){ix, q, n, k, out_ids, out_d, &next, 0};
        pthread_create(&th[t], NULL, query_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        ix->n_eval += jobs[t].n_eval;
    }
    free(tmp);
    return 0;
}

/* HNSWIndex.RangeQuery (src/HNSWIndex/HNSWIndex.cs:144-156): results of query i are written to
 * out_ids/out_d starting at out_off[i] (caller sizes the buffers with cap per query); returns
 * the total or -1 if cap is too small. */
ORC_API int orc_range_query(void *h, const float *q, int n, float range, int cap, int *out_cnt, int *out_ids, float *out_d)
{
    index_t *ix = (index_t *)h;
    if (!ix) return -1;
    float *tmp;
    q = incoming(ix, q, n, &tmp);
    sctx_t c = {ix, &ix->vis, 0};
    for (int i = 0; i < n; i++) {
        out_cnt[i] = 0;
        if (ix->count <= 0) continue;
        const float *qi = q + (size_t)i * (size_t)ix->dim;
        int ep = find_entry_point(&c, 0, qi);
        nd_t *res;
        int m = search_layer_range(&c, ep, 0, range, qi, &res);
        for (int a = 1; a < m; a++) { /* OrderBy(c => c.Dist): stable */
            nd_t t = res[a];
            int j = a - 1;
            while (j >= 0 && float_compare_to(t.dist, res[j].dist) < 0) { res[j + 1] = res[j]; j--; }
            res[j + 1] = t;
        }
        if (m > cap) { free(res); free(tmp); return -1; }
        for (int a = 0; a < m; a++) { out_ids[(size_t)i * cap + a] = res[a].id; out_d[(size_t)i * cap + a] = res[a].dist; }
        out_cnt[i] = m;
        free(res);
    }
    ix->n_eval += c.n_eval;
    free(tmp);
    return 0;
}

/* --- import of a graph built elsewhere (the product's), so that the oracle can (i) check
 * query parity on that very graph at full size and (ii) be timed on it as the CPU baseline.
 * Levels are imposed (no RNG draw); adjacency lists are copied verbatim, order included. --- */
ORC_API int orc_import_nodes(void *h, const float *items, const int *levels, int n, int entry)
{
    index_t *ix = (index_t *)h;
    if (!ix || ix->length != 0 || n <= 0) return -1;
    while (ix->capacity < n) grow(ix);
    if (ix->metric == ORC_SQ_EUCLID_I8) { /* quantised straight into the item array */
        for (int i = 0; i < n; i++) i8_quantize(items + (size_t)i * (size_t)ix->udim, ix->udim, ix->items + (size_t)i * (size_t)ix->dim, ix->dim);
    } else
    memcpy(ix->items, items, sizeof(float) * (size_t)n * (size_t)ix->dim);
    for (int i = 0; i < n; i++) { node_init(ix, &ix->nodes[i], levels[i]); ix->dense[i] = i; ix->sparse[i] = i; }
    ix->length = n;
    ix->count = n;
    ix->entry = entry;
    return n;
}
ORC_API int orc_import_edges(void *h, int layer, const int *counts, const int *edges, int stride, int n)
{
    index_t *ix = (index_t *)h;
    if (!ix || n > ix->length) return -1;
    for (int i = 0; i < n; i++) {
        if (counts[i] < 0) continue;
        if (layer > ix->nodes[i].max_layer) return -1;
        edges_t *e = &ix->nodes[i].out[layer];
        e->count = 0;
        for (int j = 0; j < counts[i]; j++) edges_add(e, edges[(size_t)i * (size_t)stride + (size_t)j]);
    }
    return n;
}

/* --- introspection for parity checks --- */
ORC_API int orc_count(void *h) { return ((index_t *)h)->count; }
ORC_API int orc_entry_point(void *h) { return ((index_t *)h)->entry; }
ORC_API int orc_capacity(void *h) { return ((index_t *)h)->capacity; }
ORC_API int orc_node_max_layer(void *h, int id)
{
    index_t *ix = (index_t *)h;
    return (id < 0 || id >= ix->length) ? -1 : ix->nodes[id].max_layer;
}
ORC_API int orc_get_edges(void *h, int id, int layer, int incoming, int *out, int cap)
{
    index_t *ix = (index_t *)h;
    if (id < 0 || id >= ix->length || layer < 0 || layer > ix->nodes[id].max_layer) return -1;
    if (incoming && !ix->nodes[id].in) return 0;
    const edges_t *e = incoming ? &ix->nodes[id].in[layer] : &ix->nodes[id].out[layer];
    int n = e->count < cap ? e->count : cap;
    memcpy(out, e->buf, sizeof(int) * (size_t)n);
    return e->count;
}
/* ------------------------------------------------------------------------------------
 * Pieces of the exact-window schedule, one call each, for tests/test_window_model.py: a CPU model of
 * hnsw_index.cpp::insert_exact_window (speculative searches on one snapshot, validation by read sets, the dry run of the
 * back-edge appends, "a change the reader does not see") whose result must be the sequential graph.  NOT reference code
 * paths: they only take the reference's steps apart.
 * ---------------------------------------------------------------------------------- */
/* GraphData.AddItem for n vectors (level draws in order), nothing linked yet. */
ORC_API int orc_alloc_only(void *h, const float *v, int n, int *out_ids)
{
    index_t *ix = (index_t *)h;
    float *tmp;
    v = incoming(ix, v, n, &tmp);
    for (int i = 0; i < n; i++) out_ids[i] = alloc_node(ix, v + (size_t)i * (size_t)ix->dim);
    free(tmp);
    return n;
}
/* ConnectNewNode for an allocated, unlinked node, exactly as add_one does after the allocation (used for items that move the
 * entry point, and as the sequential reference of the model). */
ORC_API void orc_connect_allocated(void *h, int id)
{
    index_t *ix = (index_t *)h;
    sctx_t c = {ix, &ix->vis, 0};
    if (ix->entry < 0) { ix->entry = id; return; }
    node_t *cur = &ix->nodes[id];
    int top = ix->nodes[ix->entry].max_layer;
    int new_ep = cur->max_layer > top;
    int best = find_entry_point(&c, cur->max_layer, item(ix, id));
    int start = cur->max_layer < top ? cur->max_layer : top;
    for (int layer = start; layer >= 0; --layer) best = connect_at_layer(&c, id, best, layer);
    if (new_ep) ix->entry = id;
    ix->n_eval += c.n_eval;
}
/* The search half of an insert on the graph as it stands (nothing is written): selections per layer into
 * sel[layer * stride ..] / cnt[layer]; returns min(level, top).  With the access log on, the reads (and the farthest
 * distance at every expansion) are recorded as for a sequential Add. */
ORC_API int orc_window_search(void *h, int id, int *sel, int *cnt, int stride)
{
    index_t *ix = (index_t *)h;
    sctx_t c = {ix, &ix->vis, 0};
    edges_t *s = (edges_t *)calloc((size_t)ix->nodes[id].max_layer + 1, sizeof(edges_t));
    alog_put(ix, 2, 0, id);
    batch_search(&c, id, s);
    int top = ix->nodes[ix->entry].max_layer, lvl = ix->nodes[id].max_layer;
    int start = lvl < top ? lvl : top;
    for (int l = 0; l <= start; l++) {
        cnt[l] = s[l].count;
        for (int e = 0; e < s[l].count && e < stride; e++) sel[l * stride + e] = s[l].buf[e];
        free(s[l].buf);
    }
    free(s);
    return start;
}
/* Dry run of one back-edge append: what would list (nb, layer) hold after `neighbor.OutEdges[layer].Add(item)` and, on
 * overflow, PruneOverflow (GraphConnector.cs:207-212)?  Returns 0 when it would read exactly as before, else 1 | 2 (the item
 * stays) and the ids it loses in lost[0 .. *n_lost) (cap 8; *n_lost = 255 beyond).  Nothing is written. */
ORC_API int orc_window_dry(void *h, int nb_id, int layer, int item_id, int *lost, int *n_lost)
{
    index_t *ix = (index_t *)h;
    sctx_t c = {ix, &ix->vis, 0};
    node_t *nb = &ix->nodes[nb_id];
    edges_t saved = edges_copy(&nb->out[layer]);
    edges_t saved_in = {0};
    int touch_in = ix->allow_removals;
    /* prune_overflow edits in-edge lists of dropped ids when removals are allowed: run it on copies of everything it touches */
    edges_t *in_copies = NULL;
    int n_in = 0;
    if (touch_in) {
        n_in = saved.count;
        in_copies = (edges_t *)malloc(sizeof(edges_t) * (size_t)(n_in > 0 ? n_in : 1));
        for (int i = 0; i < n_in; i++) in_copies[i] = edges_copy(&ix->nodes[saved.buf[i]].in[layer]);
    }
    (void)saved_in;
    edges_add(&nb->out[layer], item_id);
    if (nb->out[layer].count > max_edges_at(ix, layer)) prune_overflow(&c, nb_id, layer);
    const edges_t *now = &nb->out[layer];
    int same = now->count == saved.count;
    for (int i = 0; same && i < now->count; i++) same = now->buf[i] == saved.buf[i];
    int code = 0, nl = 0;
    if (!same) {
        code = 1;
        for (int i = 0; i < now->count; i++) if (now->buf[i] == item_id) code |= 2;
        for (int i = 0; i < saved.count; i++) {
            int keep = 0;
            for (int u = 0; u < now->count; u++) keep |= now->buf[u] == saved.buf[i];
            if (!keep) { if (nl < 8) lost[nl] = saved.buf[i]; nl++; }
        }
        if (nl > 8) nl = 255;
    }
    *n_lost = nl;
    /* restore */
    free(nb->out[layer].buf);
    nb->out[layer] = saved;
    if (touch_in) {
        for (int i = 0; i < n_in; i++) { free(ix->nodes[saved.buf[i]].in[layer].buf); ix->nodes[saved.buf[i]].in[layer] = in_copies[i]; }
        free(in_copies);
    }
    return code;
}
/* The link half of an insert with the selections a search on an OLDER graph returned. */
ORC_API void orc_window_link(void *h, int id, const int *sel, const int *cnt, int stride)
{
    index_t *ix = (index_t *)h;
    sctx_t c = {ix, &ix->vis, 0};
    int top = ix->nodes[ix->entry].max_layer, lvl = ix->nodes[id].max_layer;
    int start = lvl < top ? lvl : top;
    edges_t *s = (edges_t *)calloc((size_t)lvl + 1, sizeof(edges_t));
    for (int l = 0; l <= start; l++) {
        s[l] = edges_new(max_edges_at(ix, l) + 1);
        for (int e = 0; e < cnt[l]; e++) edges_add(&s[l], sel[l * stride + e]);
    }
    batch_link(&c, id, s);
    free(s);
    ix->n_eval += c.n_eval;
}
ORC_API float orc_dist_ids(void *h, int a, int b)
{
    index_t *ix = (index_t *)h;
    return ix->dist(item(ix, a), item(ix, b), ix->dim);
}

/* access log (see index_t.alog): cap entries are recorded from now on; 0 turns it off */
ORC_API void orc_access_log(void *h, long long cap)
{
    index_t *ix = (index_t *)h;
    free(ix->alog);
    ix->alog = cap > 0 ? (int64_t *)malloc(sizeof(int64_t) * (size_t)cap) : NULL;
    ix->alog_n = 0;
    ix->alog_cap = ix->alog ? (size_t)cap : 0;
}
ORC_API long long orc_access_log_fetch(void *h, int64_t *out, long long cap)
{
    index_t *ix = (index_t *)h;
    long long n = (long long)ix->alog_n < cap ? (long long)ix->alog_n : cap;
    if (out && n > 0) memcpy(out, ix->alog, sizeof(int64_t) * (size_t)n);
    return (long long)ix->alog_n;
}
ORC_API uint64_t orc_n_eval(void *h) { return ((index_t *)h)->n_eval; }
ORC_API void orc_reset_n_eval(void *h) { ((index_t *)h)->n_eval = 0; }
ORC_API void orc_set_remove_max_candidates(void *h, int v) { ((index_t *)h)->remove_max_candidates = v; }
ORC_API const float *orc_items(void *h) { return ((index_t *)h)->items; }

/* FNV-1a over (max_layer, per-layer out-edge lists) of every node: one number that pins
 * the whole graph (levels + adjacency + adjacency order). */
ORC_API uint64_t orc_graph_hash(void *h)
{
    index_t *ix = (index_t *)h;
    uint64_t x = 1469598103934665603ULL;
#define MIX(v) do { uint32_t _v = (uint32_t)(v); for (int _b = 0; _b < 4; _b++) { x ^= (_v >> (8 * _b)) & 0xff; x *= 1099511628211ULL; } } while (0)
    MIX(ix->entry);
    for (int i = 0; i < ix->length; i++) {
        node_t *nd = &ix->nodes[i];
        if (nd->is_removed) { MIX(-2); continue; } /* a removed slot: its stale lists are unreachable */
        MIX(nd->max_layer);
        for (int l = 0; l <= nd->max_layer; l++) {
            MIX(nd->out[l].count);
            for (int e = 0; e < nd->out[l].count; e++) MIX(nd->out[l].buf[e]);
        }
    }
#undef MIX
    return x;
}

/* --- unit pieces --- */
ORC_API float orc_metric(int metric, const float *a, const float *b, int n, int use_avx)
{
    if (metric == ORC_SQ_EUCLID_I8) { /* a, b: n floats each, quantised here */
        int pitch = i8_pitch(n);
        float *ra = (float *)malloc(sizeof(float) * 2 * (size_t)pitch), *rb = ra + pitch;
        i8_quantize(a, n, ra, pitch);
        i8_quantize(b, n, rb, pitch);
        float d = pick_metric(metric, use_avx)(ra, rb, pitch);
        free(ra);
        return d;
    }
    return pick_metric(metric, use_avx)(a, b, n);
}
/* the int8 record of one vector (pitch words, see the layout above) and the record size for a dimension */
ORC_API int orc_i8_pitch(int dim) { return i8_pitch(dim); }
ORC_API void orc_i8_quantize(const float *x, int dim, int32_t *record) { i8_quantize(x, dim, (float *)record, i8_pitch(dim)); }
ORC_API int orc_has_avx2(void) { return cpu_has_avx2_fma(); }

/* one query vs many rows: out[i] = metric(rows[ids[i]], q) -- the checker for the HIP
 * gather-distance kernel */
ORC_API void orc_dist_query_rows(int metric, const float *rows, int dim, const float *q, const int *ids, int n,
                                 float *out, int use_avx)
{
    if (metric == ORC_SQ_EUCLID_I8) {
        for (int i = 0; i < n; i++) out[i] = orc_metric(metric, rows + (size_t)ids[i] * (size_t)dim, q, dim, 0);
        return;
    }
    metric_fn f = pick_metric(metric, use_avx);
    for (int i = 0; i < n; i++) out[i] = f(rows + (size_t)ids[i] * (size_t)dim, q, dim);
}
ORC_API void orc_dist_pairs(int metric, const float *rows, int dim, const int *a, const int *b, int n, float *out,
                            int use_avx)
{
    if (metric == ORC_SQ_EUCLID_I8) {
        for (int i = 0; i < n; i++) out[i] = orc_metric(metric, rows + (size_t)a[i] * (size_t)dim, rows + (size_t)b[i] * (size_t)dim, dim, 0);
        return;
    }
    metric_fn f = pick_metric(metric, use_avx);
    for (int i = 0; i < n; i++) out[i] = f(rows + (size_t)a[i] * (size_t)dim, rows + (size_t)b[i] * (size_t)dim, dim);
}

ORC_API void orc_random_next(int seed, int n, int *out)
{
    dotnet_rng r;
    rng_init(&r, seed);
    for (int i = 0; i < n; i++) out[i] = rng_internal_sample(&r);
}
ORC_API void orc_random_next_double(int seed, int n, double *out)
{
    dotnet_rng r;
    rng_init(&r, seed);
    for (int i = 0; i < n; i++) out[i] = rng_sample(&r);
}
ORC_API void orc_random_next_single(int seed, int n, float *out)
{
    dotnet_rng r;
    rng_init(&r, seed);
    for (int i = 0; i < n; i++) out[i] = rng_next_single(&r);
}
/* The redraw rule on an injected stream of InternalSample() values: returns the NextSingle()
 * result and how many samples it consumed (-1.0f when the stream ran out). */
ORC_API float orc_next_single_from_samples(const int *samples, int n, int *used)
{
    for (int i = 0; i < n; i++) {
        float f = single_of_sample(samples[i]);
        if (f < 1.0f) { *used = i + 1; return f; }
    }
    *used = n;
    return -1.0f;
}
ORC_API void orc_random_levels(int seed, double rate, int n, int *out)
{
    dotnet_rng r;
    rng_init(&r, seed);
    for (int i = 0; i < n; i++) out[i] = level_from_uniform(rng_next_single(&r), rate);
}
ORC_API void orc_sort_nd(int *ids, float *dists, int n)
{
    nd_t *k = (nd_t *)malloc(sizeof(nd_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) { k[i].id = ids[i]; k[i].dist = dists[i]; }
    dotnet_sort_nd(k, n);
    for (int i = 0; i < n; i++) { ids[i] = k[i].id; dists[i] = k[i].dist; }
    free(k);
}
/* Replays a push/pop script on a BinaryHeap: ops[i] >= 0 pushes (id=ops[i], dist=d[i]);
 * ops[i] < 0 pops.  Writes the final buffer prefix; returns its length. */
ORC_API int orc_heap_script(int closer_first, const int *ops, const float *d, int n, int *out_ids, float *out_d,
                            int *popped_ids, int *n_popped)
{
    heap_t h;
    heap_init(&h, 4, closer_first);
    int np = 0;
    for (int i = 0; i < n; i++) {
        if (ops[i] >= 0) { nd_t v = {ops[i], d[i]}; heap_push(&h, v); }
        else if (h.count > 0) { nd_t v = heap_pop(&h); if (popped_ids) popped_ids[np] = v.id; np++; }
    }
    for (int i = 0; i < h.count; i++) { out_ids[i] = h.buf[i].id; out_d[i] = h.buf[i].dist; }
    if (n_popped) *n_popped = np;
    int c = h.count;
    heap_free(&h);
    return c;
}

/* SearchLayer on the built graph, result in heap order (what ConnectAtLayer consumes). */
ORC_API int orc_search_layer(void *h, int entry_id, int layer, int k, const float *q, int *out_ids, float *out_d)
{
    index_t *ix = (index_t *)h;
    sctx_t c = {ix, &ix->vis, 0};
    nd_t *res;
    float *tmp;
    q = incoming(ix, q, 1, &tmp);
    int n = search_layer(&c, entry_id, layer, k, q, &res);
    for (int i = 0; i < n; i++) { out_ids[i] = res[i].id; out_d[i] = res[i].dist; }
    free(res);
    free(tmp);
    ix->n_eval += c.n_eval;
    return n;
}
ORC_API int orc_find_entry_point(void *h, int dst_layer, const float *q)
{
    index_t *ix = (index_t *)h;
    sctx_t c = {ix, &ix->vis, 0};
    float *tmp;
    q = incoming(ix, q, 1, &tmp);
    int r = find_entry_point(&c, dst_layer, q);
    free(tmp);
    ix->n_eval += c.n_eval;
    return r;
}
