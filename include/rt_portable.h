/*
 * rt_portable.h — the two pieces of arithmetic that host and device must share
 * verbatim so that a photon's path is the same bits on a CPU and on a GPU:
 *
 *   1. a counter-based random stream (Philox4x32-10, Salmon et al., SC'11) keyed
 *      per light and indexed per (photon, stage, draw), replacing the reference's
 *      one sequential std::mt19937 per worker (include/RandomGenerator.h:6-22),
 *      which cannot be replayed in parallel;
 *   2. sin / cos / acos built only from + - * / sqrt, because glibc's and the
 *      device library's transcendental functions differ in the last bit.  The
 *      reference calls std::sin/cos/acos in src/Vector.cpp:319-329,
 *      src/AngleGenerator.cpp:24-36, src/AreaLight.cpp:126-146,
 *      src/LambertianMaterial.cpp:28-36, src/MicrofacetMaterial.cpp:107-111.
 *      The polynomials are the classic fdlibm kernels (k_sin.c, k_cos.c, e_acos.c,
 *      Sun Microsystems 1993, freely redistributable); results are within ~1 ulp
 *      of the correctly rounded value on the argument ranges the path uses
 *      (|x| <= 2*pi for sin/cos).
 *
 * Every function is `static inline`, uses no fused multiply-add (compile with
 * -ffp-contract=off), and is usable from C, C++ and HIP device code.
 */
#ifndef RT_PORTABLE_H
#define RT_PORTABLE_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define RT_SQRT(x) __builtin_sqrt(x)
#else
#include <math.h>
#define RT_SQRT(x) sqrt(x)
#endif

/* ------------------------------------------------------------------ Philox */

typedef struct rt_philox_state
{
    uint32_t key0, key1; /* (seed, light id) */
    uint32_t c0, c1;     /* photon index, low / high word */
    uint32_t stage;      /* RT_STAGE_* or RT_STAGE_SCATTER + bounce */
    uint32_t next;       /* index of the next draw in this stage */
    uint32_t have;       /* 1 if `spare` holds the odd draw of the current block */
    double spare;
} rt_philox_state;

enum
{
    RT_STAGE_EMIT = 0,   /* Light::emit draws for this photon */
    RT_STAGE_TIME = 1,   /* shutter time stamp (src/Worker.cpp:486-498) */
    RT_STAGE_SCATTER = 2 /* + bounce index: Material::sample draws at that bounce */
};

RT_HD static inline void rt_philox4x32_10(uint32_t ctr[4], uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int round = 0; round < 10; ++round)
    {
        const uint64_t p0 = (uint64_t)M0 * ctr[0];
        const uint64_t p1 = (uint64_t)M1 * ctr[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ ctr[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ ctr[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        ctr[0] = n0;
        ctr[1] = n1;
        ctr[2] = n2;
        ctr[3] = n3;
        k0 += W0;
        k1 += W1;
    }
}

/* 53-bit uniform in [0, 1) from two 32-bit words. */
RT_HD static inline double rt_u53(uint32_t hi, uint32_t lo)
{
    const uint64_t bits = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6);
    return (double)bits * (1.0 / 9007199254740992.0);
}

RT_HD static inline void rt_philox_begin(rt_philox_state* s, uint32_t seed, uint32_t light,
                                         uint64_t photon, uint32_t stage)
{
    s->key0 = seed;
    s->key1 = light;
    s->c0 = (uint32_t)photon;
    s->c1 = (uint32_t)(photon >> 32);
    s->stage = stage;
    s->next = 0;
    s->have = 0;
    s->spare = 0.0;
}

/* Draw k of a stage is word pair (k & 1) of block (k >> 1). */
RT_HD static inline double rt_philox_next(rt_philox_state* s)
{
    if (s->have)
    {
        s->have = 0;
        s->next += 1;
        return s->spare;
    }
    uint32_t ctr[4] = {s->c0, s->c1, s->stage, s->next >> 1};
    rt_philox4x32_10(ctr, s->key0, s->key1);
    s->spare = rt_u53(ctr[2], ctr[3]);
    s->have = 1;
    s->next += 1;
    return rt_u53(ctr[0], ctr[1]);
}

/* --------------------------------------------------------- portable libm */

RT_HD static inline double rt_from_bits(uint64_t b)
{
    double d;
    memcpy(&d, &b, sizeof d);
    return d;
}

RT_HD static inline uint64_t rt_to_bits(double d)
{
    uint64_t b;
    memcpy(&b, &d, sizeof b);
    return b;
}

/* sin on [-pi/4, pi/4] with a tail term (fdlibm __kernel_sin, iy = 1). */
RT_HD static inline double rt_ksin(double x, double y)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    const double v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

/* cos on [-pi/4, pi/4] with a tail term (fdlibm __kernel_cos, modern form). */
RT_HD static inline double rt_kcos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double w = z * z;
    const double r = z * (C1 + z * (C2 + z * C3)) + w * w * (C4 + z * (C5 + z * C6));
    const double hz = 0.5 * z;
    const double ww = 1.0 - hz;
    return ww + (((1.0 - ww) - hz) + (z * r - x * y));
}

/* Reduce x to y0 + y1 in [-pi/4, pi/4], return the quadrant.  Two-part Cody-Waite
 * with fdlibm's constants; exact enough for |x| up to a few thousand. */
RT_HD static inline int rt_rem_pio2(double x, double* y0, double* y1)
{
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double pio2_1t = 6.07710050650619224932e-11; /* pi/2 - pio2_1 */
    const double pio2_2 = 6.07710050630396597660e-11;  /* second 33 bits */
    const double pio2_2t = 2.02226624879595063154e-21; /* pi/2 - (pio2_1 + pio2_2) */
    const double t = x * invpio2;
    const double fn = (double)(long long)(t < 0.0 ? t - 0.5 : t + 0.5);
    const int n = (int)fn;
    double r = x - fn * pio2_1;
    double w = fn * pio2_1t;
    double y = r - w;
    /* second iteration when the first cancels too many bits */
    const int ex = (int)((rt_to_bits(x) >> 52) & 0x7ff);
    const int ey = (int)((rt_to_bits(y) >> 52) & 0x7ff);
    if (ex - ey > 16)
    {
        const double t2 = r;
        w = fn * pio2_2;
        r = t2 - w;
        w = fn * pio2_2t - ((t2 - r) - w);
        y = r - w;
    }
    *y0 = y;
    *y1 = (r - y) - w;
    return n;
}

RT_HD static inline void rt_sincos(double x, double* s, double* c)
{
    double y0, y1;
    int n = 0;
    if (x > -0.78539816339744830962 && x < 0.78539816339744830962)
    {
        y0 = x;
        y1 = 0.0;
    }
    else
    {
        n = rt_rem_pio2(x, &y0, &y1);
    }
    const double ks = rt_ksin(y0, y1);
    const double kc = rt_kcos(y0, y1);
    switch (n & 3)
    {
        case 0: *s = ks; *c = kc; break;
        case 1: *s = kc; *c = 0.0 - ks; break;
        case 2: *s = 0.0 - ks; *c = 0.0 - kc; break;
        default: *s = 0.0 - kc; *c = ks; break;
    }
}

RT_HD static inline double rt_sin(double x)
{
    double s, c;
    rt_sincos(x, &s, &c);
    return s;
}

RT_HD static inline double rt_cos(double x)
{
    double s, c;
    rt_sincos(x, &s, &c);
    return c;
}

/* acos on [-1, 1] (fdlibm e_acos.c). */
RT_HD static inline double rt_acos(double x)
{
    const double pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00,
                 pio2_lo = 6.12323399573676603587e-17;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const double ax = x < 0.0 ? 0.0 - x : x;
    if (ax >= 1.0)
    {
        if (x == 1.0)
        {
            return 0.0;
        }
        if (x == -1.0)
        {
            return pi + 2.0 * pio2_lo;
        }
        return (x - x) / (x - x); /* NaN outside the domain */
    }
    if (ax < 0.5)
    {
        if (ax <= 5.55111512312578270212e-17) /* 2^-54 */
        {
            return pio2_hi + pio2_lo;
        }
        const double z = x * x;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (x < 0.0)
    {
        const double z = (1.0 + x) * 0.5;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double s = RT_SQRT(z);
        const double r = p / q;
        const double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    {
        const double z = (1.0 - x) * 0.5;
        const double s = RT_SQRT(z);
        const double df = rt_from_bits(rt_to_bits(s) & 0xffffffff00000000ull);
        const double c = (z - df * df) / (s + df);
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        const double w = r * s + c;
        return 2.0 * (df + w);
    }
}

#endif /* RT_PORTABLE_H */
