// occ_rng.hpp -- device-side variate generators (gfx950).
//
// Counter-based streams: Philox4x32-10, key = 64-bit chain key, counter = (c0, c1, iteration, stream).
// One block = 128 random bits = two 64-bit words; u01(w) = ((w >> 12) + 1/2) 2^-52 in (0,1);
// a normal is Box-Muller (cosine branch) of the two words of ONE block.  Sequential consumers
// (PG, gamma) walk the sub-stream (index, iteration, stream) with a cursor: words come one at a
// time, c1 = block number; a normal always opens a fresh block.  DESIGN.md "Variate streams" is
// the specification; the CPU oracle implements the same specification independently.
//
// PG(1, z) stands where the reference calls polyagamma.random_polyagamma(1, b, ...)
// (occuspytial/gibbs/logit.py:191-193, 202-204): Devroye's exact sampler for J*(1, z/2)/4
// (Polson, Scott & Windle 2013, truncation point 0.64).  tau's gamma draw (logit.py:209) is
// Marsaglia-Tsang.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace occ {

enum : uint32_t {
    STREAM_OMEGA_B = 1,
    STREAM_TAU = 2,
    STREAM_ETA_SITE = 3,
    STREAM_ETA_EDGE = 4,
    STREAM_BETA = 5,
    STREAM_OMEGA_A = 6,
    STREAM_ALPHA = 7,
    STREAM_Z = 8
};

constexpr double kPi = 3.14159265358979323846;
constexpr double kSqrtHalf = 0.70710678118654752440;

struct Words {
    uint64_t w0, w1;
};

__device__ __forceinline__ Words philox(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3)
{
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Words w;
    w.w0 = ((uint64_t)c1 << 32) | c0;
    w.w1 = ((uint64_t)c3 << 32) | c2;
    return w;
}

__device__ __forceinline__ double u01(uint64_t w) { return ((double)(w >> 12) + 0.5) * 0x1.0p-52; }

__device__ __forceinline__ double box_muller(const Words &w)
{
    const double u1 = u01(w.w0), u2 = u01(w.w1);
    return sqrt(-2.0 * log(u1)) * cos(2.0 * kPi * u2);
}

__device__ __forceinline__ double block_normal(uint64_t key, uint32_t c0, uint32_t c1, uint32_t it, uint32_t stream)
{
    return box_muller(philox(key, c0, c1, it, stream));
}
__device__ __forceinline__ double block_uniform(uint64_t key, uint32_t c0, uint32_t c1, uint32_t it, uint32_t stream)
{
    return u01(philox(key, c0, c1, it, stream).w0);
}

struct Cursor {
    uint64_t key, cached;
    uint32_t index, it, stream, sub;
    bool have;
    __device__ __forceinline__ Cursor(uint64_t k, uint32_t idx, uint32_t iter, uint32_t s)
        : key(k), cached(0), index(idx), it(iter), stream(s), sub(0), have(false) {}
    __device__ __forceinline__ uint64_t word()
    {
        if (have) { have = false; return cached; }
        const Words w = philox(key, index, sub++, it, stream);
        cached = w.w1;
        have = true;
        return w.w0;
    }
    __device__ __forceinline__ double unif() { return u01(word()); }
    __device__ __forceinline__ double expo() { return -log(u01(word())); }
    __device__ __forceinline__ double norm()
    {
        have = false;
        return box_muller(philox(key, index, sub++, it, stream));
    }
};

// ---- PG(1, z) -------------------------------------------------------------------------------
constexpr double kPgT = 0.64;

__device__ __forceinline__ double pg_a(int n, double x)
{
    const double K = (n + 0.5) * kPi;
    if (x > kPgT) return K * exp(-0.5 * K * K * x);
    const double e = -1.5 * (log(0.5 * kPi) + log(x)) + log(K) - 2.0 * (n + 0.5) * (n + 0.5) / x;
    return exp(e);
}
__device__ __forceinline__ double log_phi(double x) { return log(0.5 * erfc(-x * kSqrtHalf)); }

__device__ __forceinline__ double pg_mass_texpon(double Z)
{
    const double fz = 0.125 * kPi * kPi + 0.5 * Z * Z;
    const double b = sqrt(1.0 / kPgT) * (kPgT * Z - 1.0);
    const double a = -sqrt(1.0 / kPgT) * (kPgT * Z + 1.0);
    const double x0 = log(fz) + fz * kPgT;
    const double xb = x0 - Z + log_phi(b);
    const double xa = x0 + Z + log_phi(a);
    const double qdivp = 4.0 / kPi * (exp(xb) + exp(xa));
    return 1.0 / (1.0 + qdivp);
}

__device__ inline double pg_rtigauss(Cursor &c, double Z)
{
    double X = kPgT + 1.0;
    if (1.0 / kPgT > Z) {
        double alpha = 0.0, U = 1.0;
        while (U > alpha) {
            double E1 = c.expo(), E2 = c.expo();
            while (E1 * E1 > 2.0 * E2 / kPgT) { E1 = c.expo(); E2 = c.expo(); }
            X = 1.0 + E1 * kPgT;
            X = kPgT / (X * X);
            alpha = exp(-0.5 * Z * Z * X);
            U = c.unif();
        }
    } else {
        const double mu = 1.0 / Z;
        while (X > kPgT) {
            double Y = c.norm();
            Y *= Y;
            const double half_mu = 0.5 * mu, mu_Y = mu * Y;
            X = mu + half_mu * mu_Y - half_mu * sqrt(4.0 * mu_Y + mu_Y * mu_Y);
            if (c.unif() > mu / (mu + X)) X = mu * mu / X;
        }
    }
    return X;
}

__device__ inline double pg1_draw(Cursor &c, double z)
{
    const double Z = 0.5 * fabs(z);
    const double fz = 0.125 * kPi * kPi + 0.5 * Z * Z;
    const double ptail = pg_mass_texpon(Z);
    for (;;) {
        double X;
        if (c.unif() < ptail) X = kPgT + c.expo() / fz;
        else X = pg_rtigauss(c, Z);
        double S = pg_a(0, X);
        const double Y = c.unif() * S;
        int n = 0;
        for (;;) {
            ++n;
            if (n & 1) {
                S -= pg_a(n, X);
                if (Y <= S) return 0.25 * X;
            } else {
                S += pg_a(n, X);
                if (Y > S) break;
            }
        }
    }
}

// ---- standard gamma (Marsaglia & Tsang 2000) ---------------------------------------------------
__device__ inline double std_gamma(Cursor &c, double shape)
{
    double boost = 1.0, a = shape;
    if (a < 1.0) {
        boost = pow(c.unif(), 1.0 / a);
        a += 1.0;
    }
    const double d = a - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * d);
    for (;;) {
        const double x = c.norm();
        double v = 1.0 + cc * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        const double u = c.unif();
        if (u < 1.0 - 0.0331 * (x * x) * (x * x)) return boost * d * v;
        if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return boost * d * v;
    }
}

}  // namespace occ
