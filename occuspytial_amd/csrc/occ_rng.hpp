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
    STREAM_Z = 8,
    STREAM_ETA_DENSE = 10 // the standard normals of the reference-form prior draw: c0 = column of the factor
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
        // (ONE 32 x 32 -> 64 multiply per product -- v_mad_u64_u32 -- where __umulhi and * made two: integer multiplies
        // run at a quarter of the rate, and the 40 per block were ~40 % of a Polya-Gamma draw's issue cycles)
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
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
// Devroye's alternating-series sampler for J*(1, z/2)/4 with truncation point t = 0.64, as ONE rejection loop (round 4;
// the specification and its derivation: oracle/occ_oracle.c "PG(1, z)", DESIGN.md "Variate streams").  Round r of a draw
// takes Philox block r of the sub-stream (index, iteration, stream), its four 32-bit words (x0, x1, x2, x3): Ux = u01 of
// the 64-bit word x1:x0 (52 bits: the proposal's variate), Um = (x2 + 1/2) 2^-32 picks the piece of the envelope, Us =
// (x3 + 1/2) 2^-32 decides -- probabilities to within 2^-33 -- and U2 = (Um - ptail) / (1 - ptail), uniform given that
// the left piece was picked, picks the Michael-Schucany-Haas root.  (Ten Philox rounds are 15 quarter-rate 32 x 32 -> 64
// multiplies and as many full-rate instructions again: with two blocks per round they were a third of a round's issue
// cycles.)
//   * The probability of the right piece is 1 / (1 + k f exp(f t - s)): the left piece's envelope is the Levy density
//     itself below Z = 1/t (its tilt exp(-Z^2 x/2) goes into the acceptance test) and the untruncated IG(1/Z, 1) from
//     1/t on (a proposal beyond t is a rejected round), so cosh Z cancels and no erfc is needed (rounds 1-3: two erfc and
//     three exp per draw, and a rejection loop of its own for the truncated inverse Gaussian).
//   * The alternating series is tested in ratio form, U <= E (1 - r_1 + r_2 - ...), r_n = a_n / a_0 = (2n+1) exp(...):
//     a_0 is never formed, and the first test -- one exp -- decides 99.4 % of the proposals.
//   * A wave executes every branch some lane takes, so the round is written WITHOUT branches: both proposals, both
//     quantile formulas and both left-piece forms are evaluated by every lane and selected -- their dependent chains then
//     run side by side (a Polya-Gamma wave alone on its SIMD is bound by the latency of its dependent chain, not by issue:
//     tools/pg_occupancy.py), and they share one logarithm (the right piece's exponential variate and the quantile's
//     -log p = -(log Ux + log c)).
constexpr double kPgT = 0.64;
constexpr double kPgPLevy = 0.10564977366685526;      // Phi(-1/sqrt(t)) = Phi(-1.25)
constexpr double kPgLogPLevy = -2.2476256772143182;   // its logarithm
constexpr double kPgLogHalf = -0.69314718055994531;
constexpr double kPgKBelow = 0.26903493944991954;     // 8 Phi(-1.25) / pi
constexpr double kPgKAbove = 1.2732395447351628;      // 4 / pi

// 1 / x to about an ulp: v_rcp_f64 and two Newton steps (5 instructions where the correctly rounded division takes 11)
__device__ __forceinline__ double pg_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
__device__ __forceinline__ double u32_01(uint32_t w) { return ((double)w + 0.5) * 0x1.0p-32; }

__device__ __forceinline__ double poly8(double x, double k0, double k1, double k2, double k3, double k4, double k5,
                                        double k6, double k7)
{
    return ((((((k7 * x + k6) * x + k5) * x + k4) * x + k3) * x + k2) * x + k1) * x + k0;
}
// -Phi^-1(p) for p in (0, 1/2]: Wichura (1988), algorithm AS 241 (PPND16, relative accuracy ~1e-16), with -log p supplied.
// The central and the tail formula are both evaluated (a wave meets both); the far tail (p < e^-25) is a real branch.
__device__ __forceinline__ double pg_neg_quantile(double p, double neg_log_p)
{
    const double q = p - 0.5;
    const double rc = 0.180625 - q * q;
    const double central = -q * pg_rcp(poly8(rc, 1.0, 4.2313330701600911252e1, 6.8718700749205790830e2, 5.3941960214247511077e3,
                                             2.1213794301586595867e4, 3.9307895800092710610e4, 2.8729085735721942674e4, 5.2264952788528545610e3)) *
                           poly8(rc, 3.3871328727963666080, 1.3314166789178437745e2, 1.9715909503065514427e3, 1.3731693765509461125e4,
                                 4.5921953931549871457e4, 6.7265770927008700853e4, 3.3430575583588128105e4, 2.5090809287301226727e3);
    const double r = sqrt(neg_log_p);
    const double rt = r - 1.6;
    double tail = poly8(rt, 1.42343711074968357734, 4.63033784615654529590, 5.76949722146069140550, 3.64784832476320460504,
                        1.27045825245236838258, 2.41780725177450611770e-1, 2.27238449892691845833e-2, 7.74545014278341407640e-4) *
                  pg_rcp(poly8(rt, 1.0, 2.05319162663775882187, 1.67638483018380384940, 6.89767334985100004550e-1,
                               1.48103976427480074590e-1, 1.51986665636164571966e-2, 5.47593808499534494600e-4, 1.05075007164441684324e-9));
    if (r > 5.0) {
        const double rf = r - 5.0;
        tail = poly8(rf, 6.65790464350110377720, 5.46378491116411436990, 1.78482653991729133580, 2.96560571828504891230e-1,
                     2.65321895265761230930e-2, 1.24266094738807843860e-3, 2.71155556874348757815e-5, 2.01033439929228813265e-7) /
               poly8(rf, 1.0, 5.99832206555887937690e-1, 1.36929880922735805310e-1, 1.48753612908506148525e-2,
                     7.86869131145613259100e-4, 1.84631831751005468180e-5, 1.42151175831644588870e-7, 2.04426310338993978564e-15);
    }
    return (q >= -0.425) ? central : tail;
}

// What a round needs of the draw's argument (formed once per draw)
struct PgPrep {
    double ptail, rq, rfz, mu, hzz;  // P(right piece), 1 / (1 - ptail), 1 / f, 1 / Z (from 1/t on; 1 below), Z^2 / 2 (below 1/t; 0 from there on)
    int below;                       // Z < 1 / t
};
__device__ __forceinline__ PgPrep pg_prep(double Z)
{
    PgPrep P;
    const double fz = 0.125 * kPi * kPi + 0.5 * Z * Z;
    const bool below = Z < 1.0 / kPgT;
    P.below = below ? 1 : 0;
    P.ptail = pg_rcp(1.0 + (below ? kPgKBelow : kPgKAbove) * fz * exp(fz * kPgT - (below ? 0.0 : Z)));
    P.rfz = pg_rcp(fz);
    P.mu = pg_rcp(below ? 1.0 : Z);
    P.hzz = below ? 0.5 * Z * Z : 0.0;
    P.rq = pg_rcp(1.0 - P.ptail);
    return P;
}
// Round r of the draw (index, iteration, stream), in two parts: the straight-line part -- proposal, tilt, first term of the
// series: no branch, so that two rounds written one after the other overlap their dependent chains -- and the series'
// continuation for the 0.6 % of proposals the first test leaves open.  A function of its arguments alone.
struct PgRound {
    double X, rX, E, S, Us;
    bool right, in_range;
};
__device__ __forceinline__ PgRound pg_round(const PgPrep &P, uint64_t key, uint32_t index, uint32_t r, uint32_t it, uint32_t stream)
{
    PgRound o;
    const bool below = P.below != 0;
    const double ptail = P.ptail, mu = P.mu, hm = 0.5 * mu;
    const double qc = below ? kPgPLevy : 0.5, lqc = below ? kPgLogPLevy : kPgLogHalf;
    const Words w = philox(key, index, r, it, stream);
    const double Ux = u01(w.w0), Um = u32_01((uint32_t)w.w1);
    o.Us = u32_01((uint32_t)(w.w1 >> 32));
    const double U2 = (Um - ptail) * P.rq;
    o.right = Um < ptail;
    const double lg = log(Ux);
    const double XR = kPgT - lg * P.rfz;                        // right piece: t + Exp(1) / f
    const double N = pg_neg_quantile(Ux * qc, -(lg + lqc));     // left piece
    const double Y = N * N;
    const double XA = fmin(pg_rcp(Y), kPgT);                    // below 1/t: truncated Levy by inversion
    const double muY = mu * Y;                                  // from 1/t on: IG(1/Z, 1), Michael-Schucany-Haas
    double XB = mu + hm * muY - hm * sqrt(4.0 * muY + muY * muY);
    XB = (U2 * (mu + XB) > mu) ? mu * mu * pg_rcp(XB) : XB;
    o.X = o.right ? XR : (below ? XA : XB);
    o.in_range = (o.right || below || !(XB > kPgT)) && o.X > 0.0;  // (X > 0: always, for a finite Z; a NaN must not reach the series' loop)
    const bool tilt = !o.right && below;
    o.E = exp(tilt ? -P.hzz * o.X : 0.0);                       // (exp(0) = 1 exactly)
    o.rX = pg_rcp(o.X);
    const double e1 = exp(o.right ? -(kPi * kPi) * o.X : -4.0 * o.rX);  // n = 1: -n (n+1) pi^2 X / 2, -2 n (n+1) / X
    o.S = 1.0 - 3.0 * e1;
    return o;
}
__device__ __forceinline__ bool pg_accept(const PgRound &o)
{
    bool accepted = o.in_range && o.Us <= o.E * o.S;
    if (o.in_range && !accepted && !(o.Us > o.E)) {  // between the first two partial sums (0.6 % of the proposals): the series goes on
        double S = o.S;
        for (int n = 2; n < 32; ++n) {  // (decided within a few terms; the bound is there so that no lane can stay for ever)
            const double nn = (double)n * (double)(n + 1);
            const double rn = (double)(2 * n + 1) * exp(o.right ? -0.5 * (kPi * kPi) * nn * o.X : -2.0 * nn * o.rX);
            if (n & 1) {
                S -= rn;
                if (o.Us <= o.E * S) { accepted = true; break; }
            } else {
                S += rn;
                if (o.Us > o.E * S) break;
            }
        }
    }
    return accepted;
}

// (a NaN or infinite argument -- a state that is already broken -- would never leave the rejection loop, and a wave that
// never finishes hangs the device: hand the NaN on, the Cholesky factorisation downstream reports it.  The same for a
// finite argument past |z| ~ 1e100, where Z^2 overflows)
__device__ __forceinline__ bool pg_bad_argument(double Z) { return !(Z < 1.0e100); }

// One draw: the rounds in turn until one is accepted.
// (Round 4 tried two ways of not waiting for a wave's unluckiest lane, both bit-identical to this loop, both slower.  The later
// rounds of a wave's unfinished draws SPREAD OVER THE WAVE -- k helper lanes per pending lane evaluating its rounds 1 ... k side
// by side: config 4's k_omega_a 90 us against 82, k_z_ob 64 against 62; the fetched parameters and the extra registers, 21
// spilled at three workgroups per CU, cost more than the rare third pass.  Rounds r and r + 1 evaluated TOGETHER in one pass
// for the sake of a lone wave's latency -- k_z_ob's draws, one wave per SIMD, are 8.5 us of its 14 -- : two copies of the
// round need 280 registers, one workgroup per CU, k_z_ob 14 -> 30 us.)
__device__ inline double pg1_draw(uint64_t key, uint32_t index, uint32_t it, uint32_t stream, double z)
{
    const double Z = 0.5 * fabs(z);
    if (pg_bad_argument(Z)) return (z - z) * __longlong_as_double(0x7ff8000000000000LL);
    const PgPrep P = pg_prep(Z);
    for (uint32_t r = 0;; ++r) {
        const PgRound a = pg_round(P, key, index, r, it, stream);
        if (pg_accept(a)) return 0.25 * a.X;
    }
}

// ---- standard gamma (Marsaglia & Tsang 2000) ---------------------------------------------------
__device__ inline double std_gamma(Cursor &c, double shape)
{
    double boost = 1.0, a = shape;
    if (!(a > 0.0 && a < 1.0e300)) return a - a + __longlong_as_double(0x7ff8000000000000LL);  // (NaN in, NaN out: see pg1_draw)
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
