// rt_device.hpp -- per-ray arithmetic of the RayTracer2 propagation loop for gfx950 (FP64 VALU).
//
// Everything here is a __host__ __device__ inline function of one ray, so that the HIP kernels in
// rt_kernels.hip inline it into the persistent bounce kernel, and so that tests/hostsim can compile the very
// same text with g++ to debug parity on a machine without a GPU (test aid only: the product has no CPU path).
//
// Reference lines are cited per function.  Unlike the oracle (oracle/rt_oracle.c), which restates the
// reference's formulas literally, this file is free to reorganise the algebra for the GPU:
//   * nearest-hit search keeps only (distance, surface, orientation); point and normal are recomputed for the
//     winner, orientation comes from the sign of b/2 + l*a instead of a normalised normal;
//   * Fresnel is evaluated in Mueller form (no complex amplitudes, atan2 or exp);
//   * the torus quartic is depressed geometrically (origin moved to the point of closest approach to the torus
//     centre), solved by Ferrari's resolvent and polished by Newton in registers.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef RT_HD
#define RT_HD __host__ __device__ __forceinline__
#endif
#ifndef RT_ANY
#define RT_ANY(x) (x) /* host build: one ray at a time.  device build: wave-wide any() */
#endif

namespace rt {

constexpr int ROW = 32, ATOM = 12, REC = 48, FIN = 32;
constexpr double TWO_PI = 6.283185307179586476925286766559;
enum { SH_PLANE = 0, SH_SPHERE = 1, SH_CYL = 2, SH_QUAD = 3, SH_TORUS = 4 };
enum { ST_NORMAL = 0, ST_DIFFUSE = 1, ST_UNIFIED = 2, ST_RETRO = 3 };
enum { COMPAT_PY_INDEX = 1, COMPAT_PY_ROOT2 = 2 };
enum { FLAG_FULL = 1, FLAG_SINGLE = 2, FLAG_FINAL = 4 };
enum { RF_HAS_REFL = 1, RF_PRIMARY = 2, RF_BORN_REFL = 4 };

struct V3 { double x, y, z; };
RT_HD V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
RT_HD V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
RT_HD V3 operator-(V3 a) { return { -a.x, -a.y, -a.z }; }
RT_HD V3 operator*(double s, V3 a) { return { s * a.x, s * a.y, s * a.z }; }
RT_HD V3 operator*(V3 a, double s) { return { s * a.x, s * a.y, s * a.z }; }
RT_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD V3 cross(V3 a, V3 b) {
    // no FMA contraction here: a x a must be exactly zero (the reference tests `sin_incident_angle > 0`,
    // RefractionReflectionAtInterface.py:99-100, and back-reflection uses normal = -direction)
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
template <class Ptr> RT_HD V3 ld3(Ptr p) { return { p[0], p[1], p[2] }; }
// 1 / sqrt(x), x > 0 finite: on the device the hardware estimate and two Newton steps (7 instructions, ~1 ulp) instead of
// an IEEE square root followed by an IEEE division (34) - a physics step normalises half a dozen vectors
#if defined(__HIP_DEVICE_COMPILE__)
RT_HD double rt_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x); const double xh = 0.5 * x;
    double e = fma(-xh * y, y, 0.5); y = fma(y, e, y);
    e = fma(-xh * y, y, 0.5); return fma(y, e, y);
}
// a / b and sqrt(x) the same way (6 and 10 instructions against 12 and 22).  rt_fdiv: b finite, nonzero and not
// denormal gives ~1 ulp; anything else gives NaN where IEEE gives 0, inf or a huge number - used only where the divisor
// is a sum of finite positive terms or where NaN is what the caller tests for.  rt_fsqrt(0) = 0, of a negative number NaN.
RT_HD double rt_fdiv(double a, double b) { double y = __builtin_amdgcn_rcp(b), e = fma(-b, y, 1.0); y = fma(y, e, y); e = fma(-b, y, 1.0); y = fma(y, e, y); return a * y; }
RT_HD double rt_fsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x), g = x * y, h = 0.5 * y, r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double e = fma(-g, g, x), v = fma(e, h, g);
    return x == 0 ? 0.0 : v;
}
// sin and cos of an angle of a few turns at most (azimuths 2 pi u, polar angles of facets): two-term reduction by pi/2
// (exact for |x| < 1e5) and the fdlibm kernels on [-pi/4, pi/4], ~45 instructions for the pair where the library's
// sincos - which must also serve 1e300 - takes ~200; within an ulp of it on the same argument.
struct SinCos { double s, c; };
__device__ __noinline__ SinCos rt_sincos_call(double x) { // out of line: ten polynomial temporaries that the callers' loops need not make room for
    if (!(fabs(x) < 1e5)) { SinCos r; sincos(x, &r.s, &r.c); return r; }
    const double n = rint(x * 0.63661977236758134308);
    const double t = fma(-n, 6.07710050650619224932e-11, fma(-n, 1.57079632673412561417, x));
    const double z = t * t;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double s = fma(t * z, ps, t);
    const double hz = 0.5 * z, w = 1.0 - hz, c = w + (((1.0 - w) - hz) + (z * z) * pc);
    const int q = (int)n & 3;
    return SinCos{ (q & 1) ? ((q & 2) ? -c : c) : ((q & 2) ? -s : s), (q & 1) ? ((q & 2) ? s : -s) : ((q & 2) ? -c : c) };
}
RT_HD void rt_sincos(double x, double* sn, double* cs) { const SinCos r = rt_sincos_call(x); *sn = r.s; *cs = r.c; }
#ifdef RT_INLINE_LOGEXP
// in line, fenced: the scheduler may not move anything across the fences, so the dozen temporaries of the polynomial live
// only between them.  (Out of line costs more than the call: a function waits for every memory operation of its caller at
// its entry - s_waitcnt vmcnt(0) lgkmcnt(0), the compiler cannot know what is in flight - and the software-pipelined physics
// loop has the next ray's rows in flight right there.)
__device__ __forceinline__ double rt_log(double x) { __builtin_amdgcn_sched_barrier(0); const double r = log(x); __builtin_amdgcn_sched_barrier(0); return r; }
__device__ __forceinline__ double rt_exp(double x) { __builtin_amdgcn_sched_barrier(0); const double r = exp(x); __builtin_amdgcn_sched_barrier(0); return r; }
#else
__device__ __noinline__ double rt_log(double x) { return log(x); } // the library's, out of line for the same reason
__device__ __noinline__ double rt_exp(double x) { return exp(x); }
#endif
__device__ __noinline__ double rt_cbrt(double x) { return cbrt(x); }
#else
RT_HD double rt_rsqrt(double x) { return 1.0 / sqrt(x); }
RT_HD double rt_fdiv(double a, double b) { return a / b; }
RT_HD double rt_fsqrt(double x) { return sqrt(x); }
RT_HD void rt_sincos(double x, double* sn, double* cs) { sincos(x, sn, cs); }
RT_HD double rt_log(double x) { return log(x); }
RT_HD double rt_exp(double x) { return exp(x); }
RT_HD double rt_cbrt(double x) { return cbrt(x); }
#endif
RT_HD V3 normalized_if(V3 a) {
    // no FMA contraction here either: normalising d and -d at two places of the code must give exactly opposite vectors,
    // whatever the compiler makes of the surrounding code (back-reflection: normal = -direction, cross() above)
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double n = a.x * a.x + a.y * a.y + a.z * a.z;
    if (n > 0) { const double s = rt_rsqrt(n); return { s * a.x, s * a.y, s * a.z }; }
    return a;
}
RT_HD int isgn(double x) { return (x > 0) - (x < 0); }

struct Ray10 { V3 d, pol; double s0, s1, s2, s3; };
// field-wise selects: a ?: over struct lvalues selects an address, which keeps both structs in (scratch) memory
RT_HD V3 sel(bool c, V3 a, V3 b) { return { c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z }; }
RT_HD Ray10 sel(bool c, const Ray10& a, const Ray10& b) { return { sel(c, a.d, b.d), sel(c, a.pol, b.pol), c ? a.s0 : b.s0, c ? a.s1 : b.s1, c ? a.s2 : b.s2, c ? a.s3 : b.s3 }; }

// Wave-uniform data (surface rows walked by the surface loop, bounds atoms and code) is read through the constant
// address space: with a uniform index hipcc then emits scalar loads (s_load into SGPRs, scalar cache) instead of
// vector loads, so the per-surface parameters cost no VGPRs and no LDS bandwidth.  Per-lane (divergent) reads of the
// winning surface's row go to the LDS copy `surf`.
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_UNIFORM __attribute__((address_space(4)))
#else
#define RT_UNIFORM
#endif
typedef const RT_UNIFORM double* udptr;
typedef const RT_UNIFORM uint32_t* uuptr;
// smeta: 4 uint32 per surface {shape, surface type, first bounds record, number of bounds records}; urec: the bounds
// programs in device form, one 64-byte record per step (see build_device_tables()).
// uprog/nprog: the frame program (see find_hit)
struct Geom { udptr usurf; udptr urec; uuptr smeta; const double* surf; int S; const double* lut; int lut_prec; double* tq; udptr uprog; int nprog; };
struct Prm {
    int K, flags, compat, maxiter, base;
    double min_travel, thrA, thrB, tir;
    uint32_t key0, key1;
};

// ---------------------------------------------------------------------------------------------- RNG
// Philox4x32-10; counter = (ray_id lo, ray_id hi, path hash, step << 20 | block); two 53-bit uniforms per block.
// Same convention as the oracle so that Monte-Carlo runs can be compared ray by ray.
struct Rng { uint32_t c0, c1, c2, step, k0, k1, idx; uint32_t o0, o1, o2, o3; };
RT_HD void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}
RT_HD void rng_init(Rng& g, uint32_t k0, uint32_t k1, int64_t ray_id, uint32_t hash, int step) {
    g.c0 = (uint32_t)(uint64_t)ray_id; g.c1 = (uint32_t)((uint64_t)ray_id >> 32); g.c2 = hash; g.step = (uint32_t)step;
    g.k0 = k0; g.k1 = k1; g.idx = 0; g.o0 = g.o1 = g.o2 = g.o3 = 0;
}
RT_HD double rng_uniform(Rng& g) {
    uint32_t which = g.idx & 1u;
    if (!which) philox(g.c0, g.c1, g.c2, (g.step << 20) | ((g.idx >> 1) & 0xFFFFFu), g.k0, g.k1, g.o0, g.o1, g.o2, g.o3);
    uint32_t a = which ? g.o2 : g.o0, b = which ? g.o3 : g.o1;
    g.idx++;
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}
RT_HD double rng_normal(Rng& g) {
    double u1 = rng_uniform(g), u2 = rng_uniform(g);
    double s_, c_; rt_sincos(TWO_PI * u2, &s_, &c_);
    return sqrt(-2.0 * rt_log(1.0 - u1)) * c_;
}
RT_HD uint32_t path_hash(uint32_t h, int step, int reflected) {
    h ^= (uint32_t)(2 * step + reflected + 1) * 0x9E3779B9u;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// ---------------------------------------------------------------------------------------------- bounds program
// include/rt_mi355x.h "Bounds program"; replaces surface.inbounds_function (python/surface.py:19-32,
// called at python/RayTracer2.py:316).  The program is the same for every lane: control flow is wave-uniform.
//
// Device form.  The public program (postfix code words that index a separate atom table) costs three dependent
// scalar loads per atom, and the bounds test sits on the critical path of every candidate hit, so
// build_device_tables() flattens it into self-contained 80-byte records {meta, hi, c0..c5, lo, -}: one scalar load
// per step at an address known in advance (the first record is requested as soon as the surface is known, each
// further one while its predecessor is evaluated).  The common atoms (coordinate, linear, axis-aligned quadratic) are
// one record kind, v = c0 x^2 + c1 y^2 + c2 z^2 + c3 x + c4 y + c5 z, evaluated without a branch, and "lo < v <= hi"
// written as two atoms and an AND (how every slab, ring and range in the reference's geometries is written) is
// fused into one record.  meta (low 32 bits of the first double):
//   bits 0-3 kind; 4 upper bound inclusive; 5 has lower bound; 6 lower bound inclusive; 7 "== hi"; 8 "!= hi";
//   13 push the result; 16-19 number of trailing stack operations; 20-31 those operations (3 bits each).
// '>' and '>=' atoms are stored negated: -(v) < -(rhs) <=> v > rhs in IEEE arithmetic, NaN included (both false).
enum { BK_POLY = 0, BK_ACCX = 3, BK_ACC4 = 4, BK_ATAN2 = 5, BK_CONST = 6, BK_NONE = 7 };
enum { BM_UPEQ = 16, BM_HASLO = 32, BM_LOEQ = 64, BM_EQ = 128, BM_NE = 256, BM_PUSH = 8192 };
constexpr int BREC = 10; // doubles per record
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __noinline__ double atan2_atom(double y, double x) { return atan2(y, x); } // rare: keep its registers out of the hot loop
#else
static inline double atan2_atom(double y, double x) { return atan2(y, x); }
#endif
struct BRec { double w[BREC]; };
RT_HD BRec brec_load(udptr r) { BRec b; for (int j = 0; j < BREC; j++) b.w[j] = r[j]; return b; }
// `first` = record `off`, already loaded (or anything if len == 0)
RT_HD bool inbounds(const Geom& G, const BRec& first, int off, int len, V3 p) {
    if (len == 0) return true;
    uint32_t st = 0;
    double acc = 0;
    udptr r = G.urec + (size_t)off * BREC;
    BRec nxt = first;
    for (int i = 0; i < len; i++) {
        const BRec c = nxt;
        r += BREC;
        if (i + 1 < len) nxt = brec_load(r); // in flight while this record is evaluated
        uint32_t m; { uint64_t u; memcpy(&u, &c.w[0], 8); m = (uint32_t)u; }
        const uint32_t kind = m & 15u;
        const double hi = c.w[1], lo = c.w[8];
        double v;
        if (kind == BK_POLY) { v = c.w[2] * p.x * p.x + c.w[3] * p.y * p.y + c.w[4] * p.z * p.z + c.w[5] * p.x + c.w[6] * p.y + c.w[7] * p.z + acc; acc = 0; }
        else if (kind == BK_ACCX) { acc = c.w[2] * p.x * p.y + c.w[3] * p.x * p.z + c.w[4] * p.y * p.z; v = 0; }
        else if (kind == BK_ACC4) { acc = c.w[2] * p.x + c.w[3] * p.y + c.w[4] * p.z + c.w[5]; v = 0; }
        else if (kind == BK_ATAN2) { v = atan2_atom(acc, c.w[2] * p.x + c.w[3] * p.y + c.w[4] * p.z + c.w[5]); acc = 0; }
        else v = 0; // BK_CONST: 0 < hi; BK_NONE: nothing pushed
        if (m & BM_PUSH) {
            const bool up = (m & BM_UPEQ) ? (v <= hi) : (v < hi), lw = (m & BM_LOEQ) ? (v >= lo) : (v > lo);
            bool bb = up && ((m & BM_HASLO) ? lw : true);
            if (m & (BM_EQ | BM_NE)) bb = (m & BM_EQ) ? (v == hi) : (v != hi);
            st = (st << 1) | (bb ? 1u : 0u);
        }
        uint32_t ops = m >> 20;
        for (uint32_t k = (m >> 16) & 15u; k > 0; k--, ops >>= 3) {
            const uint32_t op = ops & 7u, t0 = st & 1u, t1 = (st >> 1) & 1u;
            if (op == 4) st ^= 1u;
            else st = ((st >> 2) << 1) | (op == 2 ? (t0 & t1) : op == 3 ? (t0 | t1) : (t0 ^ t1));
        }
    }
    return (st & 1u) != 0;
}

} // namespace rt
#include <cmath>
#include <vector>
#include <algorithm>
#include <cstdlib>
namespace rt {
#if defined(__HIPCC__)
#define RT_HOST __host__
#else
#define RT_HOST
#endif
// Host: public tables of include/rt_mi355x.h -> device form: smeta (integer metadata as integers) and the bounds
// records described above.
// one bounds program (public postfix code) -> device records appended to recs; returns the first record's index
RT_HOST static inline size_t emit_bounds_program(const uint32_t* code, int len, const double* atoms, std::vector<double>& recs) {
    auto meta_of = [&](size_t rec) { uint32_t m; memcpy(&m, &recs[rec * BREC], 4); return m; };
    auto set_meta = [&](size_t rec, uint32_t m) { memcpy(&recs[rec * BREC], &m, 4); };
    auto emit = [&](uint32_t meta, double hi, const double* c, int nc) {
        size_t at = recs.size(); recs.resize(at + BREC, 0.0);
        memcpy(&recs[at], &meta, 4); recs[at + 1] = hi;
        for (int j = 0; j < nc; j++) recs[at + 2 + j] = c[j];
    };
    const int MAXOPS = 4;
    const size_t first = recs.size() / BREC;
    long last = -1; // record that can still take trailing stack operations
    for (int i = 0; i < len; i++) {
        const uint32_t ins = code[i], op = ins >> 24, arg = ins & 0xFFFFFFu;
        if (op == 1) {
            const double* a = atoms + (size_t)arg * ATOM;
            const int kind = (int)a[0], cmp = (int)a[1];
            const bool neg = (cmp == 2 || cmp == 3); // > , >=  ->  negated < , <=
            const double sg = neg ? -1.0 : 1.0, hi = sg * a[2];
            const uint32_t cm = BM_PUSH | ((cmp == 1 || cmp == 3) ? BM_UPEQ : 0u) | (cmp == 4 ? BM_EQ : 0u) | (cmp == 5 ? BM_NE : 0u);
            if (kind == 3) { double c[6] = { 0, 0, 0, 0, 0, 0 }; c[3 + (int)a[3]] = sg; emit(BK_POLY | cm, hi, c, 6); }
            else if (kind == 1) { double c[6] = { 0, 0, 0, sg * a[9], sg * a[10], sg * a[11] }; emit(BK_POLY | cm, hi, c, 6); }
            else if (kind == 0) {
                if (a[6] != 0 || a[7] != 0 || a[8] != 0) { double c[3] = { sg * a[6], sg * a[7], sg * a[8] }; emit(BK_ACCX, 0.0, c, 3); }
                double c[6] = { sg * a[3], sg * a[4], sg * a[5], sg * a[9], sg * a[10], sg * a[11] };
                emit(BK_POLY | cm, hi, c, 6);
            } else { // atan2 is odd in its first argument: -atan2(y,x) = atan2(-y,x)
                double cy[4] = { sg * a[3], sg * a[4], sg * a[5], sg * a[6] }, cx[4] = { a[7], a[8], a[9], a[10] };
                emit(BK_ACC4, 0.0, cy, 4); emit(BK_ATAN2 | cm, hi, cx, 4);
            }
            last = (long)(recs.size() / BREC) - 1;
        } else if (op == 5 || op == 6) {
            emit(BK_CONST | BM_PUSH, op == 5 ? 1.0 : 0.0, nullptr, 0); // pushes (0 < hi)
            last = (long)(recs.size() / BREC) - 1;
        } else if (op == 2 || op == 3 || op == 4 || op == 7) {
            // "f < a  AND  -f < b" on the same polynomial f: one record with two bounds
            const long nrec = (long)(recs.size() / BREC);
            if (op == 2 && last == nrec - 1 && nrec - (long)first >= 2) {
                const size_t r2 = (size_t)nrec - 1, r1 = r2 - 1;
                const uint32_t m1 = meta_of(r1), m2 = meta_of(r2), plain = ~(uint32_t)BM_UPEQ;
                bool ok = (m1 & plain) == (BK_POLY | BM_PUSH) && (m2 & plain) == (BK_POLY | BM_PUSH);
                // r1 must not add an accumulator (an ACCX record right before it)
                if (ok && r1 > first && (meta_of(r1 - 1) & 15u) == BK_ACCX) ok = false;
                for (int j = 2; ok && j < 8; j++) ok = recs[r1 * BREC + j] == -recs[r2 * BREC + j];
                if (ok) {
                    recs[r1 * BREC + 8] = -recs[r2 * BREC + 1];
                    set_meta(r1, m1 | BM_HASLO | ((m2 & BM_UPEQ) ? BM_LOEQ : 0u));
                    recs.resize(r2 * BREC); last = (long)r1;
                    continue;
                }
            }
            uint32_t meta = last >= 0 ? meta_of((size_t)last) : 0u;
            if (last < 0 || ((meta >> 16) & 15u) >= (uint32_t)MAXOPS) { emit(BK_NONE, 0.0, nullptr, 0); last = (long)(recs.size() / BREC) - 1; meta = BK_NONE; }
            const uint32_t k = (meta >> 16) & 15u;
            meta = (meta & ~(15u << 16)) | ((k + 1) << 16) | (op << (20 + 3 * k));
            set_meta((size_t)last, meta);
        }
    }
    return first;
}

// Host: public tables of include/rt_mi355x.h -> device form: smeta (integer metadata as integers) and the bounds
// records described above.  Call build_program() afterwards (it appends residual programs to recs), then
// finish_device_tables().
RT_HOST static inline void build_device_tables(const double* table, int S, const double* atoms, int nA, const uint32_t* code, std::vector<double>& recs, std::vector<uint32_t>& smeta) {
    (void)nA;
    smeta.resize((size_t)(S > 0 ? S : 1) * 4); recs.clear();
    for (int s = 0; s < S; s++) {
        const double* r = table + (size_t)s * ROW;
        const int off = (int)r[27], len = (int)r[28];
        const size_t first = emit_bounds_program(code + off, len, atoms, recs);
        smeta[4 * s] = (uint32_t)(int)r[0]; smeta[4 * s + 1] = (uint32_t)(int)r[1];
        smeta[4 * s + 2] = (uint32_t)first; smeta[4 * s + 3] = (uint32_t)(recs.size() / BREC - first);
    }
}
RT_HOST static inline void finish_device_tables(std::vector<double>& recs) {
    recs.resize(recs.size() + BREC, 0.0); // the record after the last program is read (and ignored) by surfaces without bounds
}

// ---------------------------------------------------------------------------------------------- nearest hit
struct Hit {
    double l; int surf; int orient;
#ifdef RT_PROFILE
    long long pm, ps, pc, pn, pf, pt_; // diagnostic counters of the nearest-hit search (see tools/phase_profile.py)
#endif
};

// depressed quartic y^4 + p y^2 + q y + r = 0: real roots into y[0..3], returns count.  Stands in for the
// per-ray roots() of MATLAB/RayToX/RayToTorus.m:137-141.
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef RT_INLINE_ALL
#define RT_COLD __device__ __forceinline__
#else
#define RT_COLD __device__ __noinline__ /* out of line: its cbrt/acos temporaries stay out of the surface loop's registers */
#endif
// Division and square root of the solver: hardware estimate + two Newton steps (~1 ulp; 6 and 10 instructions where the
// IEEE sequences take 12 and 22 - the solver has 20 divisions and 11 roots and runs with a quarter of its lanes on).
// A quotient whose divisor is zero, denormal or huge comes out NaN or inf instead of IEEE's value: every use below
// rejects non-finite results already.  The square root of zero is zero (tangent rays live there), of a negative number NaN.
__device__ __forceinline__ double qs_div(double a, double b) { double y = __builtin_amdgcn_rcp(b), e = fma(-b, y, 1.0); y = fma(y, e, y); e = fma(-b, y, 1.0); y = fma(y, e, y); return a * y; }
__device__ __forceinline__ double qs_sqrt(double x) {
    double y = __builtin_amdgcn_rsq(x), g = x * y, h = 0.5 * y, r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double e = fma(-g, g, x), v = fma(e, h, g);
    return x == 0 ? 0.0 : v;
}
#else
#define RT_COLD static inline
static inline double qs_div(double a, double b) { return a / b; }
static inline double qs_sqrt(double x) { return sqrt(x); }
#endif
struct Roots4 { double y0, y1, y2, y3; };
RT_HD Roots4 solve_depressed_quartic_body(double p, double q, double r) {
    double y0, y1, y2, y3;
    // all four slots are always written (NaN = no real root there): static register indices only
    y0 = y1 = y2 = y3 = NAN;
    double scale = fabs(p) + qs_sqrt(fabs(r)) + 1e-300;
    if (fabs(q) <= 1e-14 * scale * qs_sqrt(scale)) { // biquadratic
        double disc = p * p - 4 * r;
        if (disc >= 0) {
            double sq = qs_sqrt(disc), w1 = 0.5 * (-p + sq), w2 = 0.5 * (-p - sq);
            if (w1 >= 0) { double s = qs_sqrt(w1); y0 = s; y1 = -s; }
            if (w2 >= 0) { double s = qs_sqrt(w2); y2 = s; y3 = -s; }
        }
    } else {
        // largest real root z > 0 of z^3 + 2p z^2 + (p^2 - 4r) z - q^2 = 0
        constexpr double third = 1.0 / 3.0; // constants as factors: what they feed is polished by Newton steps below
        double A = -p * p * third - 4 * r, B = -2 * p * p * p * (1.0 / 27.0) + 8 * p * r * third - q * q;
        double hB = 0.5 * B, tA = A * third, D = hB * hB + tA * tA * tA, w;
        if (D > 0) { double sD = qs_sqrt(D); w = cbrt(-hB + sD) + cbrt(-hB - sD); }
        else { double m = qs_sqrt(-tA), c = qs_div(-hB, m * m * m); c = c > 1 ? 1 : (c < -1 ? -1 : c); 
            // cos(acos(c) / 3) is the root t in [1/2, 1] of 4 t^3 - 3 t = c.  Newton from t0 = 1/2 + 1/2 sqrt(2 (1 + c) / 3) >= t
            // (from (t + 1)(2 t - 1)^2 = 1 + c and t + 1 >= 3/2; exact where the root is double, c = -1): the function is convex
            // there, the steps come down monotonically, four of them reach 1e-16 from the worst start (0.077 off at c = 1).
            // 66 instructions where acos and cos take ~170; the two Newton steps on the resolvent below follow in any case.
            double t = 0.5 + 0.5 * qs_sqrt((1 + c) * (2.0 / 3.0));
#pragma unroll
            for (int it = 0; it < 4; it++) { const double t2 = t * t, f = (4 * t2 - 3) * t - c, fp = 12 * t2 - 3; if (fp > 0) t -= qs_div(f, fp); }
            w = 2 * m * t; }
        double z = w - 2 * p * third, c1 = 2 * p, c2 = p * p - 4 * r, c3 = -q * q;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            double f = ((z + c1) * z + c2) * z + c3, fp = (3 * z + 2 * c1) * z + c2;
            if (fp != 0) { double zn = z - qs_div(f, fp); if (zn > 0 && isfinite(zn)) z = zn; }
        }
        if (z > 0) {
            double s = qs_sqrt(z), qs = qs_div(q, s), t = 0.5 * (p + z - qs), u = 0.5 * (p + z + qs);
            double d1 = z - 4 * t, d2 = z - 4 * u;
            if (d1 >= 0) { double h = -0.5 * (s + qs_sqrt(d1)); y0 = h; y1 = (h != 0) ? qs_div(t, h) : 0.0; } // y^2 + s y + t
            if (d2 >= 0) { double h = 0.5 * (s + qs_sqrt(d2)); y2 = h; y3 = (h != 0) ? qs_div(u, h) : 0.0; }  // y^2 - s y + u
        }
    }
#define RT_POLISH(v)                                                                                 \
    {                                                                                                \
        for (int it = 0; it < 1; it++) { /* one step: a second one changed no root of the known-answer tests (worst 1e-10, grazing) */ \
            double f = ((v * v + p) * v + q) * v + r, fp = (4 * v * v + 2 * p) * v + q;              \
            if (fp != 0) { double vn = v - qs_div(f, fp); if (isfinite(vn)) v = vn; }                \
        }                                                                                            \
    }
    RT_POLISH(y0) RT_POLISH(y1) RT_POLISH(y2) RT_POLISH(y3)
#undef RT_POLISH
    return Roots4{ y0, y1, y2, y3 };
}
RT_COLD Roots4 solve_depressed_quartic(double p, double q, double r) { return solve_depressed_quartic_body(p, q, r); }

// torus normal (unnormalised sign only matters for the orientation): RayToTorus.m:149-158
template <class Ptr> RT_HD V3 torus_normal(Ptr g, V3 p) {
    V3 ax = ld3(g + 3), xx = p - ld3(g);
    double xxa = dot(xx, ax); V3 yv = xxa * ax, uu = xx - yv;
    double un = sqrt(dot(uu, uu));
    return (1.0 / g[7]) * (uu * (1 - g[6] / un) + yv);
}

// Wave primitives of the nearest-hit search.  Host build (tests/hostsim): a "wave" of one lane.
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_BALLOT(x) __ballot(x)
#define RT_WAVE 64
RT_HD int rt_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
RT_HD int rt_rank(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); } // set bits below my lane
// LDS instructions of one wave execute in issue order, so a read sees an earlier write of another lane: only the
// compiler has to be kept from reordering across this point (a fence would also wait for the caller's global atomics)
RT_HD void rt_wave_sync() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }
// Reciprocal and square root of the nearest-hit search: hardware estimate (v_rcp_f64 / v_rsq_f64, ~26 bits) plus two
// Newton steps, ~1 ulp, without the scaling and fix-up instructions of the IEEE sequences (5 and 8 instructions
// instead of ~14 and ~17).  Out-of-range arguments give NaN where IEEE gives 0 or inf (0, negative, inf): every
// caller treats all of those as "no root".
RT_HD double rt_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x), e = fma(-x, y, 1.0);
    y = fma(y, e, y); e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
RT_HD double rt_sqrt(double x) {
    double y = __builtin_amdgcn_rsq(x), g = x * y, h = 0.5 * y, r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    double e = fma(-g, g, x);
    return fma(e, h, g);
}
#else
#define RT_BALLOT(x) ((x) ? 1ull : 0ull)
#define RT_WAVE 1
RT_HD int rt_lane() { return 0; }
RT_HD int rt_rank(unsigned long long) { return 0; }
RT_HD void rt_wave_sync() {}
RT_HD double rt_rcp(double x) { return 1.0 / x; }
RT_HD double rt_sqrt(double x) { return sqrt(x); }
#endif
constexpr int TQ_SLOTS = 64, TQ_TORI = 4; // per-wave torus queue: 64 slots of 4 doubles, at most 4 distinct tori pending

// One candidate root l_ of surface n: validity tests of python/RayTracer2.py:316 and the running minimum over
// (distance, surface order).  The reference takes, per surface, the first smallest root (:319-321) and then replaces
// the best so far only on strict "<" while walking the list (:362), so among equal distances the surface that comes
// first in the list wins.  Surfaces are visited here in program order (frames first, see below), not list order:
// a candidate wins a tie only against a surface that comes later in the list, which gives the same winner.
RT_HD bool root_wins(double l_, int n, double lmin, const Hit& h) { return (l_ > lmin) && ((l_ < h.l) || (l_ == h.l && n < h.surf)); }
RT_HD void try_root(const Geom& G, V3 sp, V3 d, int n, int shape, udptr g, const BRec& b0, int boff, int blen, double lmin, double l_, double s_, bool act, Hit& h) {
    bool c_ = act && root_wins(l_, n, lmin, h);
    if (shape != SH_TORUS) c_ = c_ && (s_ != 0);
    if (RT_ANY(c_)) {
#ifdef RT_PROFILE
        long long t0_ = (long long)__builtin_readcyclecounter();
#endif
        if (c_) {
            V3 p_ = sp + l_ * d;
            if (shape == SH_TORUS) { // sign of d . normal; normal = (1/r2)(u (1 - R1/|u|) + y): scaled by |u| r2^2 > 0
                V3 ax = ld3(g + 3), xx = p_ - ld3(g);
                double xxa = dot(xx, ax); V3 yv = xxa * ax, uu = xx - yv;
                double un = sqrt(dot(uu, uu));
                s_ = dot(d, (un - g[6]) * uu + un * yv) * g[7];
            }
            int o_ = -isgn(s_);
            if (o_ != 0 && inbounds(G, b0, boff, blen, p_)) { h.l = l_; h.surf = n; h.orient = o_; }
        }
#ifdef RT_PROFILE
        h.pc += (long long)__builtin_readcyclecounter() - t0_; h.pn += 1;
#endif
    }
}

// Torus quartics are not solved where they are met: only a few lanes of a wave pass the prefilter of any one torus,
// but nearly every wave has some lane that does, so solving in place would run the (expensive) solver once per
// torus per wave at a few percent lane utilisation.  Instead the lanes that need a solve push their depressed
// quartic into a per-wave queue (G.tq, LDS); when the queue is full, and after the last surface, lanes 0..n-1 solve
// the n queued quartics in one call and the owners take their roots back.
struct TorusQ {
    double* tq;
    int npend, ntor;              // wave-uniform: slots used, distinct tori pending
    unsigned long long torpack;   // wave-uniform: 16-bit surface index of each pending torus
    uint32_t slotpack;            // per lane: 8-bit (slot + 1) for each pending torus, 0 = none
};
RT_HD void torus_flush(const Geom& G, const Prm& P, V3 sp, V3 d, int six_last, bool act, Hit& h, TorusQ& T) {
    if (T.npend == 0) return;
#ifdef RT_PROFILE
    h.ps += 1;
    long long tf_ = (long long)__builtin_readcyclecounter();
#endif
    rt_wave_sync();
    for (int j = rt_lane(); j < T.npend; j += RT_WAVE) { // device: one slot per lane
        double* q = T.tq + 4 * j;
        double t0 = q[3];
        Roots4 y = solve_depressed_quartic(q[0], q[1], q[2]);
        q[0] = y.y0 + t0; q[1] = y.y1 + t0; q[2] = y.y2 + t0; q[3] = y.y3 + t0;
    }
    rt_wave_sync();
#pragma unroll 1
    for (int k = 0; k < T.ntor; k++) {
        const int nt = (int)((T.torpack >> (16 * k)) & 0xFFFFull);
        const uint32_t my = (T.slotpack >> (8 * k)) & 0xFFu;
        udptr g = G.usurf + nt * ROW + 2;
        uuptr sm = G.smeta + 4 * nt;
        const int boff = (int)sm[2], blen = (int)sm[3];
        const BRec b0 = brec_load(G.urec + (size_t)boff * BREC);
        const double lmin = (six_last == nt + P.base) ? P.min_travel : 0.0;
        const double* q = T.tq + 4 * (my ? my - 1 : 0);
#pragma unroll 1
        for (int m = 0; m < 4; m++) try_root(G, sp, d, nt, SH_TORUS, g, b0, boff, blen, lmin, my ? q[m] : NAN, 0.0, act, h);
    }
    rt_wave_sync();
#ifdef RT_PROFILE
    h.pf += (long long)__builtin_readcyclecounter() - tf_;
#endif
    T.npend = 0; T.ntor = 0; T.torpack = 0; T.slotpack = 0;
}

// One surface by the reference's own formulas (RayToPlane/Sphere/Cylinder/Quadsurface/Torus): every shape produces
// its candidate roots (NaN = none) and the sign of d.normal there; try_root() applies the validity tests and keeps
// the running minimum.  `d` is a unit vector (normalised once per step, RayToShape.py:11-14), so the leading
// coefficient of the sphere quadratic and of the torus quartic is taken as 1.
RT_HD void generic_surface(const Geom& G, const Prm& P, V3 sp, V3 d, int six_last, bool act, int n, Hit& h, TorusQ& T) {
    const bool py2 = (P.compat & COMPAT_PY_ROOT2) != 0;
    udptr row = G.usurf + n * ROW;
    udptr g = row + 2;
    uuptr sm = G.smeta + 4 * n;
    const int shape = (int)sm[0], boff = (int)sm[2], blen = (int)sm[3];
    const BRec b0 = brec_load(G.urec + (size_t)boff * BREC); // first bounds record: on its way while the roots are found
    const double lmin = (six_last == n + P.base) ? P.min_travel : 0.0; // RayTracer2.py:316
    double l0 = NAN, l1 = NAN, s0 = 0, s1 = 0; // s = d . (outward normal) up to a positive factor
    int M = 2;
    if (shape == SH_PLANE) { // RayToPlane.py:35-45
        V3 pn = ld3(g + 3);
        double den = dot(d, pn), num = dot(ld3(g) - sp, pn);
        l0 = num / den; s0 = den; M = 1;
    } else if (shape == SH_SPHERE) { // RayToSphere.m:72-87 with a = 1: l = -B +- sqrt(B^2 - C)
        V3 x = sp - ld3(g);
        double r = g[3], B = dot(d, x), C = dot(x, x) - r * r, D = B * B - C;
        if (RT_ANY(D >= 0)) {
            double sq = sqrt(D); // NaN if D < 0: fails every comparison below
            l0 = sq - B; l1 = py2 ? -l0 : -B - sq;
            s0 = (B + l0) * r; s1 = (B + l1) * r;
        }
    } else if (shape == SH_CYL) { // RayToCylinder.m:84-122
        V3 ax = ld3(g + 3), x = sp - ld3(g);
        double r = g[6], xa = dot(x, ax), da = dot(d, ax);
        V3 u = xa * ax - x, v = da * ax - d;
        double a = dot(v, v), hb = dot(u, v), c = dot(u, u) - r * r; // hb = b/2
        if (a != 0) {
            double ia = 1.0 / a, sq = sqrt(hb * hb - a * c);
            l0 = (sq - hb) * ia; l1 = py2 ? -l0 : (-hb - sq) * ia;
        } else if (hb != 0) { l0 = l1 = -0.5 * c / hb; }
        s0 = (hb + l0 * a) * r; s1 = (hb + l1 * a) * r;
    } else if (shape == SH_QUAD) { // RayToQuadsurface.m:83-123
        udptr Q = g; V3 Pv = ld3(g + 9); double R = g[12];
        V3 dQ = { d.x * Q[0] + d.y * Q[3] + d.z * Q[6], d.x * Q[1] + d.y * Q[4] + d.z * Q[7], d.x * Q[2] + d.y * Q[5] + d.z * Q[8] };
        V3 sQ = { sp.x * Q[0] + sp.y * Q[3] + sp.z * Q[6], sp.x * Q[1] + sp.y * Q[4] + sp.z * Q[7], sp.x * Q[2] + sp.y * Q[5] + sp.z * Q[8] };
        double a = dot(dQ, d), sQd = dot(sQ, d), dP = dot(d, Pv), b = dP + dot(dQ, sp) + sQd, c = R + dot(sp, Pv) + dot(sQ, sp);
        bool lin = (b != 0) && (fabs(4 * a * c) < (100 * 2.220446049250313e-16) * (b * b));
        if (lin) { l0 = -c / b; l1 = -b / a; }
        else if (a != 0) { double ia = 0.5 / a, sq = sqrt(b * b - 4 * a * c); l0 = (sq - b) * ia; l1 = (-b - sq) * ia; }
        double g0 = 2 * sQd + dP; // d . (2 pQ + P) = g0 + 2 l a
        s0 = g0 + 2 * l0 * a; s1 = g0 + 2 * l1 * a;
    } else if (shape == SH_TORUS) { // RayToTorus.m:88-159
        M = 0;
#ifdef RT_PROFILE
        long long tt_ = (long long)__builtin_readcyclecounter();
#endif
        V3 ax = ld3(g + 3), x = sp - ld3(g);
        double R1 = g[6], r2 = g[7];
        // The torus lies inside the slab |axial| <= r2 and between the cylinders of radius R1 -+ r2 about its
        // axis.  Ray parameter interval inside slab, outer cylinder and (lmin, best hit so far); nothing to
        // solve when it is empty or when the ray stays inside the inner cylinder throughout it (radius^2 is
        // convex in t: its maximum over an interval is at an end).  r2 is widened by 1e-9 so that rounding
        // never drops a real root.
        double re = fabs(r2) * (1 + 1e-9), Ro = fabs(R1) + re, Ri = fabs(R1) - re;
        double xa0 = dot(x, ax), da = dot(d, ax), xd = dot(x, d), xx = dot(x, x);
        double ida = 1.0 / da, ta = (-re - xa0) * ida, tb = (re - xa0) * ida;
        double lo = fmin(ta, tb), hi = fmax(ta, tb);
        if (da == 0) { bool in = fabs(xa0) <= re; lo = in ? -INFINITY : INFINITY; hi = in ? INFINITY : -INFINITY; }
        double qa = 1 - da * da, qb = xd - xa0 * da, qc = xx - xa0 * xa0; // radius^2(t) = qa t^2 + 2 qb t + qc
        double disc = qb * qb - qa * (qc - Ro * Ro);
        if (qa > 0) { double iq = 1.0 / qa, sq = sqrt(disc); lo = fmax(lo, (-qb - sq) * iq); hi = fmin(hi, (sq - qb) * iq); }
        lo = fmax(lo, lmin); hi = fmin(hi, h.l);
        bool maybe = act && (disc >= 0) && (lo <= hi);
        if (Ri > 0) maybe = maybe && !(((qa * lo + 2 * qb) * lo + qc < Ri * Ri) && ((qa * hi + 2 * qb) * hi + qc < Ri * Ri));
        const unsigned long long mb = RT_BALLOT(maybe);
#ifdef RT_PROFILE
        h.pm += __popcll(mb);
#endif
        if (mb != 0ull) {
            const int cnt = __builtin_popcountll(mb);
            if (T.npend + cnt > TQ_SLOTS || T.ntor == TQ_TORI) torus_flush(G, P, sp, d, six_last, act, h, T); // no room: solve what is queued first
            if (maybe) { // depressed quartic about the point of closest approach to the torus centre
                int slot = T.npend + rt_rank(mb);
                double t0 = -xd;
                V3 xs = x + t0 * d;
                double X2 = dot(xs, xs), xa = dot(xs, ax), K = X2 + R1 * R1 - r2 * r2, R4 = 4 * R1 * R1;
                double* q = T.tq + 4 * slot;
                q[0] = 2 * K - R4 * (1 - da * da); q[1] = 2 * R4 * xa * da; q[2] = K * K - R4 * (X2 - xa * xa); q[3] = t0;
                T.slotpack |= (uint32_t)(slot + 1) << (8 * T.ntor);
            }
            T.torpack |= (unsigned long long)(unsigned)n << (16 * T.ntor);
            T.ntor++; T.npend += cnt;
        }
#ifdef RT_PROFILE
        h.pt_ += (long long)__builtin_readcyclecounter() - tt_;
#endif
    }
#pragma unroll 1
    for (int m = 0; m < M; m++) try_root(G, sp, d, n, shape, g, b0, boff, blen, lmin, m == 0 ? l0 : l1, m == 0 ? s0 : s1, act, h);
}

// Frame program.  Detector geometries are overwhelmingly coaxial: in the jar geometry 26 of 39 surfaces share the z
// axis and the other 13 the camera axis.  build_program() groups the surfaces whose own axis (cylinder/torus axis,
// plane normal) lies on a common line ("frame": point c0, unit vector a^), and spheres / quadrics of revolution
// centred on that line.  Per ray and frame the kernel computes once
//     x = sp - c0, xa = x.a^, da = d.a^, u = xa a^ - x, v = da a^ - d, a = v.v, hb = u.v, uu = u.u, xd = x.d, xx = x.x
// (the cylinder quantities of RayToCylinder.m:84-100) and every member surface then costs a handful of operations:
//     cylinder r        : c = uu - r^2, l = (-hb +- sqrt(hb^2 - a c)) / a                     (RayToCylinder.m:101-105)
//     plane  n^ = +-a^  : l = (ka - xa) / da, ka = (p0 - c0).a^                                (RayToPlane.m:78)
//     sphere c0 + t a^  : B = xd - t da, C = xx - 2 t xa + t^2 - r^2, l = -B +- sqrt(B^2 - C)  (RayToSphere.m:72-77)
//     quadric alpha(x^2 + y^2) + gamma z^2 + pa z + R' in frame coordinates                   (RayToQuadsurface.m:83-107)
// The sign of d.normal at the roots of a quadratic is the sign of +-sqrt(discriminant), so orientations cost nothing.
// Bounds that are conjunctions of ranges of the frame's own coordinates - axial (xa + l da), radial^2
// ((a l + 2 hb) l + uu) and distance^2 from a point of the axis - are evaluated from those quantities without
// forming the hit point; whatever else a bounds program contains stays a (residual) generic program evaluated at
// the point.  Surfaces that fit no frame run generic_surface().  Everything is wave-uniform control flow over
// 128-byte records read with scalar loads.
constexpr int PREC = 16; // doubles per program record
enum { PK_FRAME = 0, PK_PLANE = 1, PK_SPHERE = 2, PK_CYL = 3, PK_QUAD = 4, PK_GENERIC = 5, PK_TORUS = 6, PK_GROUP = 7 };
enum { FA_AXIAL = 0, FA_RADIAL2 = 1, FA_DIST2 = 2 };
// record word 0 (64-bit integer): bits 0-3 kind, 4-15 surface number, 16-17 number of inline range atoms,
// 20+4j..23+4j atom j: bits 0-1 coordinate (FA_*) (bounds are exclusive; inclusive ones are stored one ulp wider);
// 32-51 first record / 52-63 number of records of the residual bounds program.
// words 1-6: constants of the kind; words 7+3j..9+3j: atom j {lo, hi, t}.
//   PK_FRAME c0(3) a^(3) | PK_PLANE ka, sigma | PK_CYL r, r^2 | PK_SPHERE t, r, t^2 - r^2 | PK_QUAD alpha, gamma, pa, R'
//   PK_TORUS t, R1, r2, centre(3)
struct FrameQ { double xa, da, ida, a, ia, hb, uu, xd, xx; int sda; };
// Only the meta word is read when the record is, the others where the kind's code uses them (scalar loads of just those
// words): a record held in full is 32 scalar registers, and the surface loop then spills ~175 of them - as vector-lane
// writes, which take issue slots of the unit the kernel is bound by.  Measured (nearest hit, ms/step): all 16 words up
// front 242.6, 10 words 236.6, 7 words 249.4, 4 words 237.6, the meta word alone 231.4 (41 spilled).
#ifndef RT_PREC_EAGER
#define RT_PREC_EAGER 1
#endif
constexpr int PREC_EAGER = RT_PREC_EAGER;
struct PRec { double w[PREC_EAGER]; udptr p; };
RT_HD double prec_w(const PRec& R, int j) { return j < PREC_EAGER ? R.w[j] : R.p[j]; }
RT_HD PRec prec_load(udptr r) { PRec p; for (int j = 0; j < PREC_EAGER; j++) p.w[j] = r[j]; p.p = r; return p; }

RT_HD double frame_coord(const FrameQ& F, uint32_t coord, double t, double l) {
    if (coord == FA_AXIAL) return F.xa + l * F.da;
    if (coord == FA_RADIAL2) return (F.a * l + 2 * F.hb) * l + F.uu;
    return (l + 2 * (F.xd - t * F.da)) * l + (F.xx - 2 * t * F.xa + t * t);
}
// candidate tests without short-circuit evaluation: the operands are cheap and per lane, a branch per "&&" is not
RT_HD bool wins_nb(double l_, double lmin, double bl, bool first) { return (l_ > lmin) & ((l_ < bl) | ((l_ == bl) & first)); }
// both bounds strict: build_program() moves an inclusive bound one ulp outwards (v >= lo <=> v > nextafter(lo, -inf))
RT_HD bool in_range(double v, double lo, double hi) { return (v > lo) & (v < hi); }

// Roots l0, l1 of frame member `R` (orientations o0, o1; ok0, ok1 = root exists and is not tangent): range atoms,
// residual bounds, running minimum.
template <bool TWO> RT_HD void frame_candidates(const Geom& G, const Prm& P, const FrameQ& F, const PRec& R, uint32_t mlo, uint32_t mhi, V3 sp, V3 d, int six_last, bool act,
                            double l0, double l1, int o0, int o1, bool ok0, bool ok1, Hit& h) {
    const int n = (int)((mlo >> 4) & 0xFFFu);
    const double lmin = (six_last == n + P.base) ? P.min_travel : 0.0; // RayTracer2.py:316
    const bool first = n < h.surf;
    bool c0_ = act & ok0 & wins_nb(l0, lmin, h.l, first), c1_ = TWO ? (act & ok1 & wins_nb(l1, lmin, h.l, first)) : false;
    if (!RT_ANY(c0_ | c1_)) return; // no ray of the wave has a root here in front of its best hit (nearest hit: -1.8 %)
    const uint32_t na = (mlo >> 16) & 3u;
    if (na > 0) { const uint32_t ck = (mlo >> 20) & 3u; c0_ = c0_ & in_range(frame_coord(F, ck, prec_w(R, 9), l0), prec_w(R, 7), prec_w(R, 8)); if (TWO) c1_ = c1_ & in_range(frame_coord(F, ck, prec_w(R, 9), l1), prec_w(R, 7), prec_w(R, 8)); }
    if (na > 1) { const uint32_t ck = (mlo >> 24) & 3u; c0_ = c0_ & in_range(frame_coord(F, ck, prec_w(R, 12), l0), prec_w(R, 10), prec_w(R, 11)); if (TWO) c1_ = c1_ & in_range(frame_coord(F, ck, prec_w(R, 12), l1), prec_w(R, 10), prec_w(R, 11)); }
    if (na > 2) { const uint32_t ck = (mlo >> 28) & 3u; c0_ = c0_ & in_range(frame_coord(F, ck, prec_w(R, 15), l0), prec_w(R, 13), prec_w(R, 14)); if (TWO) c1_ = c1_ & in_range(frame_coord(F, ck, prec_w(R, 15), l1), prec_w(R, 13), prec_w(R, 14)); }
    const int blen = (int)(mhi >> 20);
    if (blen != 0) { // residual bounds program, at the point
        const int boff = (int)(mhi & 0xFFFFFu);
        if (RT_ANY(c0_)) { if (c0_) c0_ = inbounds(G, brec_load(G.urec + (size_t)boff * BREC), boff, blen, sp + l0 * d); }
        if (TWO) { if (RT_ANY(c1_)) { if (c1_) c1_ = inbounds(G, brec_load(G.urec + (size_t)boff * BREC), boff, blen, sp + l1 * d); } }
    }
    if (TWO) { // same surface: the first smallest root (RayTracer2.py:319-321)
        const bool take1 = c1_ & (!c0_ | (l1 < l0)), any = c0_ | c1_;
        const double ln = take1 ? l1 : l0; const int on = take1 ? o1 : o0;
        h.l = any ? ln : h.l; h.surf = any ? n : h.surf; h.orient = any ? on : h.orient;
    } else { h.l = c0_ ? l0 : h.l; h.surf = c0_ ? n : h.surf; h.orient = c0_ ? o0 : h.orient; }
}

// Torus queue of frame members: slot = {p, q, r, t0, window lo, hi} in, {l0..l3 (in the window, ascending; NaN = none), sign bits of f'(l)} out (6 doubles per slot).
constexpr int TQ_STRIDE = 6;
// One queued quartic, solved in place by one lane (out of line with the solver: nothing of it competes with the surface
// loop's registers).
RT_COLD void frame_torus_slot(double* q) {
    const double p_ = q[0], q_ = q[1], t0 = q[3];
    // the owner's window of ray parameters (slab, annulus, min travel, best hit when it queued), a little wider
    const double wlo = q[4] - 1e-9 * (1 + fabs(q[4])), whi = q[5] + 1e-9 * (1 + fabs(q[5]));
    Roots4 y = solve_depressed_quartic_body(p_, q_, q[2]);
    // d.normal at a root has the sign of f'(y) = 4 y^3 + 2 p y + q (times sign(R1 r2), applied by the owner).
    // Roots outside the window cannot be the owner's hit: they are dropped here, where one lane works per quartic, and
    // the others sorted to the front (smallest first), so that the owners - all 64 lanes of the wave walking through
    // every pending torus for the sake of a few - usually have one pair of candidates to test per torus, not two.
    double l[4] = { y.y0 + t0, y.y1 + t0, y.y2 + t0, y.y3 + t0 }; uint32_t c[4];
    const double yy[4] = { y.y0, y.y1, y.y2, y.y3 };
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const double f = (4 * yy[m] * yy[m] + 2 * p_) * yy[m] + q_;
        c[m] = f > 0 ? 1u : (f < 0 ? 2u : 0u);
        l[m] = ((l[m] >= wlo) & (l[m] <= whi)) ? l[m] : INFINITY; // NaN (no root) fails both
    }
#define RT_CE(a_, b_) { const bool sw = l[b_] < l[a_]; const double tl = sw ? l[b_] : l[a_], th = sw ? l[a_] : l[b_]; const uint32_t tc = sw ? c[b_] : c[a_], uc = sw ? c[a_] : c[b_]; l[a_] = tl; l[b_] = th; c[a_] = tc; c[b_] = uc; }
    RT_CE(0, 1) RT_CE(2, 3) RT_CE(0, 2) RT_CE(1, 3) RT_CE(1, 2)
#undef RT_CE
    const uint64_t sg = (uint64_t)c[0] | ((uint64_t)c[1] << 2) | ((uint64_t)c[2] << 4) | ((uint64_t)c[3] << 6);
#pragma unroll
    for (int m = 0; m < 4; m++) q[m] = l[m] < INFINITY ? l[m] : NAN;
    double sd; memcpy(&sd, &sg, 8); q[4] = sd;
}
RT_HD void frame_torus_flush(const Geom& G, const Prm& P, const FrameQ& F, udptr prog, V3 sp, V3 d, int six_last, bool act, Hit& h, TorusQ& T) {
    if (T.npend == 0) return;
    rt_wave_sync();
    for (int j = rt_lane(); j < T.npend; j += RT_WAVE) { // device: one slot per lane
        frame_torus_slot(T.tq + TQ_STRIDE * j);
    }
    rt_wave_sync();
#pragma unroll 1
    for (int k = 0; k < T.ntor; k++) {
        const int ri = (int)((T.torpack >> (16 * k)) & 0xFFFFull); // program record of this torus
        const uint32_t my = (T.slotpack >> (8 * k)) & 0xFFu;
        const PRec R = prec_load(prog + (size_t)ri * PREC);
        uint64_t meta; memcpy(&meta, &R.w[0], 8);
        const uint32_t mlo = (uint32_t)meta, mhi = (uint32_t)(meta >> 32);
        const double* q = T.tq + TQ_STRIDE * (my ? my - 1 : 0);
        uint64_t sg; { double sd = q[4]; memcpy(&sg, &sd, 8); }
        const int flip = (prec_w(R, 2) * prec_w(R, 3) < 0) ? -1 : 1;
        const bool mine = act & (my != 0);
        // roots in pairs; a NaN root fails every comparison
#pragma unroll 1
        for (int m = 0; m < 4; m += 2) {
            const uint32_t s0 = (uint32_t)(sg >> (2 * m)) & 3u, s1 = (uint32_t)(sg >> (2 * m + 2)) & 3u;
            const int o0 = -flip * (s0 == 1u ? 1 : -1), o1 = -flip * (s1 == 1u ? 1 : -1);
            const double y0 = q[m], y1 = q[m + 1];
            if (!RT_ANY(mine & ((y0 == y0) | (y1 == y1)))) continue; // this quadratic factor has no real roots for anybody
            frame_candidates<true>(G, P, F, R, mlo, mhi, sp, d, six_last, mine, y0, y1, o0, o1, s0 != 0u, s1 != 0u, h);
        }
    }
    rt_wave_sync();
    T.npend = 0; T.ntor = 0; T.torpack = 0; T.slotpack = 0;
}

// GENERIC = false: the program has no PK_GENERIC record (the caller checked); that code is then not compiled in, which
// is what keeps the register count of the nearest-hit kernel low.
template <bool GENERIC = true> RT_HD void find_hit(const Geom& G, const Prm& P, V3 sp, V3 d, int six_last, bool act, Hit& h) {
    h.l = INFINITY; h.surf = -1; h.orient = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    TorusQ T{ G.tq, 0, 0, 0ull, 0u };
#else
    double tq_host[TQ_STRIDE * TQ_SLOTS]; TorusQ T{ tq_host, 0, 0, 0ull, 0u };
#endif
    TorusQ TG = T; // generic tori (separate bookkeeping, same LDS: flushed before a frame torus is queued and vice versa)
    FrameQ F = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < G.nprog; i++) {
        const PRec R = prec_load(G.uprog + (size_t)i * PREC);
        uint64_t meta; memcpy(&meta, &R.w[0], 8);
        const uint32_t mlo = (uint32_t)meta, mhi = (uint32_t)(meta >> 32);
        const uint32_t kind = mlo & 15u;
        if (kind == PK_FRAME) {
            frame_torus_flush(G, P, F, G.uprog, sp, d, six_last, act, h, T); // queued tori belong to the frame that ends here
            V3 c0 = V3{ prec_w(R, 1), prec_w(R, 2), prec_w(R, 3) }, ax = V3{ prec_w(R, 4), prec_w(R, 5), prec_w(R, 6) }, x = sp - c0;
            F.xa = dot(x, ax); F.da = dot(d, ax);
            V3 u = F.xa * ax - x, v = F.da * ax - d;
            F.a = dot(v, v); F.hb = dot(u, v); F.uu = dot(u, u); F.xd = dot(x, d); F.xx = dot(x, x);
            F.ida = rt_rcp(F.da); F.ia = rt_rcp(F.a); F.sda = isgn(F.da);
            if (mlo & (1u << 16)) { // bounding box of the members: radius <= sqrt(w7), w8 <= axial <= w9.  Nobody gets there before its best hit: skip them all
                const double a_lo = prec_w(R, 8), a_hi = prec_w(R, 9);
                const double ta = (a_lo - F.xa) * F.ida, tb = (a_hi - F.xa) * F.ida;
                double lo = fmin(ta, tb), hi = fmax(ta, tb);
                if (RT_ANY(F.sda == 0)) { if (F.sda == 0) { const bool in = (F.xa >= a_lo) & (F.xa <= a_hi); lo = in ? -INFINITY : INFINITY; hi = in ? INFINITY : -INFINITY; } }
                const double disc = F.hb * F.hb - F.a * (F.uu - prec_w(R, 7)), sq = rt_sqrt(disc);
                lo = fmax(lo, (-F.hb - sq) * F.ia); hi = fmin(hi, (sq - F.hb) * F.ia); // NaN (tangent, or ray along the axis) leaves the bound as it is
                lo = fmax(lo, 0.0); hi = fmin(hi, h.l);
                const bool reach = act & (disc >= 0) & (lo <= hi);
                if (!RT_ANY(reach)) i += (int)((mlo >> 4) & 0xFFFu);
            }
            continue;
        }
        if (kind == PK_GROUP) { // members of the frame that lie in a shell (radius^2 >= w7) or beyond an axial coordinate (>= w8, <= w9):
            // a wave none of whose rays gets there before its best hit so far skips them (count in the meta word)
            if (T.npend) frame_torus_flush(G, P, F, G.uprog, sp, d, six_last, act, h, T); // hits of queued tori count
            const double te = h.l;
            const double r2e = (F.a * te + 2 * F.hb) * te + F.uu;       // radius^2 is convex along the ray: the largest value is at an end
            const double ae = F.xa + F.da * te;                          // NaN (0 * inf) fails every comparison: the ray counts as reaching
            const bool out = (fmax(F.uu, r2e) < prec_w(R, 7)) | (fmax(F.xa, ae) < prec_w(R, 8)) | (fmin(F.xa, ae) > prec_w(R, 9));
            if (!RT_ANY(act & !out)) i += (int)((mlo >> 4) & 0xFFFu);
            continue;
        }
        if (kind == PK_GENERIC) {
            if (GENERIC) {
                if (T.npend) frame_torus_flush(G, P, F, G.uprog, sp, d, six_last, act, h, T);
                generic_surface(G, P, sp, d, six_last, act, (int)((mlo >> 4) & 0xFFFu), h, TG);
            }
            continue;
        }
        double l0 = NAN, l1 = NAN; int o0 = 0, o1 = 0; bool ok0 = false, ok1 = false;
        if (kind == PK_PLANE) {
            l0 = (prec_w(R, 1) - F.xa) * F.ida; o0 = -(int)prec_w(R, 2) * F.sda; ok0 = F.sda != 0;
            frame_candidates<false>(G, P, F, R, mlo, mhi, sp, d, six_last, act, l0, l1, o0, o1, ok0, ok1, h);
            continue;
        } else if (kind == PK_CYL) {
            const double rad = prec_w(R, 1), c = F.uu - prec_w(R, 2), disc = F.hb * F.hb - F.a * c;
            const double sq = rt_sqrt(disc);
            l0 = (sq - F.hb) * F.ia; l1 = (-F.hb - sq) * F.ia;
            o0 = rad > 0 ? -1 : 1; o1 = -o0; ok0 = ok1 = sq > 0;
            if (RT_ANY(F.a == 0)) { // ray parallel to the axis: RayToCylinder.m:106-112
                if (F.a == 0) { l0 = l1 = (F.hb != 0) ? -0.5 * c / F.hb : NAN; o0 = o1 = -isgn(F.hb * rad); ok0 = o0 != 0; ok1 = false; }
            }
        } else if (kind == PK_SPHERE) {
            const double t = prec_w(R, 1), rad = prec_w(R, 2), B = F.xd - t * F.da, C = (F.xx - 2 * t * F.xa) + prec_w(R, 3), D = B * B - C;
            const double sq = rt_sqrt(D);
            l0 = sq - B; l1 = -B - sq;
            o0 = rad > 0 ? -1 : 1; o1 = -o0; ok0 = ok1 = sq > 0;
        } else if (kind == PK_QUAD) {
            const double al = prec_w(R, 1), ga = prec_w(R, 2), pa = prec_w(R, 3);
            const double aq = al * F.a + ga * F.da * F.da, b = 2 * (al * F.hb + ga * F.xa * F.da) + pa * F.da, c = al * F.uu + (ga * F.xa + pa) * F.xa + prec_w(R, 4);
            const double sq = rt_sqrt(b * b - 4 * aq * c), ia = 0.5 * rt_rcp(aq);
            l0 = (sq - b) * ia; l1 = (-b - sq) * ia;
            o0 = -1; o1 = 1; ok0 = ok1 = sq > 0; // d.(2pQ + P) = b + 2 l aq = +-sqrt(discriminant)
            const bool lin = (b != 0) & (fabs(4 * aq * c) < (100 * 2.220446049250313e-16) * (b * b));
            if (RT_ANY(lin | (aq == 0))) {
                if (lin) { l0 = -c / b; l1 = -b / aq; o0 = -isgn(b + 2 * l0 * aq); o1 = -isgn(b + 2 * l1 * aq); ok0 = o0 != 0; ok1 = o1 != 0; }
                else if (aq == 0) { ok0 = ok1 = false; }
            }
        } else { // PK_TORUS: RayToTorus.m:88-159; prefilter and quartic as in generic_surface(), from the frame's quantities
            const double t = prec_w(R, 1), R1 = prec_w(R, 2), r2 = prec_w(R, 3);
            const int n = (int)((mlo >> 4) & 0xFFFu);
            const double lmin = (six_last == n + P.base) ? P.min_travel : 0.0;
            // slab and annulus that hold the bounded part of the torus: w13 <= axial <= w14 (frame coordinate), radius^2 >= w15,
            // radius <= |R1| + |r2| - the torus' own extent cut down by its range bounds on the host; everything 1e-9 wider
            const double Ro = (fabs(R1) + fabs(r2)) * (1 + 1e-9), Ri2 = prec_w(R, 15);
            const double ta = (prec_w(R, 13) - F.xa) * F.ida, tb = (prec_w(R, 14) - F.xa) * F.ida;
            double lo = fmin(ta, tb), hi = fmax(ta, tb);
            if (RT_ANY(F.sda == 0)) { if (F.sda == 0) { const bool in = (F.xa >= prec_w(R, 13)) & (F.xa <= prec_w(R, 14)); lo = in ? -INFINITY : INFINITY; hi = in ? INFINITY : -INFINITY; } }
            const double disc = F.hb * F.hb - F.a * (F.uu - Ro * Ro), sq = rt_sqrt(disc);
            const bool qpos = F.a > 0;
            lo = qpos ? fmax(lo, (-F.hb - sq) * F.ia) : lo; hi = qpos ? fmin(hi, (sq - F.hb) * F.ia) : hi;
            lo = fmax(lo, lmin); hi = fmin(hi, h.l);
            bool maybe = act & (disc >= 0) & (lo <= hi);
            maybe = maybe & !(((F.a * lo + 2 * F.hb) * lo + F.uu < Ri2) & ((F.a * hi + 2 * F.hb) * hi + F.uu < Ri2)); // radius^2 is convex in l
            const double xa0 = F.xa - t;
            const unsigned long long mb = RT_BALLOT(maybe);
            if (mb != 0ull) {
                const int cnt = __builtin_popcountll(mb);
                if (GENERIC && TG.npend) torus_flush(G, P, sp, d, six_last, act, h, TG);
                if (T.npend + cnt > TQ_SLOTS || T.ntor == TQ_TORI) frame_torus_flush(G, P, F, G.uprog, sp, d, six_last, act, h, T);
                if (maybe) { // depressed quartic about the point of closest approach to the torus centre
                    const int slot = T.npend + rt_rank(mb);
                    V3 x = sp - V3{ prec_w(R, 4), prec_w(R, 5), prec_w(R, 6) };
                    const double t0 = -dot(x, d);
                    V3 xs = x + t0 * d;
                    const double X2 = dot(xs, xs), xas = xa0 + t0 * F.da, K = X2 + R1 * R1 - r2 * r2, R4 = 4 * R1 * R1;
                    double* q = T.tq + TQ_STRIDE * slot;
                    q[0] = 2 * K - R4 * F.a; q[1] = 2 * R4 * xas * F.da; q[2] = K * K - R4 * (X2 - xas * xas); q[3] = t0; q[4] = lo; q[5] = hi;
                    T.slotpack |= (uint32_t)(slot + 1) << (8 * T.ntor);
                }
                T.torpack |= (unsigned long long)(unsigned)i << (16 * T.ntor);
                T.ntor++; T.npend += cnt;
            }
            continue;
        }
        frame_candidates<true>(G, P, F, R, mlo, mhi, sp, d, six_last, act, l0, l1, o0, o1, ok0, ok1, h);
    }
    frame_torus_flush(G, P, F, G.uprog, sp, d, six_last, act, h, T);
    if (GENERIC) torus_flush(G, P, sp, d, six_last, act, h, TG);
}

// ---------------------------------------------------------------------------------------------- frame program (host)
// Host: surface table + public bounds programs -> the records find_hit() walks.  frames = false gives one
// PK_GENERIC record per surface in list order (the reference's formulas throughout; used for COMPAT_PY_ROOT2).
// Tolerances: a surface joins a frame only if the misfit (angle between axes, distance of its centre from the
// frame's line, deviation of a quadric from one of revolution) is below 1e-13 relative, far inside the 1e-9 the
// parity tests allow; anything else stays generic.
namespace hostprog {
struct V { double x, y, z; };
static inline V mk(const double* p) { return V{ p[0], p[1], p[2] }; }
static inline double dt(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V sub(V a, V b) { return V{ a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline V add(V a, V b) { return V{ a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline V mul(double s, V a) { return V{ s * a.x, s * a.y, s * a.z }; }
static inline V crs(V a, V b) { return V{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static inline double nrm(V a) { return sqrt(dt(a, a)); }
struct Frame { V c0, ax; int members; };
struct Range { bool used; double lo, hi; bool ilo, ihi; double t; };
constexpr double TOL = 1e-13;
static inline bool parallel(V a, V b, double& sigma) { sigma = dt(a, b) >= 0 ? 1.0 : -1.0; return nrm(crs(a, b)) <= TOL; }
static inline bool on_line(const Frame& f, V p) { V dp = sub(p, f.c0); V perp = sub(dp, mul(dt(dp, f.ax), f.ax)); return nrm(perp) <= 1e-12 * (1.0 + nrm(dp)); }
static inline V mat(const double* S, V v) { return V{ S[0] * v.x + S[1] * v.y + S[2] * v.z, S[3] * v.x + S[4] * v.y + S[5] * v.z, S[6] * v.x + S[7] * v.y + S[8] * v.z }; }
// x'Sx + b.x (S symmetric) as alpha |x_perp|^2 + gamma xa^2 + pa xa + k0 in frame coordinates, if it is of that form
static inline bool revolution_form(const Frame& f, const double* S, V b, double& alpha, double& gamma, double& pa, double& k0) {
    V e1 = crs(f.ax, fabs(f.ax.x) < 0.9 ? V{ 1, 0, 0 } : V{ 0, 1, 0 }); e1 = mul(1.0 / nrm(e1), e1);
    V e2 = crs(f.ax, e1);
    double scale = 0; for (int i = 0; i < 9; i++) scale = fmax(scale, fabs(S[i]));
    if (scale == 0) return false;
    V s1 = mat(S, e1), s2 = mat(S, e2), sa = mat(S, f.ax);
    alpha = dt(e1, s1); gamma = dt(f.ax, sa);
    if (nrm(sub(s1, mul(alpha, e1))) > TOL * scale || nrm(sub(s2, mul(alpha, e2))) > TOL * scale || nrm(sub(sa, mul(gamma, f.ax))) > TOL * scale) return false;
    V w = add(mul(2.0, mat(S, f.c0)), b); // gradient on the axis line must be axial
    double wa = dt(w, f.ax);
    if (nrm(sub(w, mul(wa, f.ax))) > TOL * (scale * (1.0 + nrm(f.c0)) + nrm(b))) return false;
    pa = wa; k0 = dt(f.c0, mat(S, f.c0)) + dt(b, f.c0);
    return true;
}
} // namespace hostprog

RT_HOST static inline void build_program(const double* table, int S, const double* atoms, const uint32_t* code, bool frames_on,
                                         std::vector<double>& recs, const std::vector<uint32_t>& smeta, std::vector<double>& prog) {
    using namespace hostprog;
    prog.clear();
    auto put = [&](uint64_t meta, const double* w, int nw) -> size_t {
        size_t at = prog.size(); prog.resize(at + PREC, 0.0);
        memcpy(&prog[at], &meta, 8);
        for (int j = 0; j < nw; j++) prog[at + 1 + j] = w[j];
        return at;
    };
    std::vector<Frame> frames; std::vector<int> fr(S > 0 ? S : 1, -1); std::vector<double> sg(S > 0 ? S : 1, 1.0);
    if (frames_on) {
        // frames come from the surfaces that have an axis line of their own
        for (int s = 0; s < S; s++) {
            const double* r = table + (size_t)s * ROW; const int shape = (int)r[0];
            if (shape != SH_CYL && shape != SH_TORUS) continue;
            V c = mk(r + 2), ax = mk(r + 5); double sig; bool found = false;
            if (fabs(nrm(ax) - 1.0) > 1e-12) continue;
            for (size_t f = 0; f < frames.size() && !found; f++) if (parallel(frames[f].ax, ax, sig) && on_line(frames[f], c)) { found = true; fr[s] = (int)f; sg[s] = sig; }
            if (!found) { frames.push_back(Frame{ c, ax, 0 }); fr[s] = (int)frames.size() - 1; }
        }
        for (int s = 0; s < S; s++) {
            const double* r = table + (size_t)s * ROW; const int shape = (int)r[0]; double sig;
            if (shape == SH_PLANE) { V nn = mk(r + 5); if (fabs(nrm(nn) - 1.0) <= 1e-12) for (size_t f = 0; f < frames.size() && fr[s] < 0; f++) if (parallel(frames[f].ax, nn, sig)) { fr[s] = (int)f; sg[s] = sig; } }
            else if (shape == SH_SPHERE) { for (size_t f = 0; f < frames.size() && fr[s] < 0; f++) if (on_line(frames[f], mk(r + 2))) fr[s] = (int)f; }
            else if (shape == SH_QUAD) {
                const double* Q = r + 2; bool sym = Q[1] == Q[3] && Q[2] == Q[6] && Q[5] == Q[7];
                double al, ga, pa, k0;
                for (size_t f = 0; sym && f < frames.size() && fr[s] < 0; f++) if (revolution_form(frames[f], Q, mk(r + 11), al, ga, pa, k0)) fr[s] = (int)f;
            }
        }
        for (int s = 0; s < S; s++) if (fr[s] >= 0) frames[fr[s]].members++;
        for (int s = 0; s < S; s++) if (fr[s] >= 0 && frames[fr[s]].members < 2) fr[s] = -1; // a frame must pay for itself
    }
    // one atom as a one-sided bound on a frame coordinate, if it is one
    auto atom_range = [&](uint32_t ins, const Frame& f, int& ck, double& t, bool& upper, double& bound, bool& incl) -> bool {
        const double* a = atoms + (size_t)(ins & 0xFFFFFFu) * ATOM; const int kind = (int)a[0], cmp = (int)a[1]; const double rhs = a[2];
        incl = (cmp == 1 || cmp == 3); upper = (cmp == 0 || cmp == 1); t = 0.0;
        if (cmp <= 3 && (kind == 3 || kind == 1)) { // linear: c.p cmp rhs
            V c = kind == 3 ? V{ (int)a[3] == 0 ? 1.0 : 0.0, (int)a[3] == 1 ? 1.0 : 0.0, (int)a[3] == 2 ? 1.0 : 0.0 } : V{ a[9], a[10], a[11] };
            const double lam = dt(c, f.ax);
            if (lam != 0 && nrm(sub(c, mul(lam, f.ax))) <= TOL * nrm(c)) { // c.p = lam (xa + c0.a^)
                bound = rhs / lam - dt(f.c0, f.ax); if (lam < 0) upper = !upper;
                ck = FA_AXIAL; return true;
            }
        } else if (cmp <= 3 && kind == 0) { // quadratic: p'Ap + b.p cmp rhs
            const double A[9] = { a[3], 0.5 * a[6], 0.5 * a[7], 0.5 * a[6], a[4], 0.5 * a[8], 0.5 * a[7], 0.5 * a[8], a[5] }; V b = V{ a[9], a[10], a[11] };
            double al, ga, pa, k0;
            if (revolution_form(f, A, b, al, ga, pa, k0)) {
                double scale = fmax(fabs(al), fabs(ga));
                if (al != 0 && fabs(ga) <= TOL * scale && fabs(pa) <= TOL * scale * (1.0 + nrm(f.c0))) { // al |x_perp|^2 + k0
                    if (al < 0) upper = !upper;
                    bound = (rhs - k0) / al; ck = FA_RADIAL2; return true;
                } else if (al != 0 && fabs(ga - al) <= TOL * scale) { // al (|x_perp|^2 + xa^2) + pa xa + k0 = al |x - t a^|^2 + k0 - al t^2
                    t = -pa / (2 * al); if (al < 0) upper = !upper;
                    bound = (rhs - k0) / al + t * t; ck = FA_DIST2; return true;
                }
            }
        }
        return false;
    };
    // Superset of where a bounds program can be true, as a box in frame coordinates (radius^2 <= r2, lo <= axial <= hi):
    // the program evaluated over boxes - atom: its half-space or everything, AND: intersection, OR: hull, anything else: everything.
    struct Box { double r2, lo, hi; };
    auto program_box = [&](int s, const Frame& f) {
        const double* r = table + (size_t)s * ROW; const int off = (int)r[27], len = (int)r[28];
        const Box all{ INFINITY, -INFINITY, INFINITY };
        std::vector<Box> st;
        for (int i = off; i < off + len; i++) {
            const uint32_t op = code[i] >> 24;
            if (op == 1) {
                Box b = all; int ck; double t, bound; bool upper, incl;
                if (atom_range(code[i], f, ck, t, upper, bound, incl) && upper) {
                    if (ck == FA_AXIAL) b.hi = bound;
                    else if (ck == FA_RADIAL2) b.r2 = bound;
                    else { b.r2 = bound; const double d = sqrt(fmax(bound, 0.0)); b.lo = t - d; b.hi = t + d; }
                } else if (atom_range(code[i], f, ck, t, upper, bound, incl) && ck == FA_AXIAL) b.lo = bound;
                st.push_back(b);
            } else if ((op == 2 || op == 3) && st.size() >= 2) {
                Box y = st.back(); st.pop_back(); Box& x = st.back();
                if (op == 2) { x.r2 = fmin(x.r2, y.r2); x.lo = fmax(x.lo, y.lo); x.hi = fmin(x.hi, y.hi); }
                else { x.r2 = fmax(x.r2, y.r2); x.lo = fmin(x.lo, y.lo); x.hi = fmax(x.hi, y.hi); }
            } else if (op == 5) st.push_back(all);
            else if (op == 6) st.push_back(Box{ 0.0, INFINITY, -INFINITY });
            else { st.clear(); st.push_back(all); break; } // NOT, XOR: no claim
        }
        return st.size() == 1 ? st[0] : all;
    };
    // bounds of a frame member: conjunction of ranges of frame coordinates (inline) + residual generic program
    auto member_bounds = [&](int s, const Frame& f, uint64_t& meta, double* aw, int max_atoms) {
        const double* r = table + (size_t)s * ROW; const int off = (int)r[27], len = (int)r[28];
        std::vector<uint32_t> conj; bool pure = true; int depth = 0;
        for (int i = off; i < off + len && pure; i++) { const uint32_t op = code[i] >> 24; if (op == 1) { conj.push_back(code[i]); depth++; } else if (op == 2) depth--; else pure = false; }
        if (len == 0) return;
        if (!pure || depth != 1) { meta |= ((uint64_t)smeta[4 * s + 2] << 32) | ((uint64_t)smeta[4 * s + 3] << 52); return; }
        std::vector<Range> rg; std::vector<int> rkind; std::vector<uint32_t> rest;
        auto narrow = [&](int kind, double t, bool upper, double bound, bool incl) {
            size_t j = 0; for (; j < rg.size(); j++) if (rkind[j] == kind && (kind != FA_DIST2 || rg[j].t == t)) break;
            if (j == rg.size()) { if ((int)rg.size() == max_atoms) return false; rg.push_back(Range{ true, -INFINITY, INFINITY, false, false, t }); rkind.push_back(kind); }
            Range& R = rg[j];
            if (upper) { if (bound < R.hi || (bound == R.hi && !incl)) { R.hi = bound; R.ihi = incl; } }
            else { if (bound > R.lo || (bound == R.lo && !incl)) { R.lo = bound; R.ilo = incl; } }
            return true;
        };
        for (uint32_t ins : conj) {
            int ck; double t, bound; bool upper, incl;
            bool done = atom_range(ins, f, ck, t, upper, bound, incl) && narrow(ck, t, upper, bound, incl);
            if (!done) rest.push_back(ins);
        }
        meta |= (uint64_t)rg.size() << 16;
        for (size_t j = 0; j < rg.size(); j++) {
            meta |= (uint64_t)((uint32_t)rkind[j]) << (20 + 4 * j);
            aw[3 * j] = rg[j].ilo ? nextafter(rg[j].lo, -INFINITY) : rg[j].lo; aw[3 * j + 1] = rg[j].ihi ? nextafter(rg[j].hi, INFINITY) : rg[j].hi; aw[3 * j + 2] = rg[j].t;
        }
        if (!rest.empty()) {
            std::vector<uint32_t> c2;
            for (size_t j = 0; j < rest.size(); j++) { c2.push_back(rest[j]); if (j) c2.push_back(2u << 24); }
            const size_t first = emit_bounds_program(c2.data(), (int)c2.size(), atoms, recs), cnt = recs.size() / BREC - first;
            meta |= ((uint64_t)first << 32) | ((uint64_t)cnt << 52);
        }
    };
    // extent of a member in frame coordinates from its shape and inline ranges (a superset is fine): radius^2 <= r2, lo <= axial <= hi
    auto member_extent = [&](uint64_t meta, const double* w, const Box& pb, double& r2, double& lo, double& hi) {
        const uint32_t mlo = (uint32_t)meta, kind = mlo & 15u, na = (mlo >> 16) & 3u;
        double A_lo = pb.lo, A_hi = pb.hi, R_hi = pb.r2; (void)na;
        // rho^2(x) = -(ga x^2 + pa x + Rp) / al of a quadric of revolution: maximum over [A_lo, A_hi]
        auto quad_max = [&](double al, double ga, double pa, double Rp) {
            if (al == 0) return (double)INFINITY;
            const double qa = -ga / al, qb = -pa / al, qc = -Rp / al; // rho^2 = qa x^2 + qb x + qc
            double m = -INFINITY;
            const double ends[2] = { A_lo, A_hi };
            for (double x : ends) { if (std::isinf(x)) { if (qa > 0 || (qa == 0 && qb != 0)) return (double)INFINITY; } else m = fmax(m, (qa * x + qb) * x + qc); }
            if (qa < 0) { const double xv = -qb / (2 * qa); if (xv >= A_lo && xv <= A_hi) m = fmax(m, (qa * xv + qb) * xv + qc); }
            else if (qa == 0 && qb == 0) m = fmax(m, qc);
            return m;
        };
        lo = A_lo; hi = A_hi; r2 = INFINITY;
        if (kind == PK_PLANE) { lo = fmax(lo, w[0]); hi = fmin(hi, w[0]); }
        else if (kind == PK_CYL) r2 = w[1];
        else if (kind == PK_SPHERE) { const double t = w[0], rr = fabs(w[1]); lo = fmax(lo, t - rr); hi = fmin(hi, t + rr); A_lo = lo; A_hi = hi; r2 = quad_max(1.0, 1.0, -2 * t, w[2]); }
        else if (kind == PK_TORUS) { const double t = w[0], R1 = fabs(w[1]), rr = fabs(w[2]); lo = fmax(lo, t - rr); hi = fmin(hi, t + rr); r2 = (R1 + rr) * (R1 + rr); }
        else if (kind == PK_QUAD) r2 = quad_max(w[0], w[1], w[2], w[3]);
        r2 = fmin(r2, R_hi);
    };
    for (size_t f = 0; f < frames.size(); f++) {
        if (frames[f].members < 2) continue;
        const Frame& F = frames[f];
        size_t frec;
        { double w[6] = { F.c0.x, F.c0.y, F.c0.z, F.ax.x, F.ax.y, F.ax.z }; frec = put((uint64_t)PK_FRAME, w, 6); }
        double box_r2 = 0, box_lo = INFINITY, box_hi = -INFINITY; int nmem = 0;
        struct Mem { uint64_t meta; double w[PREC - 1]; double r2min, r2max, lo, hi; };
        std::vector<Mem> mem;
        for (int s = 0; s < S; s++) {
            if (fr[s] != (int)f) continue;
            const double* r = table + (size_t)s * ROW; const int shape = (int)r[0];
            Mem M; double* w = M.w; for (int j = 0; j < PREC - 1; j++) w[j] = 0.0;
            uint64_t meta = (uint64_t)((uint32_t)s << 4);
            if (shape == SH_PLANE) { meta |= PK_PLANE; w[0] = dt(sub(mk(r + 2), F.c0), F.ax); w[1] = sg[s]; }
            else if (shape == SH_CYL) { meta |= PK_CYL; w[0] = r[8]; w[1] = r[8] * r[8]; }
            else if (shape == SH_SPHERE) { meta |= PK_SPHERE; const double t = dt(sub(mk(r + 2), F.c0), F.ax); w[0] = t; w[1] = r[5]; w[2] = t * t - r[5] * r[5]; }
            else if (shape == SH_TORUS) { meta |= PK_TORUS; w[0] = dt(sub(mk(r + 2), F.c0), F.ax); w[1] = r[8]; w[2] = r[9]; w[3] = r[2]; w[4] = r[3]; w[5] = r[4]; }
            else { meta |= PK_QUAD; double al, ga, pa, k0; revolution_form(F, r + 2, mk(r + 11), al, ga, pa, k0); w[0] = al; w[1] = ga; w[2] = pa; w[3] = k0 + r[14]; }
            member_bounds(s, F, meta, w + 6, shape == SH_TORUS ? 2 : 3);
            if (shape == SH_TORUS) { // prefilter slab / inner radius: the torus' own extent cut down by its range bounds
                const Box pb = program_box(s, F);
                const double t = w[0], R1 = fabs(w[1]), rr = fabs(w[2]) * (1 + 1e-9);
                double alo = fmax(t - rr, pb.lo), ahi = fmin(t + rr, pb.hi), ri = fmax(R1 - rr, 0.0), ri2 = ri * ri;
                const uint32_t na = (uint32_t)(meta >> 16) & 3u;
                for (uint32_t j = 0; j < na; j++) if (((uint32_t)(meta >> (20 + 4 * j)) & 3u) == FA_RADIAL2) ri2 = fmax(ri2, w[6 + 3 * j]); // lower bound of a radial range
                const double pad = 1e-9 * (fabs(alo) + fabs(ahi) + rr);
                w[12] = alo - pad; w[13] = ahi + pad; w[14] = ri2 * (1 - 1e-9);
            }
            M.meta = meta;
            member_extent(meta, w, program_box(s, F), M.r2max, M.lo, M.hi);
            // smallest radius^2 a hit on this member can have: its shape over [lo, hi], raised by the lower end of an inline radial range
            {
                const uint32_t kind = (uint32_t)meta & 15u;
                auto quad_min = [&](double al, double ga, double pa, double Rp) { // rho^2 = -(ga x^2 + pa x + Rp) / al over [M.lo, M.hi]
                    if (al == 0) return 0.0;
                    const double qa = -ga / al, qb = -pa / al, qc = -Rp / al;
                    double m = INFINITY;
                    const double ends[2] = { M.lo, M.hi };
                    for (double x : ends) { if (std::isinf(x)) return 0.0; m = fmin(m, (qa * x + qb) * x + qc); }
                    if (qa > 0) { const double xv = -qb / (2 * qa); if (xv >= M.lo && xv <= M.hi) m = fmin(m, (qa * xv + qb) * xv + qc); }
                    return m;
                };
                double r2 = 0.0;
                if (kind == PK_CYL) r2 = w[1];
                else if (kind == PK_SPHERE) r2 = quad_min(1.0, 1.0, -2 * w[0], w[2]);
                else if (kind == PK_QUAD) r2 = quad_min(w[0], w[1], w[2], w[3]);
                else if (kind == PK_TORUS) { const double ri = fmax(fabs(w[1]) - fabs(w[2]), 0.0); r2 = ri * ri; }
                const uint32_t na = (uint32_t)(meta >> 16) & 3u;
                for (uint32_t j = 0; j < na; j++) if (((uint32_t)(meta >> (20 + 4 * j)) & 3u) == FA_RADIAL2) r2 = fmax(r2, w[6 + 3 * j]);
                M.r2min = (r2 > 0 && !std::isnan(r2)) ? r2 : 0.0;
            }
            if (M.lo > M.hi) continue; // bounds that nothing satisfies (e.g. z > 120 & z < -20): no record, the surface can never be hit
            mem.push_back(M);
            box_r2 = fmax(box_r2, M.r2max); box_lo = fmin(box_lo, M.lo); box_hi = fmax(box_hi, M.hi);
        }
        // Order: the members in list order, except those far out - radially (a shell) or along the axis (above, below) -
        // which go behind GROUP records, nested from the inside out.  "Far out" is measured against the median extent of
        // the frame's members, so that a group is skipped by rays that stay where most of the geometry is.
        {
            auto median = [](std::vector<double> v) { if (v.empty()) return (double)NAN; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            std::vector<int> core, shell, above, below;
            std::vector<char> used(mem.size(), 0);
            const bool groups_on = frames_on && getenv("RT_NO_GROUPS") == nullptr;
            if (groups_on && mem.size() >= 6) {
                std::vector<double> v; for (const Mem& M : mem) v.push_back(M.r2max);
                const double med_r2 = median(v);
                for (size_t j = 0; j < mem.size(); j++) if (mem[j].r2min > 0 && mem[j].r2min >= med_r2 * (1 + 1e-9)) { shell.push_back((int)j); used[j] = 1; }
                if (shell.size() < 2) { for (int j : shell) used[j] = 0; shell.clear(); }
                v.clear(); for (size_t j = 0; j < mem.size(); j++) if (!used[j]) v.push_back(mem[j].hi);
                const double med_hi = median(v);
                for (size_t j = 0; j < mem.size(); j++) if (!used[j] && std::isfinite(mem[j].lo) && mem[j].lo > med_hi) { above.push_back((int)j); used[j] = 1; }
                if (above.size() < 2) { for (int j : above) used[j] = 0; above.clear(); }
                v.clear(); for (size_t j = 0; j < mem.size(); j++) if (!used[j]) v.push_back(mem[j].lo);
                const double med_lo = median(v);
                for (size_t j = 0; j < mem.size(); j++) if (!used[j] && std::isfinite(mem[j].hi) && mem[j].hi < med_lo) { below.push_back((int)j); used[j] = 1; }
                if (below.size() < 2) { for (int j : below) used[j] = 0; below.clear(); }
            }
            for (size_t j = 0; j < mem.size(); j++) if (!used[j]) core.push_back((int)j);
            // Tori last among the core members: a torus is only *queued* where its record stands, with a window of ray parameters
            // that ends at the best hit so far - behind the other members that window is at its shortest, and fewer rays queue a
            // quartic at all.  (The order of the records decides nothing else: ties go by surface number, root_wins().)
            if (getenv("RT_TORI_IN_PLACE") == nullptr) std::stable_partition(core.begin(), core.end(), [&](int j) { return ((uint32_t)mem[j].meta & 15u) != PK_TORUS; });
            for (int j : core) { put(mem[j].meta, mem[j].w, PREC - 1); nmem++; }
            // kind 0: shell (key = r2min, ascending), 1: above (key = lo, ascending), 2: below (key = hi, descending)
            auto chain = [&](std::vector<int>& g, int kind) {
                auto key = [&](int j) { return kind == 0 ? mem[j].r2min : (kind == 1 ? mem[j].lo : -mem[j].hi); };
                std::stable_sort(g.begin(), g.end(), [&](int a_, int b_) { return key(a_) < key(b_); });
                // a new (nested) group wherever the key grows noticeably and at least two members remain
                std::vector<size_t> starts;
                double span = 0; { double a_ = INFINITY, b_ = -INFINITY; for (const Mem& M : mem) { if (std::isfinite(M.lo)) a_ = fmin(a_, M.lo); if (std::isfinite(M.hi)) b_ = fmax(b_, M.hi); } if (b_ > a_) span = b_ - a_; }
                double gkey = 0;
                for (size_t p = 0; p < g.size(); p++) {
                    const double k = key(g[p]);
                    const bool first = p == 0, grows = !first && (kind == 0 ? k > 1.2 * gkey : k - gkey > 0.05 * span);
                    if ((first || grows) && g.size() - p >= 2) { starts.push_back(p); gkey = k; }
                }
                size_t si = 0;
                for (size_t p = 0; p < g.size(); p++) {
                    if (si < starts.size() && starts[si] == p) {
                        const size_t later_groups = starts.size() - si - 1, count = (g.size() - p) + later_groups;
                        double w[PREC - 1]; for (int j = 0; j < PREC - 1; j++) w[j] = 0.0;
                        const double k = key(g[p]);
                        w[6] = -INFINITY; w[7] = -INFINITY; w[8] = INFINITY;
                        if (kind == 0) w[6] = k * (1 - 1e-9) - 1e-12;
                        else if (kind == 1) w[7] = k - 1e-9 * (1 + fabs(k));
                        else w[8] = -k + 1e-9 * (1 + fabs(k));
                        put((uint64_t)PK_GROUP | ((uint64_t)count << 4), w, PREC - 1); nmem++;
                        si++;
                    }
                    put(mem[g[p]].meta, mem[g[p]].w, PREC - 1); nmem++;
                }
            };
            chain(shell, 0); chain(above, 1); chain(below, 2);
        }
        if (nmem < 4096 && !(std::isinf(box_r2) && std::isinf(box_lo) && std::isinf(box_hi)) && !std::isnan(box_r2)) { // bounding box of the frame's members, a little wider than computed
            uint64_t meta = (uint64_t)PK_FRAME | ((uint64_t)nmem << 4) | (1ull << 16);
            memcpy(&prog[frec], &meta, 8);
            const double span = std::isinf(box_hi - box_lo) ? 0.0 : (box_hi - box_lo);
            prog[frec + 7] = box_r2 * (1 + 1e-9) + 1e-12; prog[frec + 8] = box_lo - 1e-9 * (fabs(box_lo) + span) - 1e-12; prog[frec + 9] = box_hi + 1e-9 * (fabs(box_hi) + span) + 1e-12;
        }
    }
    for (int s = 0; s < S; s++) if (fr[s] < 0) put((uint64_t)PK_GENERIC | ((uint64_t)(uint32_t)s << 4), nullptr, 0);
    prog.resize(prog.size() + PREC, 0.0); // the kernel may prefetch one record past the end
}

// geometric normal of surface row at p, already multiplied by the orientation sign (what RayToX returns as
// surface_normals): RayToPlane.py:43, RayToSphere.py:61-63, RayToCylinder.py:120-135, RayToQuadsurface.m:116-122,
// RayToTorus.m:149-158
RT_HD V3 surface_normal(const double* row, V3 p, V3 d, int orient) {
    const double* g = row + 2;
    int shape = (int)row[0];
    double o = (double)orient;
    if (shape == SH_PLANE) { V3 pn = ld3(g + 3); return (-(double)isgn(dot(d, pn))) * pn; }
    if (shape == SH_SPHERE) { return (o / g[3]) * (p - ld3(g)); }
    if (shape == SH_CYL) { V3 ax = ld3(g + 3), y = p - ld3(g); double ya = dot(y, ax); V3 w = (1.0 / g[6]) * (ya * ax - y); return (-o) * w; }
    if (shape == SH_QUAD) {
        const double* Q = g;
        V3 n = { 2 * (p.x * Q[0] + p.y * Q[3] + p.z * Q[6]) + g[9], 2 * (p.x * Q[1] + p.y * Q[4] + p.z * Q[7]) + g[10], 2 * (p.x * Q[2] + p.y * Q[5] + p.z * Q[8]) + g[11] };
        return o * normalized_if(n);
    }
    V3 ax = ld3(g + 3), xx = p - ld3(g);
    double xxa = dot(xx, ax); V3 yv = xxa * ax, uu = xx - yv;
    double un = sqrt(dot(uu, uu));
    return (o / g[7]) * (uu * (1 - g[6] / un) + yv);
}

// ---------------------------------------------------------------------------------------------- Fresnel
// RefractionReflectionAtInterface.py:46-222 / .m:43-229 in Mueller form.  With hs = (s0+s1)/2, hp = (s0-s1)/2 in the
// plane-of-incidence basis, the reference's 3x2 complex amplitude matrix collapses to
//   s0' = hs|cs|^2 + hp|cp|^2,  s1' = hs|cs|^2 - hp|cp|^2,  s2' + i s3' = (s2 + i s3) conj(cs) cp
// for (cs,cp) = (rs,rp) or (ts,tp), ts = sqrt(|1-|rs|^2|), tp likewise (.py:145-146).  Valid for physical Stokes
// vectors (s0 >= |polarised part|).
struct FresnelCore {
    V3 d, n, pol, inorm;          // normalised direction / normal, child polref, interface normal (z of the local basis)
    double s0, s1, s2, s3;        // Stokes vector rotated into the plane-of-incidence basis
    double cos_i, sin_i, sin_t, cos_t;
    double Rs, Rp, cr, ci, Ts, Tp, tst;
    bool goodhit;
};
RT_HD void fresnel_core(const Ray10& in, V3 nrm, double n1, double n2, FresnelCore& c) {
    c.d = normalized_if(in.d);
    c.n = normalized_if(nrm);
    V3 pol = in.pol - dot(in.pol, c.d) * c.d;
    pol = normalized_if(pol);
    V3 md = -c.d;
    c.cos_i = dot(md, c.n);
    c.goodhit = c.cos_i > 0;
    c.inorm = cross(md, c.n);
    const double in2 = dot(c.inorm, c.inorm), irs = in2 > 0 ? rt_rsqrt(in2) : 0.0;
    c.sin_i = in2 * irs;
    c.s0 = in.s0; c.s1 = in.s1; c.s2 = in.s2; c.s3 = in.s3; c.pol = pol;
    if (c.sin_i > 0) {
        c.inorm = irs * c.inorm;
        double c_rot = dot(c.inorm, pol), s_rot = dot(cross(c.inorm, pol), c.d);
        double c2 = c_rot * c_rot - s_rot * s_rot, sn2 = 2 * c_rot * s_rot;
        c.s1 = in.s1 * c2 - in.s2 * sn2; c.s2 = in.s1 * sn2 + in.s2 * c2;
        c.pol = c.inorm;
    }
    c.sin_t = c.sin_i * n1 / n2;
    double arg = 1 - c.sin_t * c.sin_t;
    c.cos_t = 0;
    if (isinf(n2)) { c.Rs = 1; c.Rp = 1; c.cr = -1; c.ci = 0; c.Ts = 0; c.Tp = 0; c.tst = 0; c.cos_t = 1; }
    else if (arg >= 0) {
        c.cos_t = rt_fsqrt(arg);
        double a1 = n1 * c.cos_i, a2 = n2 * c.cos_t, b1 = n1 * c.cos_t, b2 = n2 * c.cos_i;
        double rs = rt_fdiv(a1 - a2, a1 + a2), rp = -rt_fdiv(b1 - b2, b1 + b2);
        c.Rs = rs * rs; c.Rp = rp * rp; c.cr = rs * rp; c.ci = 0; c.Ts = fabs(1 - c.Rs); c.Tp = fabs(1 - c.Rp); c.tst = rt_fsqrt(c.Ts * c.Tp);
    } else { // total internal reflection: |rs| = |rp| = 1, only the relative phase matters
        double kap = rt_fsqrt(-arg), A = n1 * c.cos_i, B = n2 * kap, C = n1 * kap, Dd = n2 * c.cos_i;
        double ia = rt_fdiv(1.0, A * A + B * B), ic = rt_fdiv(1.0, Dd * Dd + C * C);
        double rsr = (A * A - B * B) * ia, rsi = -2 * A * B * ia, rpr = (Dd * Dd - C * C) * ic, rpi = -2 * C * Dd * ic;
        c.Rs = 1; c.Rp = 1; c.Ts = 0; c.Tp = 0; c.tst = 0;
        c.cr = rsr * rpr + rsi * rpi; c.ci = rsr * rpi - rsi * rpr; // conj(rs) * rp
        if (!(arg < 0)) { c.Rs = c.Rp = c.cr = c.ci = NAN; } // NaN input
    }
}
RT_HD double fresnel_refl_s0(const FresnelCore& c) { return c.goodhit ? 0.5 * (c.s0 + c.s1) * c.Rs + 0.5 * (c.s0 - c.s1) * c.Rp : 0.0; }
RT_HD double fresnel_refr_s0(const FresnelCore& c) { return c.goodhit ? 0.5 * (c.s0 + c.s1) * c.Ts + 0.5 * (c.s0 - c.s1) * c.Tp : c.s0; }
// new y axis of the local basis; false if the hit is bad or the incidence normal (no plane of incidence)
RT_HD bool fresnel_yaxis(const FresnelCore& c, V3& ny) {
    ny = cross(c.n, c.inorm);
    double nyn = dot(ny, ny);
    bool goodcut = c.goodhit && nyn > 0;
    if (goodcut) ny = rt_rsqrt(nyn) * ny;
    return goodcut;
}
RT_HD void fresnel_reflected(const FresnelCore& c, Ray10& refl) {
    refl.d = -c.d; refl.pol = c.pol; refl.s0 = refl.s1 = refl.s2 = refl.s3 = 0;
    if (c.goodhit) {
        double hs = 0.5 * (c.s0 + c.s1), hp = 0.5 * (c.s0 - c.s1);
        refl.s0 = hs * c.Rs + hp * c.Rp; refl.s1 = hs * c.Rs - hp * c.Rp; refl.s2 = c.s2 * c.cr - c.s3 * c.ci; refl.s3 = c.s2 * c.ci + c.s3 * c.cr;
    }
    V3 ny;
    if (fresnel_yaxis(c, ny)) refl.d = c.cos_i * c.n - c.sin_i * ny;
}
RT_HD void fresnel_refracted(const FresnelCore& c, double tir, Ray10& refr) {
    refr.d = c.d; refr.pol = c.pol; refr.s0 = c.s0; refr.s1 = c.s1; refr.s2 = c.s2; refr.s3 = c.s3;
    if (c.goodhit) {
        double hs = 0.5 * (c.s0 + c.s1), hp = 0.5 * (c.s0 - c.s1);
        refr.s0 = hs * c.Ts + hp * c.Tp; refr.s1 = hs * c.Ts - hp * c.Tp; refr.s2 = c.s2 * c.tst; refr.s3 = c.s3 * c.tst;
    }
    V3 ny;
    if (fresnel_yaxis(c, ny)) {
        bool tircut = c.sin_t >= 1;
        if (!tircut) refr.d = (-c.cos_t) * c.n - c.sin_t * ny;
        else if (tir < 0) refr.d = -ny;
        else { Ray10 refl; fresnel_reflected(c, refl); refr.d = refl.d; refr.s0 = refl.s0 * tir; refr.s1 = refl.s1 * tir; refr.s2 = refl.s2 * tir; refr.s3 = refl.s3 * tir; }
    }
}
RT_HD void fresnel(const Ray10& in, V3 nrm, double n1, double n2, double tir, Ray10& refr, Ray10& refl) {
    FresnelCore c; fresnel_core(in, nrm, n1, n2, c);
    fresnel_refracted(c, tir, refr); fresnel_reflected(c, refl);
}

// ---------------------------------------------------------------------------------------------- UNIFIED
// UnifiedReflectorModel.m:48-320 (python :52-272 where it runs).  One Fresnel evaluation per loop pass: pass type
// FACET draws a micro-facet and decides refraction/reflection kind; pass type SECOND applies the reflection off the
// chosen normal (spike: mechanical normal, back: -dir, Lambertian: sampled) and ends the crossing.
RT_HD V3 lambert_dir(Rng& g, V3 sn, V3 sx, V3 sy) { // UnifiedReflectorModel.m:309-316
    double ct = sqrt(rng_uniform(g)), st = sqrt(1 - ct * ct), ph = TWO_PI * rng_uniform(g), sp, cp;
    rt_sincos(ph, &sp, &cp);
    return ct * sn + (st * cp) * sx + (st * sp) * sy;
}
RT_HD V3 facet_normal(Rng& g, V3 indir, V3 sn, V3 sx, V3 sy, double sig, int maxiter) { // .m:278-306
    V3 f = sn;
    if (sig == 0) return f;
    for (int it = 0; it < maxiter; it++) {
        double th = fabs(sig * rng_normal(g)), st, ct;
        rt_sincos(th, &st, &ct);
        bool out_of_range = th >= 0.5 * 3.14159265358979323846;
        double fmx = sig * 4 < 1 ? sig * 4 : 1;
        bool fail_jac = rng_uniform(g) * fmx > st;
        double ph = TWO_PI * rng_uniform(g), sp, cp;
        rt_sincos(ph, &sp, &cp);
        f = ct * sn + (st * cp) * sx + (st * sp) * sy;
        bool wrong = dot(indir, f) >= 0;
        if (!(out_of_range || fail_jac || wrong)) break;
    }
    return f;
}
RT_HD void renorm(Ray10& t, double s0) { double f = rt_fdiv(s0, t.s0); t.s0 *= f; t.s1 *= f; t.s2 *= f; t.s3 *= f; }

// Out of line on the device: the rough-surface loop needs ~170 VGPRs on its own, which must not set the register
// budget (occupancy) of the whole bounce kernel.  The generator travels by value; the caller re-aligns its stream to
// the next Philox block afterwards (rng_align), so nothing has to come back but the ray.
struct UnifiedOut { Ray10 r; uint32_t idx; };
RT_HD UnifiedOut unified_body(Rng g, Ray10 in, V3 nrm, double n1, double n2, double sig, double pdiff, double c1, double c2, double c3, int maxiter) {
    Ray10 R = in;
    R.d = normalized_if(R.d);
    V3 N = normalized_if(nrm);
    R.pol = R.pol - dot(R.pol, R.d) * R.d;
    R.pol = normalized_if(R.pol);
    V3 md = -R.d;
    bool scattering = dot(md, N) > 0;
    V3 sx = cross(md, N);
    const double sx2 = dot(sx, sx), sxr = sx2 > 0 ? rt_rsqrt(sx2) : 0.0, sin_i = sx2 * sxr;
    if (sin_i > 0) sx = sxr * sx;
    else {
        V3 t = cross(V3{ 1, 0, 0 }, N);
        if (!(dot(t, t) > 0)) t = cross(V3{ 0, 1, 0 }, N);
        sx = rt_rsqrt(dot(t, t)) * t;
    }
    V3 sy = cross(N, sx);
    bool insurface = false;
    int guard = 0;
    while (scattering && guard++ < maxiter) {
        bool crossing = true, second = false;
        V3 nuse = N;
        int guard2 = 0;
        while (crossing && (second || guard2++ < maxiter)) { // the cap counts passes of the reference's loop: the Fresnel step of a chosen normal (`second`) belongs to the pass that chose it
            if (!second) nuse = facet_normal(g, R.d, N, sx, sy, sig, maxiter);
            FresnelCore fc; fresnel_core(R, nuse, n1, n2, fc);
            const double s0_in = R.s0;
            if (second) { fresnel_reflected(fc, R); renorm(R, s0_in); crossing = false; second = false; continue; }
            double t_s0 = fresnel_refr_s0(fc), r_s0 = fresnel_refl_s0(fc);
            bool reflect_here = rng_uniform(g) > (t_s0 / (t_s0 + r_s0));
            if (!reflect_here) {
                fresnel_refracted(fc, -1.0, R); renorm(R, s0_in);
                N = -N;
                double t = n1; n1 = n2; n2 = t;
                insurface = !insurface;
                crossing = dot(R.d, N) <= 0;
            } else {
                double roll = rng_uniform(g);
                if (roll < c1) { fresnel_reflected(fc, R); renorm(R, s0_in); crossing = dot(R.d, N) <= 0; }
                else {
                    if (roll < c2) nuse = N;
                    else if (roll < c3) nuse = -R.d;
                    else { V3 od = lambert_dir(g, N, sx, sy); V3 t = od - R.d; nuse = rt_rsqrt(dot(t, t)) * t; }
                    second = true;
                }
            }
        }
        scattering = scattering && insurface;
        if (scattering) { // .m:259-267: Lambertian re-emission from inside the surface material
            R.d = lambert_dir(g, -N, sx, sy);
            R.s1 = R.s2 = R.s3 = 0; R.s0 *= pdiff;
            V3 t = cross(V3{ 1, 0, 0 }, R.d);
            if (dot(t, t) <= 0) t = cross(V3{ 0, 1, 0 }, R.d);
            R.pol = rt_rsqrt(dot(t, t)) * t;
        }
    }
    return UnifiedOut{ R, g.idx };
}
RT_COLD UnifiedOut unified(Rng g, Ray10 in, V3 nrm, double n1, double n2, double sig, double pdiff, double c1, double c2, double c3, int maxiter) {
    return unified_body(g, in, nrm, n1, n2, sig, pdiff, c1, c2, c3, maxiter);
}
RT_HD void rng_align(Rng& g, uint32_t idx) { g.idx = (idx + 1u) & ~1u; } // continue with a fresh Philox block

#if defined(__HIP_DEVICE_COMPILE__)
// ---- the same model for a whole wave (k_level_phys PASS 2: every lane holds a ray that ended on a UNIFIED surface)
// The rejection loop of the facet normal (UnifiedReflectorModel.m:278-306) accepts about one draw in five at small
// sigma_alpha, and a wave waits for its unluckiest ray.  The generator is counter-based, so try number j of a ray can
// be evaluated before tries 0..j-1 are known to fail: the rays still drawing are spread over the 64 lanes, 64 / (rays
// left) tries each (rounded down), and every ray keeps the first of its tries that passes - the same normal, the same number of
// uniforms consumed as the ray-by-ray loop (the oracle's).  All 64 lanes must call (EXEC full); `need` marks the rays.
__device__ __forceinline__ double rng_uniform_at(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t step, uint32_t k0, uint32_t k1, uint32_t i) {
    uint32_t o0, o1, o2, o3;
    philox(c0, c1, c2, (step << 20) | ((i >> 1) & 0xFFFFFu), k0, k1, o0, o1, o2, o3);
    const uint32_t a = (i & 1u) ? o2 : o0, b = (i & 1u) ? o3 : o1;
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ void rng_seek(Rng& g, uint32_t idx) { // continue the stream at uniform number idx
    g.idx = idx;
    if (idx & 1u) philox(g.c0, g.c1, g.c2, (g.step << 20) | ((idx >> 1) & 0xFFFFFu), g.k0, g.k1, g.o0, g.o1, g.o2, g.o3);
}
__device__ __forceinline__ double shfl_d(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl((int)b, src), hi = __shfl((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ V3 shfl_v3(V3 v, int src) { return V3{ shfl_d(v.x, src), shfl_d(v.y, src), shfl_d(v.z, src) }; }

__device__ __forceinline__ V3 facet_normal_wave(bool need, Rng& g, V3 indir, V3 sn, V3 sx, V3 sy, double sig, int maxiter) {
    V3 f = sn;
    const int lane = (int)(threadIdx.x & 63);
    unsigned long long want = __ballot(need && sig != 0 && maxiter > 0); // rays that still need a normal
    int it = 0;                      // tries of my ray that are behind it
    const uint32_t idx0 = g.idx;
    const double fmx_own = sig * 4 < 1 ? sig * 4 : 1;
    while (want) {
        // Stage 1, in rounds: the polar angle of a try and the two tests that need nothing else (range, Jacobian), tried
        // speculatively for the rays that have no such try yet.  Stage 2, once for all of them: azimuth, the normal
        // itself and the facing test, each ray for its own first try that passed stage 1.
        unsigned long long pending = want;
        double t_u1 = 0, t_u2 = 0, t_u3 = 0; bool t_last = false;
        while (pending) {
            const int np = __popcll(pending);
            const int c = 64 / np;                                          // tries per ray this round
            const int rank = __popcll(pending & ((1ull << lane) - 1ull));  // my ray's place among the rays still drawing
            const bool mine = (pending >> lane) & 1ull;
            // rank -> lane: every drawing lane sends its number to the lane of its rank
            const int r2l = __builtin_amdgcn_ds_permute((mine ? rank : 63) << 2, mine ? lane : 0); // idle lanes all write to lane 63, which no rank < np <= 63 reads unless np == 64 (then nobody is idle)
            const int s = lane / c, j = lane % c;
            const bool work = s < np;
            const int src = __shfl(r2l, work ? s : 0);
            // the ray whose try this lane evaluates
            const float rs = (float)shfl_d(sig, src); const double fmx = shfl_d(fmx_own, src);
            const uint32_t r0 = (uint32_t)__shfl((int)g.c0, src), r1 = (uint32_t)__shfl((int)g.c1, src), r2 = (uint32_t)__shfl((int)g.c2, src), rstep = (uint32_t)__shfl((int)g.step, src);
            const uint32_t ridx = (uint32_t)__shfl((int)idx0, src);
            const int rit = __shfl(it, src);
            const int tr = rit + j;                                        // number of the try
            const uint32_t b = ridx + 4u * (uint32_t)tr;
            // uniforms b .. b+2 of that ray's stream: two Philox blocks (the fourth, the azimuth, is drawn below for the one try that needs it)
            uint32_t p0, p1, p2, p3, q0, q1, q2, q3;
            const uint32_t blk = b >> 1, odd = b & 1u;
            philox(r0, r1, r2, (rstep << 20) | (blk & 0xFFFFFu), g.k0, g.k1, p0, p1, p2, p3);
            philox(r0, r1, r2, (rstep << 20) | ((blk + 1u) & 0xFFFFFu), g.k0, g.k1, q0, q1, q2, q3);
            auto u53 = [](uint32_t a, uint32_t c_) { return ((double)(a >> 5) * 67108864.0 + (double)(c_ >> 6)) * (1.0 / 9007199254740992.0); };
            const double u1 = odd ? u53(p2, p3) : u53(p0, p1), u2 = odd ? u53(q0, q1) : u53(p2, p3);
            const double u3 = odd ? u53(q2, q3) : u53(q0, q1);
            // Single-precision screening: the polar angle to within dl (hardware log2 / cos / sin, errors of 1e-6 against a
            // margin of 4e-3 sigma), and a try is dropped here only if it fails the range test or the Jacobian test for every
            // angle in that interval; whatever survives is decided in double precision below, exactly as the loop does.
            const float z32 = __builtin_sqrtf(-2.0f * 0.69314718f * __builtin_amdgcn_logf((float)(1.0 - u1))) * __builtin_amdgcn_cosf((float)u2);
            const float th32 = __builtin_fabsf(rs * z32), dl = 4e-3f * __builtin_fabsf(rs) + 1e-6f, hi = th32 + dl;
            const bool drop_range = th32 - dl > 1.5708f;
            const bool drop_jac = hi < 1.57f && u3 * fmx > (double)(__builtin_amdgcn_sinf(hi * 0.15915494f) * 1.001f + 4e-6f);
            const bool last = tr == maxiter - 1;                           // the loop ends with this try whatever it gives
            const bool ok = work && tr < maxiter && (last || !(drop_range || drop_jac));
            const unsigned long long okm = __ballot(ok);
            // back at the rays: the first try of my group that was not dropped, if any
            const unsigned long long grp = (c == 64 ? okm : (okm >> (rank * c)) & ((1ull << c) - 1ull));
            const bool hit = mine && grp != 0;
            const int fj = hit ? __ffsll((long long)grp) - 1 : 0;
            const int from = mine ? rank * c + fj : lane;
            const double g_u1 = shfl_d(u1, from), g_u2 = shfl_d(u2, from), g_u3 = shfl_d(u3, from);
            const int g_last = __shfl((int)last, from);
            if (hit) { t_u1 = g_u1; t_u2 = g_u2; t_u3 = g_u3; t_last = g_last != 0; it += fj; }
            else if (mine) it += c;
            pending &= ~__ballot(hit);
        }
        // every ray still drawing has a try that the screening let through: the tests themselves, in double precision
        double z_s, z_c; rt_sincos(TWO_PI * t_u2, &z_s, &z_c);
        const double th = fabs(sig * (sqrt(-2.0 * rt_log(1.0 - t_u1)) * z_c));
        double t_st, t_ct; rt_sincos(th, &t_st, &t_ct);
        const bool pass1 = t_last || !(th >= 0.5 * 3.14159265358979323846 || t_u3 * fmx_own > t_st);
        const bool mine2 = (want >> lane) & 1ull;
        const double t_u4 = rng_uniform_at(g.c0, g.c1, g.c2, g.step, g.k0, g.k1, idx0 + 4u * (uint32_t)it + 3u);
        double sp, cp; rt_sincos(TWO_PI * t_u4, &sp, &cp);
        const V3 cand = t_ct * sn + (t_st * cp) * sx + (t_st * sp) * sy;
        const bool wrong = dot(indir, cand) >= 0;
        if (mine2) { it += 1; if (pass1) f = cand; }                     // that try is behind the ray now, whatever it gave
        want &= ~__ballot(mine2 && pass1 && (t_last || !wrong));          // failed after all, or faces the wrong way: on with the next tries
    }
    if (it > 0) rng_seek(g, idx0 + 4u * (uint32_t)it);
    return f;
}

// unified_body() as a state machine in which every pass of the wave does at most one pass of the reference's inner
// loop per ray, so that the facet normals of all rays that need one are drawn together.
__device__ __forceinline__ UnifiedOut unified_wave(bool need, Rng g, Ray10 in, V3 nrm, double n1, double n2, double sig, double pdiff, double c1, double c2, double c3, int maxiter) {
    Ray10 R = in;
    R.d = normalized_if(R.d);
    V3 N = normalized_if(nrm);
    R.pol = R.pol - dot(R.pol, R.d) * R.d;
    R.pol = normalized_if(R.pol);
    V3 md = -R.d;
    bool scattering = need && dot(md, N) > 0;
    V3 sx = cross(md, N);
    const double sx2 = dot(sx, sx), sxr = sx2 > 0 ? rt_rsqrt(sx2) : 0.0, sin_i = sx2 * sxr;
    if (sin_i > 0) sx = sxr * sx;
    else {
        V3 t = cross(V3{ 1, 0, 0 }, N);
        if (!(dot(t, t) > 0)) t = cross(V3{ 0, 1, 0 }, N);
        sx = rt_rsqrt(dot(t, t)) * t;
    }
    V3 sy = cross(N, sx);
    bool insurface = false, crossing = false, second = false;
    int guard = 0, guard2 = 0;
    V3 nuse = N;
    bool inner = false;              // inside the reference's `while crossing` loop
    for (;;) {
        // per ray: walk the loop heads until the next pass of the inner loop (or the end)
        bool run = false;
        if (need) {
            for (int hop = 0; hop < 3; hop++) {
                if (inner) {
                    if (crossing && (second || guard2++ < maxiter)) { run = true; break; }
                    inner = false;
                    scattering = scattering && insurface;
                    if (scattering) { // .m:259-267: Lambertian re-emission from inside the surface material
                        R.d = lambert_dir(g, -N, sx, sy);
                        R.s1 = R.s2 = R.s3 = 0; R.s0 *= pdiff;
                        V3 t = cross(V3{ 1, 0, 0 }, R.d);
                        if (dot(t, t) <= 0) t = cross(V3{ 0, 1, 0 }, R.d);
                        R.pol = rt_rsqrt(dot(t, t)) * t;
                    }
                }
                if (!(scattering && guard++ < maxiter)) { need = false; break; }
                inner = true; crossing = true; second = false; nuse = N; guard2 = 0;
            }
        }
        if (!__ballot(run)) break;
        const V3 fn = facet_normal_wave(run && !second, g, R.d, N, sx, sy, sig, maxiter);
        if (run) {
            if (!second) nuse = fn;
            FresnelCore fc; fresnel_core(R, nuse, n1, n2, fc);
            const double s0_in = R.s0;
            if (second) { fresnel_reflected(fc, R); renorm(R, s0_in); crossing = false; second = false; }
            else {
                double t_s0 = fresnel_refr_s0(fc), r_s0 = fresnel_refl_s0(fc);
                bool reflect_here = rng_uniform(g) > (t_s0 / (t_s0 + r_s0));
                if (!reflect_here) {
                    fresnel_refracted(fc, -1.0, R); renorm(R, s0_in);
                    N = -N;
                    double t = n1; n1 = n2; n2 = t;
                    insurface = !insurface;
                    crossing = dot(R.d, N) <= 0;
                } else {
                    double roll = rng_uniform(g);
                    if (roll < c1) { fresnel_reflected(fc, R); renorm(R, s0_in); crossing = dot(R.d, N) <= 0; }
                    else {
                        if (roll < c2) nuse = N;
                        else if (roll < c3) nuse = -R.d;
                        else { V3 od = lambert_dir(g, N, sx, sy); V3 t = od - R.d; nuse = rt_rsqrt(dot(t, t)) * t; }
                        second = true;
                    }
                }
            }
        }
    }
    return UnifiedOut{ R, g.idx };
}
#endif


// ---------------------------------------------------------------------------------------------- Rayleigh
// MATLAB/Physics/RayleighScattering.m:62-148; lut = rayleigh_azimuth (:55-60), prec = azimuth_precision
RT_HD Ray10 rayleigh(Rng& g, const Ray10& in, const double* lut, int prec) {
    Ray10 r = in;
    r.d = normalized_if(r.d);
    r.pol = r.pol - dot(r.pol, r.d) * r.d;
    r.pol = normalized_if(r.pol);
    double u1 = rng_uniform(g), u2 = rng_uniform(g), u3 = rng_uniform(g);
    double linpol = sqrt(r.s1 * r.s1 + r.s2 * r.s2), frac = linpol / r.s0;
    double phi = TWO_PI * u2;
    if (frac > u1) {
        // The angle of the polarisation axis, atan2(s2, s1) / 2 up to a multiple of pi (an axis: same state, same
        // distribution), taken with the cut of atan2 at s1 = 0, s2 > 0 instead of s2 = 0, s1 < 0 - where light sits after
        // every s-polarising Fresnel step with an s2 that is zero up to noise, whose sign would pick a scattered direction
        // or its mirror image.  The oracle does the same (oracle/rt_oracle.c, where the noise is explained).
        double ang = 0.5 * atan2_atom(r.s1, -r.s2) - 0.25 * 3.14159265358979323846, pu = (double)prec * u2;
        int ix = (int)floor(pu) + 1;
        if (ix > prec) ix = prec;
        double ixf = pu - (double)(ix - 1), a0 = lut[ix - 1], a1 = lut[ix];
        phi = ang + a0 + ixf * (a1 - a0);
    }
    double c, s;
    rt_sincos(phi, &s, &c);
    double c2 = c * c - s * s, sn2 = 2 * c * s, o1 = r.s1, o2 = r.s2;
    r.s1 = o1 * c2 + o2 * sn2; r.s2 = -o1 * sn2 + o2 * c2;
    V3 oref = r.pol, operp = cross(r.d, r.pol);
    r.pol = c * oref + s * operp;
    V3 nperp = cross(r.d, r.pol);
    // cos(theta) = (cbrt(z + sq) + cbrt(z - sq)) / cbrt(1 + a) with sq = sqrt(z^2 + w): for a -> 1 (w -> 0) one of the two
    // arguments is a difference of equal numbers; it is taken from the other through (z + sq)(z - sq) = -w, as in the oracle
    // (oracle/rt_oracle.c has the story)
    double a = r.s1 / r.s0, w = (1 - a) * (1 - a) * (1 - a) / (1 + a), z = (2 - a) * (1 - 2 * u3), sq = sqrt(z * z + w), zp, zm;
    if (z >= 0) { zp = z + sq; zm = zp != 0 ? -w / zp : 0.0; } else { zm = z - sq; zp = -w / zm; }
    double A = (double)isgn(zp) * rt_cbrt(fabs(zp)), B = (double)isgn(zm) * rt_cbrt(fabs(zm));
    double ct = (A + B) / rt_cbrt(1 + a), st = sqrt(1 - ct * ct);
    Ray10 o = r;
    o.d = ct * r.d + (st * c) * oref + (st * s) * operp;
    o.pol = cross(nperp, o.d);
    double c2t = ct * ct, R11 = c2t + 1, R12 = c2t - 1, rn = r.s0 / (r.s0 * R11 + r.s1 * R12);
    o.s1 = (r.s0 * R12 + r.s1 * R11) * rn;
    double R33 = 2 * ct * rn;
    o.s2 = r.s2 * R33; o.s3 = r.s3 * R33;
    return o;
}

// ---------------------------------------------------------------------------------------------- one step of one ray
struct RayState {
    V3 p; Ray10 r;
    int64_t ray_id, parent;
    uint32_t hash; int32_t k, smix_last, six_last, flags;
};

struct Bounce {
    int scat, surf_sc, ray_sc;      // scatter_cut, surface_scatter_cut, rayleigh_scatter_cut (RayTracer2.py:404-414)
    int smix_next, six_next, surf0, orient;
    double l; V3 p, nmech;
    Ray10 inc, refr, refl;
    double n1, n2, absl1, absl2, rayl1, rayl2;
    double bulk_abs, surf_abs, inc_intensity;
    int fa, fb;                     // follow refracted / reflected (RayTracer2.py:518-519)
};

// python/RayTracer2.py:275-519 (MATLAB/RayTracer2.m:301-553) for one ray of the live list.
#if defined(RT_OOL_FIND_HIT) && defined(__HIP_DEVICE_COMPILE__)
// The kernel keeps the nearest-hit search out of line (defined in rt_kernels.hip): its registers are then allocated
// on their own instead of competing with the ray state and the physics temporaries of the caller.
__device__ __noinline__ Hit find_hit_ool(V3 sp, V3 d, int six_last, bool act);
#endif
// The whole wave calls this together (see find_hit); `act` = this lane carries a live ray.
RT_HD void bounce_hit(const Geom& G, const Prm& P, RayState& q, bool act, Hit& h) {
    q.r.d = normalized_if(q.r.d); // RayToShape.py:11-14
#if defined(RT_OOL_FIND_HIT) && defined(__HIP_DEVICE_COMPILE__)
    h = find_hit_ool(q.p, q.r.d, q.six_last, act);
#else
    find_hit(G, P, q.p, q.r.d, q.six_last, act, h);
#endif
}
// CF >= 0: the flags are known at compile time (the kernel is instantiated per flag set, so that fields of Bounce that
// only feed records/final rows are dead code in the tally-only variants); CF < 0: read P.flags.
// UNI = 2: no ray that comes here ends on a UNIFIED surface (the caller checked); the rough-surface model and the
// registers its call costs are then not compiled in.  UNI = 1: the model in line (a launch of UNIFIED rays only).
template <int CF = -1, int UNI = 0> RT_HD void bounce_phys(const Geom& G, const Prm& P, RayState& q, const Hit& h, Bounce& b) {
    const int flags = CF >= 0 ? CF : P.flags;
    Rng g; rng_init(g, P.key0, P.key1, q.ray_id, q.hash, q.k);
    b.inc_intensity = q.r.s0;
    b.scat = 0; b.surf_sc = 0; b.ray_sc = 0; b.fa = 0; b.fb = 0; b.smix_next = 0; b.six_next = 0; b.surf0 = h.surf; b.orient = h.orient;
    b.l = h.l;
    if (h.surf < 0) return; // no surface ahead: rayleigh_next stays inf, so scatter_cut is false (RayTracer2.py:404)
    const double* row = G.surf + h.surf * ROW;
    int st = (int)row[1];
    bool in = h.orient < 0; // RayTracer2.py:351-358
    b.n1 = in ? row[16] : row[15]; b.n2 = in ? row[15] : row[16];
    b.absl1 = in ? row[19] : row[18]; b.absl2 = in ? row[18] : row[19];
    b.rayl1 = in ? row[21] : row[20]; b.rayl2 = in ? row[20] : row[21];
    double absn = row[17];
    V3 pn = q.p + h.l * q.r.d;
    b.nmech = surface_normal(row, pn, q.r.d, h.orient);
    V3 sopt = sel(st == ST_RETRO, -q.r.d, b.nmech); // RayTracer2.m:332-333
    b.smix_next = b.six_next = (h.surf + P.base) * h.orient; // RayTracer2.py:371
    if (st == ST_DIFFUSE) { // RayTracer2.m:394-415
        double ct = sqrt(rng_uniform(g)), stt = sqrt(1 - ct * ct), ph = rng_uniform(g) * TWO_PI, sph, cph;
        rt_sincos(ph, &sph, &cph);
        V3 xt = cross(sopt, V3{ 1, 0, 0 }), yt = cross(sopt, V3{ 0, 1, 0 });
        if (xt.x == 0 && xt.y == 0 && xt.z == 0) xt = yt;
        xt = rt_rsqrt(dot(xt, xt)) * xt;
        yt = cross(sopt, xt);
        V3 od = ct * sopt + (stt * cph) * xt + (stt * sph) * yt - q.r.d;
        sopt = rt_rsqrt(dot(od, od)) * od;
    }
    b.scat = 1; // l_next < inf
    double l_bulk = INFINITY;
    if (b.rayl1 < INFINITY) l_bulk = -b.rayl1 * rt_log(rng_uniform(g)); // RayTracer2.py:409
    b.surf_sc = h.l <= l_bulk; b.ray_sc = !b.surf_sc;
    bool uni_sc = UNI != 2 && b.surf_sc && st == ST_UNIFIED, nor_sc = b.surf_sc && !uni_sc;
    if (b.ray_sc) { b.six_next = 0; b.l = l_bulk; pn = q.p + l_bulk * q.r.d; } // :417-421
    b.p = pn;
    double trans = rt_exp(-b.l / b.absl1); // :424-427
    b.bulk_abs = b.inc_intensity * (1 - trans);
    b.inc = q.r; b.inc.s0 *= trans; b.inc.s1 *= trans; b.inc.s2 *= trans; b.inc.s3 *= trans;
    b.refr = b.inc; b.refr.s0 = b.refr.s1 = b.refr.s2 = b.refr.s3 = 0; b.refl = b.refr;
    if (nor_sc) fresnel(b.inc, sopt, b.n1, b.n2, P.tir, b.refr, b.refl); // :438-439
    if constexpr (UNI != 2) {
        if (uni_sc) { // :443-444
            double c1 = row[24], c2 = c1 + row[25], c3 = c2 + row[26];
            UnifiedOut u;
            if constexpr (UNI == 1) u = unified_body(g, b.inc, b.nmech, b.n1, b.n2, row[22], row[23], c1, c2, c3, P.maxiter);
            else u = unified(g, b.inc, b.nmech, b.n1, b.n2, row[22], row[23], c1, c2, c3, P.maxiter);
            b.refl = u.r; rng_align(g, u.idx);
        }
    }
    if (b.surf_sc) { double f = 1 - absn; b.refr.s0 *= f; b.refr.s1 *= f; b.refr.s2 *= f; b.refr.s3 *= f; b.refl.s0 *= f; b.refl.s1 *= f; b.refl.s2 *= f; b.refl.s3 *= f; } // :447-449
    if (b.ray_sc) b.refl = rayleigh(g, b.inc, G.lut, G.lut_prec); // :452-454
    if (flags & FLAG_SINGLE) { // :458-467
        double tot = b.refl.s0 + b.refr.s0;
        bool roll = rng_uniform(g) < (b.refl.s0 / tot);
        if (roll) b.refr = b.refl;
        double resc = tot / b.refr.s0;
        if (isnan(resc)) resc = 0;
        if (isnan(tot)) tot = 0;
        b.refr.s0 = tot; b.refr.s1 *= resc; b.refr.s2 *= resc; b.refr.s3 *= resc;
        b.refl.s0 = b.refl.s1 = b.refl.s2 = b.refl.s3 = 0;
    }
    b.surf_abs = b.inc.s0 - b.refr.s0 - b.refl.s0; // :471
    b.fa = b.refr.s0 > P.thrA; b.fb = b.refl.s0 > P.thrB; // :518-519
}

// Does the ray scatter in the bulk (Rayleigh) before it reaches the surface of its nearest hit?  The draw and the
// comparison of python/RayTracer2.py:409-417 exactly as bounce_phys_single() / bounce_phys() make them (same stream
// position: a diffuse surface has drawn two uniforms before).  Used by the level path to sort rays into launches.
RT_HD bool level_ray_scatters(const Geom& G, const Prm& P, int64_t ray_id, uint32_t hash, int k, const Hit& h) {
    if (h.surf < 0) return false;
    const double* row = G.surf + h.surf * ROW;
    const bool in = h.orient < 0;
    const double rayl1 = in ? row[21] : row[20];
    if (!(rayl1 < INFINITY)) return false;
    Rng g; rng_init(g, P.key0, P.key1, ray_id, hash, k);
    if ((int)row[1] == ST_DIFFUSE) { (void)rng_uniform(g); (void)rng_uniform(g); }
    const double l_bulk = -rayl1 * rt_log(rng_uniform(g));
    return !(h.l <= l_bulk);
}

// ---------------------------------------------------------------------------------------------- single-child step, tallies only
// The same step as bounce_phys<FLAG_SINGLE>() for runs that keep neither records nor final rows: the dice of
// python/RayTracer2.py:458-467 needs only the two children's intensities, so they are computed first and only the
// surviving child is built.  Fewer live values (one child instead of three Ray10) is what lets the level kernel
// run without register spills.
struct LeanBounce {
    int scat, surf_sc, smix_next, six_next, fa;
    double bulk_abs, surf_abs, inc_intensity, s0_child; // s0_child: refracted_rays[:,6] after the dice
    V3 p; Ray10 child;
};
// UNI = 1: the kernel pass that sees nothing but rays ending on UNIFIED surfaces carries the rough-surface loop in
// line, with the register budget that needs; UNI = 2: the pass that has put those rays aside carries no UNIFIED code at
// all; UNI = 0 (everywhere else): out of line.
// NORAY (with UNI = 2): 1 = the caller has also put aside the rays that scatter in the bulk before they reach their surface
// (level_ray_scatters() below, the same arithmetic): every ray that comes here scatters on its surface, no Rayleigh code.
// 2 = no Rayleigh code either, but the draw of python/RayTracer2.py:409 is made here: a ray that scatters in the bulk
// comes back untouched with b.scat = 2, for the caller to hand to the launch that has that code (which starts the
// ray's stream afresh and makes the same draw).
template <int UNI = 0, int NORAY = 0> RT_HD void bounce_phys_single(const Geom& G, const Prm& P, const RayState& q, const Hit& h, LeanBounce& b, bool act = true) {
    Rng g; rng_init(g, P.key0, P.key1, q.ray_id, q.hash, q.k);
    b.inc_intensity = q.r.s0;
    b.scat = 0; b.surf_sc = 0; b.fa = 0; b.smix_next = 0; b.six_next = 0; b.bulk_abs = 0; b.surf_abs = 0; b.s0_child = 0;
    // RayTracer2.py:404: no surface ahead, nothing happens.  UNI = 1 keeps such a lane (and one without a ray, !act) in step with
    // its wave - unified_wave() below needs all 64 - on the first surface's row, and takes its result back at the end.
    const bool nohit = h.surf < 0;
    if constexpr (UNI != 1) { if (nohit) return; }
    const double* row = G.surf + (nohit ? 0 : h.surf) * ROW;
    const int st = (int)row[1];
    const bool in = h.orient < 0; // RayTracer2.py:351-358
    const double n1 = in ? row[16] : row[15], n2 = in ? row[15] : row[16], absl1 = in ? row[19] : row[18], rayl1 = in ? row[21] : row[20];
    V3 pn = q.p + h.l * q.r.d;
    const V3 nmech = surface_normal(row, pn, q.r.d, h.orient);
    V3 sopt = sel(st == ST_RETRO, -q.r.d, nmech); // RayTracer2.m:332-333
    b.smix_next = b.six_next = (h.surf + P.base) * h.orient; // RayTracer2.py:371
    if (st == ST_DIFFUSE) { // RayTracer2.m:394-415
        double ct = sqrt(rng_uniform(g)), stt = sqrt(1 - ct * ct), ph = rng_uniform(g) * TWO_PI, sph, cph;
        rt_sincos(ph, &sph, &cph);
        V3 xt = cross(sopt, V3{ 1, 0, 0 }), yt = cross(sopt, V3{ 0, 1, 0 });
        if (xt.x == 0 && xt.y == 0 && xt.z == 0) xt = yt;
        xt = rt_rsqrt(dot(xt, xt)) * xt;
        yt = cross(sopt, xt);
        V3 od = ct * sopt + (stt * cph) * xt + (stt * sph) * yt - q.r.d;
        sopt = rt_rsqrt(dot(od, od)) * od;
    }
    b.scat = 1;
    double l_bulk = INFINITY;
    if (rayl1 < INFINITY) l_bulk = -rayl1 * rt_log(rng_uniform(g)); // RayTracer2.py:409
    const bool surf_sc = (UNI == 2 && NORAY == 1) ? true : (h.l <= l_bulk);
    b.surf_sc = surf_sc;
    if constexpr (UNI == 2 && NORAY == 2) { if (!surf_sc) { b.scat = 2; return; } }
    const bool uni_sc = UNI != 2 && surf_sc && st == ST_UNIFIED, nor_sc = surf_sc && !uni_sc;
    double l = h.l;
    if (!surf_sc) { b.six_next = 0; l = l_bulk; pn = q.p + l_bulk * q.r.d; } // :417-421
    b.p = pn;
    const double trans = rt_exp(-l / absl1); // :424-427
    b.bulk_abs = b.inc_intensity * (1 - trans);
    Ray10 inc = q.r; inc.s0 *= trans; inc.s1 *= trans; inc.s2 *= trans; inc.s3 *= trans;
    const double f = surf_sc ? 1 - row[17] : 1.0; // :447-449
    // the child that survives the dice, before the rescaling: chosen; its intensity and the sum of both
    Ray10 ch = inc; ch.s0 = ch.s1 = ch.s2 = ch.s3 = 0;
    double tot = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    UnifiedOut uw;
    if constexpr (UNI == 1) { // all 64 lanes are here (see k_level_phys PASS 2); those with uni_sc are the rays
        const double c1 = row[24], c2 = c1 + row[25], c3 = c2 + row[26];
        uw = unified_wave(uni_sc && act && !nohit, g, inc, nmech, n1, n2, row[22], row[23], c1, c2, c3, P.maxiter);
    }
#endif
    if (nor_sc) { // :438-439
        FresnelCore c; fresnel_core(inc, sopt, n1, n2, c);
        V3 ny; const bool goodcut = fresnel_yaxis(c, ny);
        const double hs = 0.5 * (c.s0 + c.s1), hp = 0.5 * (c.s0 - c.s1);
        const bool tirc = goodcut && c.sin_t >= 1 && P.tir >= 0; // RefractionReflectionAtInterface.m:221-225
        double r_s0 = c.goodhit ? hs * c.Rs + hp * c.Rp : 0.0, t_s0 = c.goodhit ? hs * c.Ts + hp * c.Tp : c.s0;
        if (tirc) t_s0 = r_s0 * P.tir;
        r_s0 *= f; t_s0 *= f;
        tot = r_s0 + t_s0;
        const bool roll = rng_uniform(g) < rt_fdiv(r_s0, tot); // :458-462 (0 / 0: NaN, never smaller)
        // reflected child, or the refracted one where it is a copy of the reflected (tir_handling >= 0)
        const bool as_refl = roll || tirc;
        ch.pol = c.pol;
        if (as_refl) {
            ch.d = goodcut ? c.cos_i * c.n - c.sin_i * ny : -c.d;
            const double k = roll ? 1.0 : P.tir;
            if (c.goodhit) { ch.s0 = (hs * c.Rs + hp * c.Rp) * k; ch.s1 = (hs * c.Rs - hp * c.Rp) * k; ch.s2 = (c.s2 * c.cr - c.s3 * c.ci) * k; ch.s3 = (c.s2 * c.ci + c.s3 * c.cr) * k; }
        } else {
            ch.d = c.d; ch.s0 = c.s0; ch.s1 = c.s1; ch.s2 = c.s2; ch.s3 = c.s3;
            if (c.goodhit) { ch.s0 = hs * c.Ts + hp * c.Tp; ch.s1 = hs * c.Ts - hp * c.Tp; ch.s2 = c.s2 * c.tst; ch.s3 = c.s3 * c.tst; }
            if (goodcut) ch.d = (c.sin_t >= 1) ? -ny : (-c.cos_t) * c.n - c.sin_t * ny;
        }
        ch.s0 *= f; ch.s1 *= f; ch.s2 *= f; ch.s3 *= f;
    } else if constexpr (!(UNI == 2 && NORAY != 0)) {
        Ray10 r;
        bool ray_sc = true;
        if constexpr (UNI != 2) {
            if (uni_sc) { // :443-444
                const double c1 = row[24], c2 = c1 + row[25], c3 = c2 + row[26];
                UnifiedOut u;
                if constexpr (UNI == 1) {
#if defined(__HIP_DEVICE_COMPILE__)
                    u = uw;
#else
                    u = unified_body(g, inc, nmech, n1, n2, row[22], row[23], c1, c2, c3, P.maxiter);
#endif
                } else u = unified(g, inc, nmech, n1, n2, row[22], row[23], c1, c2, c3, P.maxiter);
                r = u.r; rng_align(g, u.idx);
                r.s0 *= f; r.s1 *= f; r.s2 *= f; r.s3 *= f;
                ray_sc = false;
            }
        }
        if (ray_sc) r = rayleigh(g, inc, G.lut, G.lut_prec); // :452-454
        tot = r.s0 + 0.0;
        const bool roll = rng_uniform(g) < (r.s0 / tot);
        if (roll) ch = r;
    }
    double resc = rt_fdiv(tot, ch.s0); // :463-466
    if (isnan(resc)) resc = 0;
    if (isnan(tot)) tot = 0;
    ch.s0 = tot; ch.s1 *= resc; ch.s2 *= resc; ch.s3 *= resc;
    b.child = ch; b.s0_child = tot;
    b.surf_abs = inc.s0 - tot; // :471
    b.fa = tot > P.thrA; // :518
    if constexpr (UNI == 1) { if (nohit) { b.scat = 0; b.surf_sc = 0; b.fa = 0; b.smix_next = 0; b.six_next = 0; b.bulk_abs = 0; b.surf_abs = 0; b.s0_child = 0; } }
}
RT_HD RayState lean_child(const RayState& q, const LeanBounce& b) { // child_of(q, b, false, true)
    RayState c;
    c.p = b.p; c.r = b.child;
    c.ray_id = q.ray_id; c.parent = -1; c.k = q.k + 1;
    c.hash = path_hash(q.hash, q.k, 0);
    c.smix_last = -b.smix_next;
    c.six_last = b.six_next < 0 ? -b.six_next : b.six_next;
    c.flags = q.flags;
    return c;
}

template <int CF = -1> RT_HD void bounce(const Geom& G, const Prm& P, RayState& q, Bounce& b) {
    Hit h; bounce_hit(G, P, q, true, h); bounce_phys<CF>(G, P, q, h, b);
}

// state of a child ray (python/RayTracer2.py:531-540)
RT_HD RayState child_of(const RayState& q, const Bounce& b, bool reflected, bool primary_goes_to_refr) {
    RayState c;
    c.p = b.p; c.r = sel(reflected, b.refl, b.refr);
    c.ray_id = q.ray_id; c.parent = -1; c.k = q.k + 1;
    c.hash = path_hash(q.hash, q.k, reflected ? 1 : 0);
    c.smix_last = reflected ? b.smix_next : -b.smix_next;
    c.six_last = b.six_next < 0 ? -b.six_next : b.six_next;
    int f = q.flags & RF_HAS_REFL;
    if (reflected) f |= RF_HAS_REFL | RF_BORN_REFL;
    bool prim = (q.flags & RF_PRIMARY) && (reflected ? !primary_goes_to_refr : true);
    if (prim) f |= RF_PRIMARY;
    c.flags = f;
    return c;
}

// record of one interaction, column order of include/rt_mi355x.h (RT_REC_COLS); element c goes to dst[c*stride]
RT_HD double ray_index_of(const RayState& q, const Prm& P) { double r = (double)(q.ray_id + P.base); return (q.flags & RF_HAS_REFL) ? -r : r; }
RT_HD void put_ray10(double* dst, size_t stride, int c, const Ray10& r) {
    dst[(c + 0) * stride] = r.d.x; dst[(c + 1) * stride] = r.d.y; dst[(c + 2) * stride] = r.d.z;
    dst[(c + 3) * stride] = r.pol.x; dst[(c + 4) * stride] = r.pol.y; dst[(c + 5) * stride] = r.pol.z;
    dst[(c + 6) * stride] = r.s0; dst[(c + 7) * stride] = r.s1; dst[(c + 8) * stride] = r.s2; dst[(c + 9) * stride] = r.s3;
}
RT_HD void fill_record(const RayState& q, const Bounce& b, const Prm& P, double* dst, size_t stride) {
    put_ray10(dst, stride, 0, b.inc); put_ray10(dst, stride, 10, b.refr); put_ray10(dst, stride, 20, b.refl);
    dst[30 * stride] = b.p.x; dst[31 * stride] = b.p.y; dst[32 * stride] = b.p.z;
    dst[33 * stride] = b.nmech.x; dst[34 * stride] = b.nmech.y; dst[35 * stride] = b.nmech.z;
    dst[36 * stride] = ray_index_of(q, P); dst[37 * stride] = (double)b.six_next; dst[38 * stride] = b.l;
    dst[39 * stride] = b.n1; dst[40 * stride] = b.n2; dst[41 * stride] = b.absl1; dst[42 * stride] = b.absl2;
    dst[43 * stride] = b.rayl1; dst[44 * stride] = b.rayl2;
    dst[45 * stride] = (double)b.orient; dst[46 * stride] = (double)b.surf0; dst[47 * stride] = (double)b.ray_sc;
}
// final row (RT_FINAL_COLS), rewritten at every interaction of the lineage that owns it
RT_HD void fill_final(const RayState& q, const Bounce& b, const Prm& P, double* dst) {
    dst[0] = b.p.x; dst[1] = b.p.y; dst[2] = b.p.z;
    put_ray10(dst, 1, 3, b.inc);
    dst[13] = b.nmech.x; dst[14] = b.nmech.y; dst[15] = b.nmech.z;
    dst[16] = (double)b.six_next; dst[17] = (double)b.orient; dst[18] = ray_index_of(q, P); dst[19] = (double)(q.k + 1);
    dst[20] = (b.fa ? b.refr.s0 : 0.0) + (b.fb ? b.refl.s0 : 0.0);
    dst[21] = (b.fa || b.fb) ? ((q.k + 1 >= P.K) ? 2.0 : 0.0) : 1.0;
    const Ray10 o = sel(b.fa != 0, b.refr, b.refl);
    dst[22] = o.d.x; dst[23] = o.d.y; dst[24] = o.d.z; dst[25] = o.pol.x; dst[26] = o.pol.y; dst[27] = o.pol.z;
    dst[28] = o.s1; dst[29] = o.s2; dst[30] = o.s3; dst[31] = 0.0;
}

} // namespace rt
