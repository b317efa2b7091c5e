// rt_kernels.hip -- HIP kernels (gfx950) and the C ABI of include/rt_mi355x.h.
//
// Execution model (DESIGN.md has the long version):
//   * k_trace is a persistent kernel.  Every lane owns one ray and keeps it in registers over all of its
//     bounces (python/RayTracer2.py:266 `while` loop, one iteration = one call of rt::bounce).  When lanes retire
//     (ray escaped / absorbed / below threshold / max_scatters) the wave refills them from the input queue with
//     one wave-aggregated atomic (ballot + popcount prefix); `refill_min` is the compaction cadence.
//   * Non-dice mode (ray splitting, RayTracer2.py:531-540): the lane keeps following one child and appends the
//     other to an output queue (ballot + prefix sum + one atomic per wave); the host launches one generation per
//     queue until the queue is empty.
//   * Surface table, bounds atoms/code and the absorption_table tallies live in LDS; tallies are flushed with
//     FP64 atomics once per workgroup.
//   * Records (full_output) are appended structure-of-arrays with the same ballot/prefix allocation and carry
//     (step, ray id, parent slot, born-reflected flag), from which the host rebuilds the reference's row order.
// No scalar-cache writes, no inter-workgroup waits: every wave's exit condition (queue drained and all its lanes
// retired, each ray bounded by max_scatters and the UNIFIED loop cap) is reached independently.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <atomic>
#include <thread>
#include <mutex>
#include <memory>
#include <algorithm>

#define RT_HD __host__ __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_ANY(x) (__ballot((x)) != 0ull)
#else
#define RT_ANY(x) (x)
#endif
#ifndef RT_INLINE_ALL
#define RT_OOL_FIND_HIT 1
#endif
#include "rt_device.hpp"
#include "../../include/rt_mi355x.h"

using namespace rt;

static_assert(RT_REC_COLS == REC && RT_FINAL_COLS == FIN && RT_ROW == ROW && RT_ATOM == ATOM, "header/device constants differ");

namespace {

constexpr int BLOCK = 256;
constexpr int QD = 13, QI = 2, QM = 5; // queue fields: doubles, int64 (ray_id, parent), int32 (hash,k,smix_last,six_last,flags)
constexpr int GEN_STEP = 4095;         // Philox step reserved for ray generation

struct Queue { double* d; int64_t* i; int32_t* m; int64_t cap; };

struct KArgs {
    // input
    int mode;               // 0: initial rays, SoA in_d[f*in_stride+i]; 1: queue of spawned rays; 2: point source, 3: camera rays generated in-kernel
    const double* in_d; const int64_t* in_i; const int32_t* in_m; int64_t in_stride; int64_t n_in;
    int64_t ray_id_base;    // id of ray 0 of this launch's input (modes 0, 2)
    int64_t id_stride;      // id of input ray i = ray_id_base + i * id_stride (1; rt_params.ray_id_stride: interleaved shards)
    int64_t slot_base;      // final-row index = ray_id - slot_base
    double ox, oy, oz;      // point source
    const double* cam; int64_t cam_base; // mode 3: camera block (see rt_trace_camera), ray id of pixel 0
    unsigned long long* head;
    // spawn queue
    Queue out; unsigned long long* tail;
    // records
    double* rec; int64_t rec_cap; unsigned long long* rec_count; int32_t* rec_step; int64_t* rec_ray; int64_t* rec_parent; int32_t* rec_flags;
    // final rows
    double* fin;
    // tallies
    double* table; unsigned long long* n_live; unsigned long long* n_scat;
    int* overflow;
    unsigned long long* prof; // RT_PROFILE builds: per-phase cycle sums (diagnostic only, never read by the kernel)
    // geometry
    const double* surf; int S; const double* recs; const uint32_t* smeta; const double* lut; int lut_prec;
    const double* prog; int nprog; // frame program (rt_device.hpp find_hit)
    Prm P;
    int rows_lds;           // absorption_table rows kept in LDS (the rest goes straight to global atomics)
    int refill_min;         // refill when at least this many lanes of the wave are idle (or all are)
    // level-synchronous path (k_level_hit / k_level_phys)
    int step;                               // the step k both kernels work on
    uint32_t lean_hash;                     // tally-only single-child runs: the path hash of every ray of that step
    const unsigned long long* cnt_in;       // number of live rays of this step (null: n_in)
    double* hit_l; int32_t* hit_so;         // nearest hit of live ray i: distance, surface * 4 + orientation + 1 (-1: none)
    // segments: block b of a level kernel owns queue slots [b * seg_cap, b * seg_cap + seg_in[b]) and writes the children
    // of its rays, compacted, into the same range of the output queue (count -> seg_out[b]).  No counter is shared
    // between blocks: one atomic on one address per wave and step was what the physics kernel spent most of its time on.
    const uint32_t* seg_in; uint32_t* seg_out; int64_t seg_cap;
    int64_t seg_fill;       // k_level_init: rays per segment at the start (== seg_cap unless rays split and segments need room to grow)
    // rays that end on a UNIFIED surface are put aside by the first physics pass (their offsets within the segment,
    // defer[b * seg_cap + j], j < defer_n[2 b]) and taken by a second launch in which every lane runs the rough-surface loop;
    // rays that scatter in the bulk go there too, listed from the other end (defer[(b + 1) * seg_cap - 1 - j], j < defer_n[2 b + 1])
    uint32_t* defer; uint32_t* defer_n;
    uint32_t* dense;        // pass 1: the other rays of the segment, listed the same way, so that its waves are full
    int lean;               // the level queues hold lean rows (tally-only single-child runs): set by the host, for writers and readers alike
};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// reference tally rule (python/RayTracer2.py:474-486, 521-528) into LDS rows / global rows
__device__ __forceinline__ void tally_add(const KArgs& a, double* lds_tab, int k, int cat, int smix, double v) {
    const int S = a.S, base = a.P.base;
    int idx, o, both = 0;
    if (smix == 0) { if (base != 0 || S <= 0) return; idx = 0; o = 0; both = 1; }
    else { idx = (smix > 0 ? smix : -smix) - base; if (idx < 0 || idx >= S) return; o = smix > 0 ? 0 : 1; }
    size_t e = ((size_t)(k * 5 + cat) * S + idx) * 2 + o;
    if (k < a.rows_lds) { atomicAdd(&lds_tab[e], v); if (both) atomicAdd(&lds_tab[e + 1], v); }
    else { atomicAdd(&a.table[e], v); if (both) atomicAdd(&a.table[e + 1], v); }
}

__device__ __forceinline__ void gen_point_ray(const KArgs& a, int64_t ray_id, V3& p, Ray10& r) {
    // MATLAB/Examples/RunExamples/RunSBCGeometry.m:45-68: isotropic direction, unpolarised, s0 = 1, polref = x^ x d
    Rng g; rng_init(g, a.P.key0, a.P.key1, ray_id, 0u, GEN_STEP);
    double ct = 1 - 2 * rng_uniform(g), st = sqrt(1 - ct * ct), ph = TWO_PI * rng_uniform(g), sp, cp;
    sincos(ph, &sp, &cp);
    p = V3{ a.ox, a.oy, a.oz };
    r.d = V3{ st * cp, st * sp, ct };
    V3 t = cross(V3{ 1, 0, 0 }, r.d);
    if (dot(t, t) == 0) t = cross(V3{ 0, 1, 0 }, r.d);
    r.pol = (1.0 / sqrt(dot(t, t))) * t;
    r.s0 = 1; r.s1 = r.s2 = r.s3 = 0;
}

// MATLAB/Examples/ImagingExamples/GenerateRaysFromCamera.m:42-81 for pixel `pix` (column-major over (i, j) as i_pix(:),
// j_pix(:) are, :44-46).  Camera block (doubles): [0..1] resolution, [2] lens type, [3] number of distortion terms,
// [4..5] pixel pitch, [6..7] pixel centre, [8] focal length, [9..16] distortion, [17..25] M = M1 M2 M3 row-major
// (:74-78, built on the host), [26..28] position, [29..31] polref, [32..35] Stokes vector.
constexpr int CAM_DOUBLES = 40;
__device__ __forceinline__ void gen_camera_ray(const double* cam, int64_t pix, V3& p, Ray10& r) {
    const int64_t res0 = (int64_t)cam[0];
    const double i_pix = (double)(pix % res0 + 1), j_pix = (double)(pix / res0 + 1), f = cam[8];
    const double lx = (cam[6] - i_pix) * cam[4], lz = -(cam[7] - j_pix) * cam[5]; // pixel_location(:,[1 3]), :48-50
    const double d2 = lx * lx + lz * lz;                                         // :52
    double sum = 0, pw = 1; const double u = d2 / (f * f);
    const int nd = (int)cam[3];
    for (int k = 0; k < nd; k++) { pw *= u; sum += cam[9 + k] * pw; }
    const double eff = f * (1 + sum), rr = sqrt(d2) / eff;                       // :54-56
    const int lens = (int)cam[2];
    const double theta = lens == 1 ? rr : (lens == 2 ? asin(rr) : atan(rr));     // :58-67
    const double phi = atan2(-lz, -lx);                                          // :69
    double st, ct, sph, cph; sincos(theta, &st, &ct); sincos(phi, &sph, &cph);
    const V3 c = { st * cph, ct, st * sph };                                     // :70
    r.d = V3{ cam[17] * c.x + cam[18] * c.y + cam[19] * c.z, cam[20] * c.x + cam[21] * c.y + cam[22] * c.z, cam[23] * c.x + cam[24] * c.y + cam[25] * c.z }; // :81
    p = V3{ cam[26], cam[27], cam[28] };
    r.pol = V3{ cam[29], cam[30], cam[31] };
    r.s0 = cam[32]; r.s1 = cam[33]; r.s2 = cam[34]; r.s3 = cam[35];
}

// row of the final-row table that belongs to ray id (rows are in input order)
__device__ __forceinline__ int64_t fin_slot(const KArgs& a, int64_t ray_id) { const int64_t o = ray_id - a.slot_base; return a.id_stride == 1 ? o : o / a.id_stride; }
__device__ __forceinline__ void load_queue_ray(const KArgs& a, int64_t idx, RayState& q) {
    const double* d = a.in_d; const int64_t st = a.in_stride;
    q.p = V3{ d[0 * st + idx], d[1 * st + idx], d[2 * st + idx] };
    q.r.d = V3{ d[3 * st + idx], d[4 * st + idx], d[5 * st + idx] };
    q.r.pol = V3{ d[6 * st + idx], d[7 * st + idx], d[8 * st + idx] };
    q.r.s0 = d[9 * st + idx]; q.r.s1 = d[10 * st + idx]; q.r.s2 = d[11 * st + idx]; q.r.s3 = d[12 * st + idx];
    q.ray_id = a.in_i[idx]; q.parent = a.in_i[st + idx];
    q.hash = (uint32_t)a.in_m[idx]; q.k = a.in_m[st + idx]; q.smix_last = a.in_m[2 * st + idx]; q.six_last = a.in_m[3 * st + idx]; q.flags = a.in_m[4 * st + idx];
}
__device__ __forceinline__ void load_ray(const KArgs& a, int64_t idx, RayState& q, bool wantfin) {
    if (a.mode == 1) { load_queue_ray(a, idx, q); return; }
    q.ray_id = a.ray_id_base + idx * a.id_stride; q.parent = -1; q.hash = 0; q.k = 0; q.smix_last = 1; q.six_last = 0; q.flags = RF_PRIMARY; // RayTracer2.py:258-262
    if (a.mode == 2) gen_point_ray(a, q.ray_id, q.p, q.r);
    else if (a.mode == 3) {
        gen_camera_ray(a.cam, q.ray_id - a.cam_base, q.p, q.r);
        q.r.d = (1.0 / sqrt(dot(q.r.d, q.r.d))) * q.r.d; q.r.pol = (1.0 / sqrt(dot(q.r.pol, q.r.pol))) * q.r.pol; // RayTracer2.py:221-222
    } else {
        const double* d = a.in_d; const int64_t st = a.in_stride;
        q.p = V3{ d[0 * st + idx], d[1 * st + idx], d[2 * st + idx] };
        V3 dir = V3{ d[3 * st + idx], d[4 * st + idx], d[5 * st + idx] }, pol = V3{ d[6 * st + idx], d[7 * st + idx], d[8 * st + idx] };
        q.r.d = (1.0 / sqrt(dot(dir, dir))) * dir; q.r.pol = (1.0 / sqrt(dot(pol, pol))) * pol; // RayTracer2.py:221-222
        q.r.s0 = d[9 * st + idx]; q.r.s1 = d[10 * st + idx]; q.r.s2 = d[11 * st + idx]; q.r.s3 = d[12 * st + idx];
    }
    if (wantfin) { // row of a ray that never interacts
        double* f = a.fin + fin_slot(a, q.ray_id) * FIN;
        f[0] = q.p.x; f[1] = q.p.y; f[2] = q.p.z;
        put_ray10(f, 1, 3, q.r);
#pragma unroll
        for (int j = 13; j < FIN; j++) f[j] = 0.0;
        f[18] = ray_index_of(q, a.P); f[21] = 3.0;
    }
}

__device__ __forceinline__ void store_ray(const Queue& Q, int64_t idx, const RayState& c) {
    double* d = Q.d; const int64_t st = Q.cap;
    d[0 * st + idx] = c.p.x; d[1 * st + idx] = c.p.y; d[2 * st + idx] = c.p.z;
    d[3 * st + idx] = c.r.d.x; d[4 * st + idx] = c.r.d.y; d[5 * st + idx] = c.r.d.z;
    d[6 * st + idx] = c.r.pol.x; d[7 * st + idx] = c.r.pol.y; d[8 * st + idx] = c.r.pol.z;
    d[9 * st + idx] = c.r.s0; d[10 * st + idx] = c.r.s1; d[11 * st + idx] = c.r.s2; d[12 * st + idx] = c.r.s3;
    Q.i[idx] = c.ray_id; Q.i[st + idx] = c.parent;
    Q.m[idx] = (int32_t)c.hash; Q.m[st + idx] = c.k; Q.m[2 * st + idx] = c.smix_last; Q.m[3 * st + idx] = c.six_last; Q.m[4 * st + idx] = c.flags;
}

// Tally-only single-child runs (the level kernels' lean step) keep a 96-byte queue record: parent slot and flags mean
// nothing there (every ray is its own primary and never marked reflected), the step is a kernel argument, and the path
// hash is the same for every ray of a step (one child per interaction: KArgs::lean_hash).  The record is ONE ROW of six
// 16-byte pieces in the queue's `d` array (row i at byte 96 i):
//   piece 0 (p.x, p.y)  1 (p.z, d.a)  2 (d.b, {ray id - first id of the pass : 32 | word : 32})  3 (pol.a, pol.b)  4 (s0, s1)  5 (s2, s3)
// word: bits 0-23 |smix_last|, 31 its sign, 30: six_last is 0 - after a scatter in the bulk and for initial rays - and not
// |smix_last|; bits 24-26 / 27-29 the codes of the two unit vectors (below).  Pieces 0-2 are all the nearest-hit kernel reads.
// Rows instead of planes (round 3): a ray picked through a sparse list costs the 64-byte sectors its row touches (128 B),
// whatever the list's density - with eleven planes of doubles a launch that took one ray in four read 4.3 times its bytes.
__device__ __forceinline__ int32_t lean_pack(int smix, int six) { return (int32_t)((uint32_t)(smix < 0 ? -smix : smix) | (smix < 0 ? 0x80000000u : 0u) | (six == 0 ? 0x40000000u : 0u)); }
__device__ __forceinline__ int lean_smix(int32_t w) { const int m = (int)((uint32_t)w & 0xFFFFFFu); return w < 0 ? -m : m; }
__device__ __forceinline__ int lean_six(int32_t w) { return ((uint32_t)w & 0x40000000u) ? 0 : (int)((uint32_t)w & 0xFFFFFFu); }
// Direction and polarisation reference are unit vectors: two components travel, the largest one is rebuilt from them
// (error ~1e-16).  Code: bits 0-1 which component is left out, bit 2 its sign; 3 = the vector is not a unit vector (zero: a
// polarisation reference along the direction; NaN) and goes as (x, y, 0).  `lossy`: a finite vector that is not unit and has
// a z component - nothing the kernels hand on is one (every producer normalises); the caller reports it (overflow code 3)
// instead of dropping the component in silence.
__device__ __forceinline__ uint32_t uv_pack(V3 v, double& a_, double& b_, bool& lossy) {
    const double ax = fabs(v.x), ay = fabs(v.y), az = fabs(v.z);
    const uint32_t j = (ax >= ay && ax >= az) ? 0u : (ay >= az ? 1u : 2u);
    const double n2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const bool unit = fabs(n2 - 1.0) < 1e-6; // (what the kernels hand on is unit to rounding)
    const double vj = j == 0u ? v.x : (j == 1u ? v.y : v.z);
    a_ = unit ? (j == 0u ? v.y : (j == 1u ? v.z : v.x)) : v.x;
    b_ = unit ? (j == 0u ? v.z : (j == 1u ? v.x : v.y)) : v.y;
    lossy = lossy | (!unit & (n2 < INFINITY) & (v.z != 0.0));
    return unit ? (j | (vj < 0 ? 4u : 0u)) : 3u;
}
__device__ __forceinline__ V3 uv_unpack(uint32_t code, double a_, double b_) {
    const uint32_t j = code & 3u;
    double w = rt_fsqrt(fmax(fma(-a_, a_, fma(-b_, b_, 1.0)), 0.0));
    w = (code & 4u) ? -w : w;
    if (j == 3u) return V3{ a_, b_, 0.0 };
    return j == 0u ? V3{ w, a_, b_ } : (j == 1u ? V3{ b_, w, a_ } : V3{ a_, b_, w });
}
typedef double __attribute__((ext_vector_type(2))) d2;
constexpr int LROW = 6;                 // 16-byte pieces of a lean row
struct LeanRow { d2 v[LROW]; };
// rid: the ray's number within the pass (its id = first id of the pass + rid * stride); a child takes its parent's as it is
__device__ __forceinline__ LeanRow lean_row_pack(const RayState& c, uint32_t rid, bool& lossy) {
    double da, db, pa, pb;
    const uint32_t cd = uv_pack(c.r.d, da, db, lossy), cp = uv_pack(c.r.pol, pa, pb, lossy);
    const uint32_t w = (uint32_t)lean_pack(c.smix_last, c.six_last) | (cd << 24) | (cp << 27);
    LeanRow r;
    r.v[0] = d2{ c.p.x, c.p.y }; r.v[1] = d2{ c.p.z, da }; r.v[2] = d2{ db, __longlong_as_double((long long)(((unsigned long long)w << 32) | rid)) };
    r.v[3] = d2{ pa, pb }; r.v[4] = d2{ c.r.s0, c.r.s1 }; r.v[5] = d2{ c.r.s2, c.r.s3 };
    return r;
}
__device__ __forceinline__ uint32_t lean_row_rid(const LeanRow& r) { return (uint32_t)(unsigned long long)__double_as_longlong(r.v[2].y); }
__device__ __forceinline__ int32_t lean_row_word(const LeanRow& r) { return (int32_t)((unsigned long long)__double_as_longlong(r.v[2].y) >> 32); }
__device__ __forceinline__ void lean_row_unpack(const LeanRow& r, const KArgs& a, RayState& q) {
    const int32_t w = lean_row_word(r); const uint32_t rid = lean_row_rid(r);
    q.p = V3{ r.v[0].x, r.v[0].y, r.v[1].x };
    q.r.d = uv_unpack(((uint32_t)w >> 24) & 7u, r.v[1].y, r.v[2].x);
    q.r.pol = uv_unpack(((uint32_t)w >> 27) & 7u, r.v[3].x, r.v[3].y);
    q.r.s0 = r.v[4].x; q.r.s1 = r.v[4].y; q.r.s2 = r.v[5].x; q.r.s3 = r.v[5].y;
    q.ray_id = a.slot_base + (int64_t)rid * a.id_stride; q.parent = -1; q.hash = a.lean_hash; q.k = a.step; q.smix_last = lean_smix(w); q.six_last = lean_six(w); q.flags = RF_PRIMARY;
}
// A lane moves its own row: six 16-byte loads or stores at a stride of 96 bytes between lanes.  Measured against moving
// rows with the whole wave (six lanes per row, whole rows per instruction, an LDS tile to turn pieces into one row per lane;
// also with the incoming pieces sent straight to LDS, global_load_lds_dwordx4): as a pure copy the wave-wide way streams 13 %
// faster (tools/mb_queue.hip) - but these launches are not bound by how their bytes move (profiles/r3_layout_experiment.txt:
// half the bytes, the same cycles), and beside the other lane's nearest-hit launch the LDS traffic of the transposition
// cost both kernels 2-3 %.
__device__ __forceinline__ void row_load(const d2* rows, int64_t idx, LeanRow& r) {
#pragma unroll
    for (int f = 0; f < LROW; f++) r.v[f] = rows[idx * LROW + f];
}
__device__ __forceinline__ void row_store(d2* rows, int64_t idx, const LeanRow& r) {
#pragma unroll
    for (int f = 0; f < LROW; f++) rows[idx * LROW + f] = r.v[f];
}

__device__ __forceinline__ void store_ray_lean(const Queue& Q, int64_t idx, const RayState& c, uint32_t rid, int* overflow) {
    bool lossy = false; const LeanRow r = lean_row_pack(c, rid, lossy);
    row_store((d2*)Q.d, idx, r);
    if (lossy) *overflow = 3;
}
__device__ __forceinline__ void load_queue_ray_lean(const KArgs& a, int64_t idx, RayState& q) {
    LeanRow r; row_load((const d2*)a.in_d, idx, r);
    lean_row_unpack(r, a, q);
}
// slot allocation for the lanes that are active here: one atomic per wave, consecutive slots in lane order
__device__ __forceinline__ unsigned long long wave_alloc(unsigned long long* counter) {
    unsigned long long act = __ballot(1);
    int lane = lane_id(), leader = __ffsll((long long)act) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(act));
    unsigned int lo = __shfl((unsigned int)base, leader), hi = __shfl((unsigned int)(base >> 32), leader);
    base = ((unsigned long long)hi << 32) | lo;
    return base + (unsigned long long)__popcll(act & ((1ull << lane) - 1ull));
}

// the same from a counter in LDS
__device__ __forceinline__ uint32_t wave_alloc_lds(uint32_t* counter) {
    unsigned long long act = __ballot(1);
    int lane = lane_id(), leader = __ffsll((long long)act) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(act));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
}

// Block-uniform arguments of the out-of-line nearest-hit search, at the start of dynamic LDS.  A non-kernel function
// receives its arguments in vector registers, which would turn every geometry read into a per-lane load; read from
// here and made scalar with readfirstlane they stay in SGPRs and the table reads stay scalar loads.
struct LdsHdr { unsigned long long surf, recs, smeta, prog; double min_travel; int S, compat, base, nprog; };
constexpr int HDR_DOUBLES = 8;
constexpr int TQ_DOUBLES = (BLOCK / 64) * TQ_SLOTS * TQ_STRIDE; // per-wave torus queues, right behind the header
static_assert(sizeof(LdsHdr) <= HDR_DOUBLES * 8, "LdsHdr");
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) { return ((unsigned long long)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v); }
} // namespace
namespace rt {
#ifdef RT_OOL_FIND_HIT
template <bool GEN> __device__ __noinline__ Hit find_hit_ool_t(V3 sp, V3 d, int six_last, bool act) {
    extern __shared__ double smem[];
    const LdsHdr* H = (const LdsHdr*)smem;
    Geom G; Prm P;
    G.usurf = (udptr)uni64(H->surf); G.urec = (udptr)uni64(H->recs); G.smeta = (uuptr)uni64(H->smeta);
    G.surf = nullptr; G.lut = nullptr; G.lut_prec = 0; G.S = (int)uni32((uint32_t)H->S);
    G.uprog = (udptr)uni64(H->prog); G.nprog = (int)uni32((uint32_t)H->nprog);
    G.tq = smem + HDR_DOUBLES + (int)uni32(threadIdx.x / 64u) * (TQ_SLOTS * TQ_STRIDE);
    P.compat = (int)uni32((uint32_t)H->compat); P.base = (int)uni32((uint32_t)H->base);
    unsigned long long mt = uni64((unsigned long long)__double_as_longlong(H->min_travel));
    P.min_travel = __longlong_as_double((long long)mt);
    P.K = 0; P.flags = 0; P.maxiter = 0; P.thrA = P.thrB = P.tir = 0; P.key0 = P.key1 = 0;
    Hit h;
#ifdef RT_PROFILE
    h.pm = h.ps = h.pc = h.pn = h.pf = h.pt_ = 0;
#endif
    find_hit<GEN>(G, P, sp, d, six_last, act, h);
    return h;
}
__device__ __noinline__ Hit find_hit_ool(V3 sp, V3 d, int six_last, bool act) { return find_hit_ool_t<true>(sp, d, six_last, act); }
#endif
} // namespace rt
namespace {
#ifndef RT_WAVES
#define RT_WAVES 2 /* the instances that keep records or final rows, or split rays: the whole register file as well (at 4 waves per SIMD they spilled 488-746 VGPRs: runs with final rows 1.5 to 2 times slower, profiles/r3_level_threshold.txt) */
#endif
#ifndef RT_LEAN_TRACE_WAVES
#define RT_LEAN_TRACE_WAVES 2
#endif
// CF: flag set this instance is compiled for (FLAG_SINGLE exact; FLAG_FULL/FLAG_FINAL = "may be requested", 0 = never)
// LEAN: tally-only single-child step (bounce_phys_single); GEN: the program has surfaces outside every frame
// UNI (LEAN only): 1 = the geometry has UNIFIED surfaces, 2 = it has none (no such code).  The lean instances are what small
// runs and the tails of large ones are traced by, where a wave's time per bounce is all that counts: they take the whole
// register file (2 waves per SIMD), the nearest-hit search in line, and with UNI = 1 the wave draws its facet normals together
// (unified_wave: every lane goes through the step, `alive` marks the rays) - out of line under a 128-register budget, with the
// rough-surface loop run lane by lane, a bounce took a wave 69 us (round 2: 251 spilled VGPRs), three times what it takes now.
template <int CF, bool LEAN = false, bool GEN = true, int UNI = 0> __global__ void __launch_bounds__(BLOCK, LEAN ? RT_LEAN_TRACE_WAVES : RT_WAVES) k_trace(const KArgs a) {
    extern __shared__ double smem[];
    // LDS image: surface rows (per-lane gather of the winning surface) | tally rows | per-step counters
    double* s_surf = smem + HDR_DOUBLES + TQ_DOUBLES;
    double* s_tab = s_surf + a.S * ROW;
    const int ntab = a.rows_lds * 5 * a.S * 2;
    uint32_t* s_live = (uint32_t*)(s_tab + ntab);
    uint32_t* s_scat = s_live + a.P.K;
    if (threadIdx.x == 0) {
        LdsHdr* H = (LdsHdr*)smem;
        H->surf = (unsigned long long)a.surf; H->recs = (unsigned long long)a.recs; H->smeta = (unsigned long long)a.smeta;
        H->min_travel = a.P.min_travel; H->S = a.S; H->compat = a.P.compat; H->base = a.P.base; H->prog = (unsigned long long)a.prog; H->nprog = a.nprog;
    }
    for (int i = threadIdx.x; i < a.S * ROW; i += BLOCK) s_surf[i] = a.surf[i];
    for (int i = threadIdx.x; i < ntab; i += BLOCK) s_tab[i] = 0.0;
    for (int i = threadIdx.x; i < 2 * a.P.K; i += BLOCK) s_live[i] = 0u;
    __syncthreads();

    Geom G{ (udptr)a.surf, (udptr)a.recs, (uuptr)a.smeta, s_surf, a.S, a.lut, a.lut_prec, smem + HDR_DOUBLES + (threadIdx.x / 64) * (TQ_SLOTS * TQ_STRIDE), (udptr)a.prog, a.nprog };
    const Prm P = a.P;
    const int lane = lane_id();
    RayState q;
    if constexpr (LEAN) { q.p = V3{ 0, 0, 0 }; q.r.d = V3{ 0, 0, 1 }; q.r.pol = V3{ 1, 0, 0 }; q.r.s0 = q.r.s1 = q.r.s2 = q.r.s3 = 0; q.ray_id = 0; q.parent = -1; q.hash = 0; q.k = 0; q.smix_last = 1; q.six_last = 0; q.flags = 0; } // idle lanes go through the step too
    bool alive = false, more = true;
    const bool full = (CF & FLAG_FULL) && (P.flags & FLAG_FULL), wantfin = (CF & FLAG_FINAL) && (P.flags & FLAG_FINAL);
    constexpr int BF = CF & FLAG_SINGLE; // what the per-ray arithmetic needs to know

#ifdef RT_PROFILE
    long long pt[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }; long long t0, t1;
#define PROF_T0() t0 = (long long)__builtin_readcyclecounter()
#define PROF_ADD(i) { t1 = (long long)__builtin_readcyclecounter(); pt[i] += t1 - t0; t0 = t1; }
#else
#define PROF_T0()
#define PROF_ADD(i)
#endif
    for (;;) {
        PROF_T0();
        if (more) {
            unsigned long long need = __ballot(!alive);
            int cnt = __popcll(need);
            if (cnt >= a.refill_min || __ballot(alive) == 0ull) {
                int leader = __ffsll((long long)need) - 1;
                unsigned long long base = 0;
                if (lane == leader) base = atomicAdd(a.head, (unsigned long long)cnt);
                unsigned int lo = __shfl((unsigned int)base, leader), hi = __shfl((unsigned int)(base >> 32), leader);
                base = ((unsigned long long)hi << 32) | lo;
                if (base >= (unsigned long long)a.n_in) more = false;
                else {
                    if (!alive) {
                        unsigned long long idx = base + (unsigned long long)__popcll(need & ((1ull << lane) - 1ull));
                        if (idx < (unsigned long long)a.n_in) { load_ray(a, (int64_t)idx, q, wantfin); alive = true; }
                    }
                    if (base + (unsigned long long)cnt >= (unsigned long long)a.n_in) more = false;
                }
            }
        }
        PROF_ADD(0)
        if (__ballot(alive) == 0ull) { if (!more) break; continue; }
#ifdef RT_PROFILE
        pt[5] += 1; pt[6] += __popcll(__ballot(alive));
#endif
        Hit h; h.l = INFINITY; h.surf = -1; h.orient = 0;
        if constexpr (LEAN) { q.r.d = normalized_if(q.r.d); find_hit<GEN>(G, P, q.p, q.r.d, q.six_last, alive, h); } // RayToShape.py:11-14
        else if constexpr (GEN) bounce_hit(G, P, q, alive, h); // all lanes: idle ones still serve the wave's torus queue
        else { q.r.d = normalized_if(q.r.d); h = find_hit_ool_t<false>(q.p, q.r.d, q.six_last, alive); }
#ifdef RT_PROFILE
        pt[4] += h.ps; pt[7] += h.pm; pt[8] += h.pc; pt[9] += h.pn; pt[10] += h.pf; pt[11] += h.pt_;
#endif
        PROF_ADD(1)
        if constexpr (LEAN) {
            LeanBounce b; bounce_phys_single<UNI>(G, P, q, h, b, alive); // every lane (UNI = 1: the wave works through the rough-surface loop together)
            PROF_ADD(2)
            if (alive) {
                const int k = q.k;
                atomicAdd(&s_live[k], 1u);
                if (!b.scat) { tally_add(a, s_tab, k, 2, q.smix_last, b.inc_intensity); alive = false; } // escaped: RayTracer2.py:485-486
                else {
                    atomicAdd(&s_scat[k], 1u);
                    if (b.surf_sc) tally_add(a, s_tab, k, 0, b.smix_next, b.surf_abs);
                    tally_add(a, s_tab, k, 1, b.smix_next, b.bulk_abs);
                    tally_add(a, s_tab, k, b.fa ? 4 : 3, b.smix_next, b.s0_child);
                    if (k + 1 >= P.K || !b.fa) alive = false;
                    else q = lean_child(q, b);
                }
            }
            PROF_ADD(3)
            continue;
        }
        Bounce b; b.scat = 0;
        if (alive) bounce_phys<BF>(G, P, q, h, b);
        PROF_ADD(2)
        if (alive) {
            const int k = q.k;
            atomicAdd(&s_live[k], 1u);
            if (!b.scat) { // escaped: RayTracer2.py:485-486
                tally_add(a, s_tab, k, 2, q.smix_last, b.inc_intensity);
                alive = false;
            } else {
                atomicAdd(&s_scat[k], 1u);
                if (b.surf_sc) tally_add(a, s_tab, k, 0, b.smix_next, b.surf_abs);
                tally_add(a, s_tab, k, 1, b.smix_next, b.bulk_abs);
                tally_add(a, s_tab, k, 3, b.smix_next, (b.fa ? 0.0 : b.refr.s0) + (b.fb ? 0.0 : b.refl.s0));
                tally_add(a, s_tab, k, 4, b.smix_next, (b.fa ? b.refr.s0 : 0.0) + (b.fb ? b.refl.s0 : 0.0));
                int64_t myslot = -1;
                if (full) {
                    unsigned long long slot = wave_alloc(a.rec_count);
                    if (slot < (unsigned long long)a.rec_cap) {
                        fill_record(q, b, P, a.rec + slot, (size_t)a.rec_cap);
                        a.rec_step[slot] = k; a.rec_ray[slot] = q.ray_id; a.rec_parent[slot] = q.parent; a.rec_flags[slot] = q.flags;
                        myslot = (int64_t)slot;
                    } else *a.overflow = 1;
                }
                if (wantfin && (q.flags & RF_PRIMARY)) fill_final(q, b, P, a.fin + fin_slot(a, q.ray_id) * FIN);
                const bool last = (k + 1 >= P.K);
                if (last || !(b.fa || b.fb)) alive = false;
                else {
                    if (b.fa && b.fb) { // split: keep the refracted child, queue the reflected one
                        RayState c = child_of(q, b, true, true);
                        c.parent = myslot;
                        unsigned long long slot = wave_alloc(a.tail);
                        if (slot < (unsigned long long)a.out.cap) store_ray(a.out, (int64_t)slot, c);
                        else *a.overflow = 2;
                    }
                    RayState c = child_of(q, b, !b.fa, b.fa != 0);
                    c.parent = myslot;
                    q = c;
                }
            }
        }
        PROF_ADD(3)
    }
#ifdef RT_PROFILE
    if (lane == 0 && a.prof) for (int i = 0; i < 12; i++) atomicAdd(&a.prof[i], (unsigned long long)pt[i]);
#endif
    __syncthreads();
    for (int i = threadIdx.x; i < ntab; i += BLOCK) { double v = s_tab[i]; if (v != 0.0) atomicAdd(&a.table[i], v); }
    for (int i = threadIdx.x; i < a.P.K; i += BLOCK) {
        uint32_t v = s_live[i]; if (v) atomicAdd(&a.n_live[i], (unsigned long long)v);
        uint32_t w = s_scat[i]; if (w) atomicAdd(&a.n_scat[i], (unsigned long long)w);
    }
}

// host layout (start[n][3], rays[n][10]) -> SoA [13][n]

// ------------------------------------------------------------------------------------------------ level-synchronous path
// Single-child runs (every ray has at most one child per step) are traced step by step, like the reference's own
// loop (python/RayTracer2.py:275): the live rays of step k sit in a queue in HBM, k_level_hit finds their nearest hits
// and k_level_phys applies the physics and writes the surviving children, compacted, into the queue of step k+1.
// The two halves want different things from the machine: the nearest-hit search is a long chain of uniform (scalar)
// loads with little per-lane state, so it runs at high occupancy to hide their latency; the physics needs ~130 FP64
// registers' worth of state and few waves.  Fused into one persistent kernel both ran at 4 waves/SIMD and the physics
// spilled ~1.2 kB per interaction through scratch to HBM - more than the ~0.4 kB per interaction the queues cost here.
// Both kernels are grid-stride over the live count, which they read from device memory (no host round trip per step).
#ifndef RT_HIT_WAVES
#define RT_HIT_WAVES 5
#endif
#ifndef RT_HITF_WAVES
#define RT_HITF_WAVES 5
#endif
#ifndef RT_PHYS_WAVES
#define RT_PHYS_WAVES 3
#endif
#ifndef RT_P1_WAVES
#define RT_P1_WAVES RT_PHYS_WAVES /* the common launch of a step with two (k_level_phys PASS 1) */
#endif
#ifndef RT_LEAN_WAVES
#define RT_LEAN_WAVES 2 /* the software-pipelined tally-only launches: a second queue record per lane in flight */
#endif
#ifndef RT_UNI_WAVES
#define RT_UNI_WAVES 2 /* the pass that runs the UNIFIED rough-surface loop in line */
#endif
// what the nearest-hit kernel needs, and nothing else (kernel arguments live in SGPRs)
struct HitArgs {
    const double* in_d; const int32_t* in_m; int64_t in_stride, n_in;
    const double* prog; int nprog; const double* recs; const double* surf; const uint32_t* smeta; int S;
    double min_travel; int base, compat;
    double* hit_l; int32_t* hit_so;
    const uint32_t* seg_in; int64_t seg_cap; int split; // split blocks share one segment
    int lean;                    // the queue holds lean records (store_ray_lean)
};

// initial rays (host layout as SoA, or generated point source) -> the ray queue both level kernels read
template <int CF> __global__ void __launch_bounds__(BLOCK) k_level_init(const KArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= a.n_in) return;
    const bool wantfin = (CF & FLAG_FINAL) && (a.P.flags & FLAG_FINAL);
    RayState q; load_ray(a, idx, q, wantfin);
    q.r.d = normalized_if(q.r.d); // RayToShape.py:11-14
    const int64_t sg = idx / a.seg_fill, off = idx % a.seg_fill;
    if (a.lean) store_ray_lean(a.out, sg * a.seg_cap + off, q, (uint32_t)idx, a.overflow); else store_ray(a.out, sg * a.seg_cap + off, q);
    if (off == 0) a.seg_out[sg] = (uint32_t)(a.n_in - idx < a.seg_fill ? a.n_in - idx : a.seg_fill);
}
// segments -> dense queue (hand-over to the persistent kernel)
__global__ void __launch_bounds__(BLOCK) k_level_compact(const KArgs a) {
    __shared__ unsigned long long s_base;
    const int64_t n = a.seg_in[blockIdx.x], base = (int64_t)blockIdx.x * a.seg_cap;
    if (n == 0) return;
    if (threadIdx.x == 0) s_base = atomicAdd(a.tail, (unsigned long long)n);
    __syncthreads();
    const int64_t ob = (int64_t)s_base;
    const bool lean = a.lean != 0;
    for (int64_t i = threadIdx.x; i < n; i += BLOCK) {
        RayState q;
        if (lean) load_queue_ray_lean(a, base + i, q); else load_queue_ray(a, base + i, q);
        store_ray(a.out, ob + i, q);
    }
}

// rt_trace_device_noalloc: the "nothing scattered" rule of python/RayTracer2.py:405-407 (a step in which no ray scatters
// tallies nothing), which run_trace() applies on the host, applied to the caller's device table; one block per step
// tallies of a pass that has ended well, into the run's (run_trace: passes that may have to be repeated)
__global__ void __launch_bounds__(BLOCK) k_table_add(double* table, const double* pass_table, int64_t n, unsigned long long* steps, const unsigned long long* pass_steps, int64_t m) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) { const double v = pass_table[i]; if (v != 0.0) atomicAdd(&table[i], v); }
    else if (i < n + m) { const unsigned long long c = pass_steps[i - n]; if (c) atomicAdd(&steps[i - n], c); }
}
__global__ void __launch_bounds__(BLOCK) k_level_finish(double* table, const unsigned long long* n_scat, int row_elems) {
    if (n_scat[blockIdx.x] != 0ull) return;
    double* row = table + (size_t)blockIdx.x * row_elems;
    for (int i = threadIdx.x; i < row_elems; i += BLOCK) row[i] = 0.0;
}

// GENERIC: the program contains surfaces that fit no frame (see rt_device.hpp find_hit)
template <bool GENERIC> __global__ void __launch_bounds__(BLOCK, GENERIC ? RT_HIT_WAVES : RT_HITF_WAVES) k_level_hit(const HitArgs a) {
    extern __shared__ double smem[]; // per-wave torus queues
    Geom G{ (udptr)a.surf, (udptr)a.recs, (uuptr)a.smeta, nullptr, a.S, nullptr, 0, smem + (threadIdx.x / 64) * (TQ_SLOTS * TQ_STRIDE), (udptr)a.prog, a.nprog };
    Prm P; P.K = 0; P.flags = 0; P.maxiter = 0; P.thrA = P.thrB = P.tir = 0; P.key0 = P.key1 = 0;
    P.compat = a.compat; P.base = a.base; P.min_travel = a.min_travel;
    const int sg = (int)blockIdx.x / a.split, part = (int)blockIdx.x % a.split;
    const int64_t st = a.in_stride, seg0 = (int64_t)sg * a.seg_cap, n = seg0 + (int64_t)a.seg_in[sg];
    for (int64_t base = seg0 + (int64_t)part * BLOCK + (threadIdx.x & ~63); base < n; base += (int64_t)a.split * BLOCK) { // wave-uniform trip count
        const int64_t idx = base + lane_id();
        const bool act = idx < n;
        V3 p = { 0, 0, 0 }, d = { 0, 0, 1 }; int six = 0;
        if (act) {
            const double* q = a.in_d;
            if (a.lean) { // the first three pieces of the ray's row
                const d2* row = (const d2*)q + idx * LROW; const d2 r0 = row[0], r1 = row[1], r2 = row[2];
                const int32_t w = (int32_t)((unsigned long long)__double_as_longlong(r2.y) >> 32);
                p = V3{ r0.x, r0.y, r1.x }; six = lean_six(w); d = normalized_if(uv_unpack(((uint32_t)w >> 24) & 7u, r1.y, r2.x));
            } else { p = V3{ q[0 * st + idx], q[1 * st + idx], q[2 * st + idx] }; d = normalized_if(V3{ q[3 * st + idx], q[4 * st + idx], q[5 * st + idx] }); six = a.in_m[3 * st + idx]; } // RayToShape.py:11-14
        }
        Hit h;
#ifdef RT_PROFILE
        h.pm = h.ps = h.pc = h.pn = h.pf = h.pt_ = 0;
#endif
        find_hit<GENERIC>(G, P, p, d, six, act, h);
        if (act) { a.hit_l[idx] = h.l; a.hit_so[idx] = h.surf < 0 ? -1 : h.surf * 4 + h.orient + 1; }
    }
}

// PASS 0: every ray of the segment.  PASS 1: rays ending on a UNIFIED surface, and rays that scatter in the bulk before
// they reach theirs, are only listed (KArgs::defer); the others are worked through from a dense list.
// PASS 2: the listed rays, children appended behind those of pass 1; for CF == FLAG_SINGLE (tallies only) the wave
// works through the UNIFIED model together (unified_wave), otherwise every lane runs it in line.
// PASS 3: PASS 0 for a geometry without UNIFIED surfaces (no such code in the kernel).
#ifndef RT_OUT_WAVES
#define RT_OUT_WAVES 2 /* the physics launches that write records or final rows: at 3 waves per SIMD they spill 16-155 VGPRs, at 2 none to 49 (image pipeline -4 % at 33.5 M pixels, -6 % at 4 M; record streaming unchanged) */
#endif
#define RT_PHYS_BOUNDS(CF, PASS) (PASS == 2 ? RT_UNI_WAVES : (CF == FLAG_SINGLE ? RT_LEAN_WAVES : ((CF & (FLAG_FULL | FLAG_FINAL)) ? RT_OUT_WAVES : (PASS == 1 ? RT_P1_WAVES : RT_PHYS_WAVES))))
template <int CF, int PASS = 0> __global__ void __launch_bounds__(BLOCK, RT_PHYS_BOUNDS(CF, PASS)) k_level_phys(const KArgs a) {
    extern __shared__ double smem[];
#ifdef RT_PROFILE
    long long pp[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; long long pt0 = (long long)__builtin_readcyclecounter(), pt1;
#define PP_ADD(i) { pt1 = (long long)__builtin_readcyclecounter(); pp[i] += pt1 - pt0; pt0 = pt1; }
#else
#define PP_ADD(i)
#endif
    if constexpr (PASS == 2) { if (a.defer_n[2 * blockIdx.x] + a.defer_n[2 * blockIdx.x + 1] == 0) return; } // nothing put aside in this segment
    // LDS image: surface rows (per-lane gather of the winning surface) | this step's tally row | step counters
    double* s_surf = smem;
    double* s_tab = s_surf + a.S * ROW;
    const int ntab = 5 * a.S * 2;
    uint32_t* s_cnt = (uint32_t*)(s_tab + ntab);
    for (int i = threadIdx.x; i < a.S * ROW; i += BLOCK) s_surf[i] = a.surf[i];
    for (int i = threadIdx.x; i < ntab; i += BLOCK) s_tab[i] = 0.0;
    const uint32_t child0 = PASS == 2 ? a.seg_out[blockIdx.x] : 0u;
    if (threadIdx.x < 8) s_cnt[threadIdx.x] = threadIdx.x == 2 ? child0 : 0u; // live, scattered, children, deferred (UNIFIED), dense, deferred (bulk scatters), next tile (PASS 2)
    __syncthreads();
    PP_ADD(0)
    Geom G{ (udptr)a.surf, (udptr)a.recs, (uuptr)a.smeta, s_surf, a.S, a.lut, a.lut_prec, nullptr, (udptr)a.prog, a.nprog };
    const Prm P = a.P;
    const bool full = (CF & FLAG_FULL) && (P.flags & FLAG_FULL), wantfin = (CF & FLAG_FINAL) && (P.flags & FLAG_FINAL);
    constexpr int BF = CF & FLAG_SINGLE;
    const int k = a.step;
    KArgs t = a; t.rows_lds = 1; // tally_add(): row 0 of the LDS image is step k
    const int64_t seg0 = (int64_t)blockIdx.x * a.seg_cap;
    // PASS 2 walks the front of its list (padded to whole waves), then the back
    const int64_t nf = PASS == 2 ? (int64_t)a.defer_n[2 * blockIdx.x] : 0, nb = PASS == 2 ? (int64_t)a.defer_n[2 * blockIdx.x + 1] : 0, nfp = (nf + 63) & ~(int64_t)63;
    int64_t n = seg0 + (PASS == 2 ? nfp + nb : (int64_t)a.seg_in[blockIdx.x]);
    constexpr bool listed = PASS == 2 || PASS == 1;
    if constexpr (PASS == 1) { // sort the segment's rays into the two lists; this launch then takes the dense one
        // Only the surface type decides here.  Whether a ray scatters in the bulk before it reaches its surface is found
        // out by the physics step itself (bounce_phys_single<2, 2>: it has to make that draw anyway), which then puts the
        // ray on the other launch's list from the back.  (Making the draw here cost a Philox block and a logarithm per
        // ray in a scattering medium a second time: the sort was 28 % of this launch's wave-cycles, r2j.)
        constexpr int SORT_U = 4;
        for (int64_t base = seg0 + (threadIdx.x & ~63); base < n; base += SORT_U * BLOCK) {
            int so[SORT_U];
#pragma unroll
            for (int u = 0; u < SORT_U; u++) { const int64_t idx = base + u * BLOCK + lane_id(); so[u] = idx < n ? a.hit_so[idx] : -2; }
#pragma unroll
            for (int u = 0; u < SORT_U; u++) {
                if (so[u] == -2) continue;
                const int64_t idx = base + u * BLOCK + lane_id();
                const bool heavy = so[u] >= 0 && (int)s_surf[(so[u] >> 2) * ROW + 1] == ST_UNIFIED;
                // the list of the other launch: UNIFIED rays from the front, bulk scatters from the back, so that its waves are of one kind
                if (heavy) a.defer[seg0 + (int64_t)wave_alloc_lds(&s_cnt[3])] = (uint32_t)(idx - seg0);
                else a.dense[seg0 + (int64_t)wave_alloc_lds(&s_cnt[4])] = (uint32_t)(idx - seg0);
            }
        }
        PP_ADD(1)
        __syncthreads();
        PP_ADD(2)
        n = seg0 + (int64_t)s_cnt[4];
    }
    const uint32_t* list = PASS == 2 ? a.defer : a.dense;
    if constexpr (CF == FLAG_SINGLE && PASS != 2) {
        // Tally-only step of the rays that take the common path, software-pipelined: the queue record of the ray a lane
        // works on next (and the list entry of the one after) is requested before the physics of the current one starts, so
        // the ~2 us an HBM round trip takes under load pass during ~1 500 vector instructions instead of in front of them.
        // (Measured on the unpipelined loop, r2d: its waves spent 55 % of their cycles in s_waitcnt at 3 waves/SIMD.)
        struct Raw { LeanRow row; double hl; int32_t so; };
        const d2* rows = (const d2*)a.in_d; const int lane = lane_id();
        // list entries travel as loaded (no arithmetic on them until the iteration that needs the slot: a use would make the wave wait for the load)
        auto entry_of = [&](int64_t base) -> uint32_t { const int64_t i = base + lane; return (listed && i < n) ? list[i] : 0u; };
        auto slot_of = [&](int64_t base, uint32_t e) -> int64_t { const int64_t i = base + lane; if (i >= n) return -1; return listed ? seg0 + (int64_t)e : i; };
        auto fetch = [&](int64_t idx, Raw& r) {
            if (idx < 0) return;
            r.so = a.hit_so[idx]; r.hl = a.hit_l[idx];
            row_load(rows, idx, r.row);
        };
        const int64_t first = seg0 + (threadIdx.x & ~63);
        int64_t idx1 = slot_of(first, entry_of(first));
        uint32_t e2 = entry_of(first + BLOCK);
        Raw nxt = {}; fetch(idx1, nxt);
        __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0), as an instruction the compiler's own wait-count pass sees (inline asm is opaque to it) */ // the loop is entered as it is re-entered: nothing in flight (or the compiler makes every iteration wait at the top for what only the first has pending)
        d2* const out_rows = (d2*)a.out.d;
        for (int64_t base = first; base < n; base += BLOCK) {
            const Raw cur = nxt; const int64_t idx = idx1;
            idx1 = slot_of(base + BLOCK, e2);
            fetch(idx1, nxt);
            e2 = entry_of(base + 2 * BLOCK); // behind the row's loads: the wave's memory counter is in order, and what waits for them must not wait for this
            LeanRow crow; int64_t at = -1; bool lossy = false;
            if (idx >= 0) {
                RayState q; lean_row_unpack(cur.row, a, q);
                q.r.d = normalized_if(q.r.d);
                Hit h; h.l = cur.hl; h.surf = cur.so < 0 ? -1 : (cur.so >> 2); h.orient = cur.so < 0 ? 0 : (cur.so & 3) - 1;
                LeanBounce b;
                if constexpr (PASS == 1) bounce_phys_single<2, 2>(G, P, q, h, b, true);
                else bounce_phys_single<PASS == 3 ? 2 : 0>(G, P, q, h, b, true);
                if (PASS == 1 && b.scat == 2) a.defer[seg0 + a.seg_cap - 1 - (int64_t)wave_alloc_lds(&s_cnt[5])] = (uint32_t)(idx - seg0); // scatters in the bulk: the other launch's
                else atomicAdd(&s_cnt[0], 1u);
                if (PASS == 1 && b.scat == 2) {}
                else if (!b.scat) tally_add(t, s_tab, 0, 2, q.smix_last, b.inc_intensity); // escaped: RayTracer2.py:485-486
                else {
                    atomicAdd(&s_cnt[1], 1u);
                    if (b.surf_sc) tally_add(t, s_tab, 0, 0, b.smix_next, b.surf_abs);
                    tally_add(t, s_tab, 0, 1, b.smix_next, b.bulk_abs);
                    tally_add(t, s_tab, 0, b.fa ? 4 : 3, b.smix_next, b.s0_child);
                    if (k + 1 < P.K && b.fa) { crow = lean_row_pack(lean_child(q, b), lean_row_rid(cur.row), lossy); at = seg0 + (int64_t)wave_alloc_lds(&s_cnt[2]); } // at most one child per ray: the segment cannot overflow
                }
            }
            // The wave's memory counter counts loads and stores together, in order.  The next row's loads are older than
            // this child's stores only if the stores come behind the point that waits for them; placed in front, every
            // iteration would wait for its own stores to be acknowledged.  So: wait here (the loads have had the whole
            // step to land; the previous child's stores are older still), then store.
            PP_ADD(3)
            __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0), as an instruction the compiler's own wait-count pass sees (inline asm is opaque to it) */
            PP_ADD(4)
            if (at >= 0) row_store(out_rows, at, crow);
            if (lossy) *a.overflow = 3;
        }
    } else
    for (int64_t base = seg0 + (threadIdx.x & ~63);; base += BLOCK) {
        if constexpr (PASS == 2) { // the rough-surface loop takes a different number of rounds in every wave: tiles are handed out as waves come free
            uint32_t tile = 0;
            if (lane_id() == 0) tile = atomicAdd(&s_cnt[6], 1u);
            base = seg0 + 64 * (int64_t)uni32(tile);
        }
        if (base >= n) break;
        int64_t idx = base + lane_id();
        // PASS 2: the wave works together on the rough-surface loop (unified_wave), so all 64 lanes go through the step;
        // those behind the end of the list repeat its last ray and leave no trace
        bool act = idx < n;
        constexpr bool WAVE = PASS == 2 && CF == FLAG_SINGLE;
        if constexpr (PASS == 2) { // place in the list; lanes behind the end of either part stand on its last ray
            const int64_t j = idx - seg0;
            if (base - seg0 < nfp) { act = j < nf; idx = seg0 + (act ? j : nf - 1); }
            else { const int64_t jb = j - nfp; act = jb < nb; idx = seg0 + a.seg_cap - 1 - (act ? jb : nb - 1); }
        }
        if (act || WAVE) {
            if (listed) idx = seg0 + (int64_t)list[idx];
            const int so = a.hit_so[idx];
            RayState q; uint32_t my_rid = 0;
            if constexpr (CF == FLAG_SINGLE) { LeanRow mine; row_load((const d2*)a.in_d, idx, mine); lean_row_unpack(mine, a, q); my_rid = lean_row_rid(mine); }
            else load_queue_ray(a, idx, q);
            q.k = k;
            q.r.d = normalized_if(q.r.d);
            Hit h; h.l = a.hit_l[idx];
            h.surf = so < 0 ? -1 : (so >> 2); h.orient = so < 0 ? 0 : (so & 3) - 1;
            if (act) atomicAdd(&s_cnt[0], 1u);
            if constexpr (CF == FLAG_SINGLE) { // tallies only: one child, built after the dice
                static_assert(CF != FLAG_SINGLE || PASS == 2, "the tally-only step of the other passes is the pipelined loop above");
                LeanBounce b;
                bounce_phys_single<1>(G, P, q, h, b, act);
                bool lossy = false;
                if (!act) {}
                else if (!b.scat) tally_add(t, s_tab, 0, 2, q.smix_last, b.inc_intensity); // escaped: RayTracer2.py:485-486
                else {
                    atomicAdd(&s_cnt[1], 1u);
                    if (b.surf_sc) tally_add(t, s_tab, 0, 0, b.smix_next, b.surf_abs);
                    tally_add(t, s_tab, 0, 1, b.smix_next, b.bulk_abs);
                    tally_add(t, s_tab, 0, b.fa ? 4 : 3, b.smix_next, b.s0_child);
                    if (k + 1 < P.K && b.fa) // the child behind those of pass 1 (at most one child per ray: the segment cannot overflow)
                        row_store((d2*)a.out.d, seg0 + (int64_t)wave_alloc_lds(&s_cnt[2]), lean_row_pack(lean_child(q, b), my_rid, lossy));
                }
                if (lossy) *a.overflow = 3;
            } else {
                Bounce b; b.scat = 0;
                bounce_phys<BF, PASS == 2 ? 1 : ((PASS == 1 || PASS == 3) ? 2 : 0)>(G, P, q, h, b);
                if (!b.scat) tally_add(t, s_tab, 0, 2, q.smix_last, b.inc_intensity); // escaped: RayTracer2.py:485-486
                else {
                    atomicAdd(&s_cnt[1], 1u);
                    if (b.surf_sc) tally_add(t, s_tab, 0, 0, b.smix_next, b.surf_abs);
                    tally_add(t, s_tab, 0, 1, b.smix_next, b.bulk_abs);
                    tally_add(t, s_tab, 0, 3, b.smix_next, (b.fa ? 0.0 : b.refr.s0) + (b.fb ? 0.0 : b.refl.s0));
                    tally_add(t, s_tab, 0, 4, b.smix_next, (b.fa ? b.refr.s0 : 0.0) + (b.fb ? b.refl.s0 : 0.0));
                    int64_t myslot = -1;
                    if (full) {
                        unsigned long long slot = wave_alloc(a.rec_count);
                        if (slot < (unsigned long long)a.rec_cap) {
                            fill_record(q, b, P, a.rec + slot, (size_t)a.rec_cap);
                            a.rec_step[slot] = k; a.rec_ray[slot] = q.ray_id; a.rec_parent[slot] = q.parent; a.rec_flags[slot] = q.flags;
                            myslot = (int64_t)slot;
                        } else *a.overflow = 1;
                    }
                    if (wantfin && (q.flags & RF_PRIMARY)) fill_final(q, b, P, a.fin + fin_slot(a, q.ray_id) * FIN);
                    if (k + 1 < P.K && (b.fa || b.fb)) { // single child: exactly one of fa, fb can be set
                        if constexpr (BF == 0) {
                            if (b.fa && b.fb) { // split (python/RayTracer2.py:531-540): the reflected child as well
                                RayState c = child_of(q, b, true, true);
                                c.parent = myslot;
                                const uint32_t at = wave_alloc_lds(&s_cnt[2]);
                                if ((int64_t)at < a.seg_cap) store_ray(a.out, seg0 + (int64_t)at, c); else *a.overflow = 2;
                            }
                        }
                        RayState c = child_of(q, b, !b.fa, b.fa != 0);
                        c.parent = myslot;
                        const uint32_t at = wave_alloc_lds(&s_cnt[2]);
                        if (BF != 0 || (int64_t)at < a.seg_cap) store_ray(a.out, seg0 + (int64_t)at, c); else *a.overflow = 2;
                    }
                }
            }
        }
    }
    PP_ADD(3)
    __syncthreads();
    PP_ADD(5)
    double* row = a.table + (size_t)k * ntab;
    for (int i = threadIdx.x; i < ntab; i += BLOCK) { double v = s_tab[i]; if (v != 0.0) atomicAdd(&row[i], v); }
    if (threadIdx.x == 0) {
        if (s_cnt[0]) atomicAdd(&a.n_live[k], (unsigned long long)s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&a.n_scat[k], (unsigned long long)s_cnt[1]);
        const uint32_t kept = (int64_t)s_cnt[2] < a.seg_cap ? s_cnt[2] : (uint32_t)a.seg_cap; // beyond that: overflow, the run is repeated
        a.seg_out[blockIdx.x] = kept;
        if (kept > child0) atomicAdd(a.tail, (unsigned long long)(kept - child0));
        if constexpr (PASS == 1) { a.defer_n[2 * blockIdx.x] = s_cnt[3]; a.defer_n[2 * blockIdx.x + 1] = s_cnt[5]; }
    }
#ifdef RT_PROFILE
    PP_ADD(6)
    if (lane_id() == 0 && a.prof) for (int i = 0; i < 7; i++) atomicAdd(&a.prof[(PASS == 2 ? 8 : 0) + i], (unsigned long long)pp[i]);
#endif
}

__global__ void k_aos_to_soa(const double* start, const double* rays, int64_t n, double* soa) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int f = 0; f < 3; f++) soa[f * n + i] = start[i * 3 + f];
    for (int f = 0; f < 10; f++) soa[(3 + f) * n + i] = rays[i * 10 + f];
}
// SoA records [REC][cap] -> rows [n][REC]
__global__ void k_rec_to_rows(const double* rec, int64_t cap, int64_t n, double* rows) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * REC) return;
    int64_t i = t / REC; int f = (int)(t % REC);
    rows[t] = rec[(int64_t)f * cap + i];
}
// one chunk's records as the record sink sees them, in one block of memory: rows [n][REC] | ray [n] | parent [n] | step [n] | flags [n]
__global__ void k_rec_pack(const double* rec, int64_t cap, int64_t n, const int32_t* step, const int32_t* flags, const int64_t* ray, const int64_t* parent, char* out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n * REC) { int64_t i = t / REC; int f = (int)(t % REC); ((double*)out)[t] = rec[(int64_t)f * cap + i]; }
    if (t < n) {
        int64_t* o_ray = (int64_t*)(out + (size_t)n * REC * 8); int64_t* o_par = o_ray + n; int32_t* o_step = (int32_t*)(o_par + n); int32_t* o_fl = o_step + n;
        o_ray[t] = ray[t]; o_par[t] = parent[t]; o_step[t] = step[t]; o_fl[t] = flags[t];
    }
}
__global__ void k_gen_point(KArgs a, int64_t n, double* start, double* rays) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 p; Ray10 r; gen_point_ray(a, a.ray_id_base + i, p, r);
    start[i * 3] = p.x; start[i * 3 + 1] = p.y; start[i * 3 + 2] = p.z;
    put_ray10(rays + i * 10, 1, 0, r);
}

__global__ void k_gen_camera(const double* cam, int64_t first, int64_t n, double* start, double* rays, int32_t* pixels) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 p; Ray10 r; gen_camera_ray(cam, first + i, p, r);
    start[i * 3] = p.x; start[i * 3 + 1] = p.y; start[i * 3 + 2] = p.z;
    put_ray10(rays + i * 10, 1, 0, r);
    const int64_t res0 = (int64_t)cam[0];
    pixels[2 * i] = (int32_t)((first + i) % res0 + 1); pixels[2 * i + 1] = (int32_t)((first + i) / res0 + 1);
}

} // namespace

// ================================================================================================ host side
// host array that grows without being filled (std::vector::resize would write 384 B per record just to have them overwritten)
struct RawBuf {
    double* p = nullptr; size_t n = 0, cap = 0;
    ~RawBuf() { free(p); }
    bool resize(size_t m) {
        if (m > cap) { size_t nc = std::max(m, cap + cap / 2); double* q = (double*)realloc(p, nc * sizeof(double)); if (!q) return false; p = q; cap = nc; }
        n = m; return true;
    }
    void clear() { n = 0; }
    double* data() { return p; }
    size_t size() const { return n; }
};
static void parallel_copy(void* dst, const void* src, size_t bytes) {
    const int nt = (int)std::min<size_t>(std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())), bytes >> 24);
    if (nt <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th; th.reserve((size_t)nt); const size_t per = ((bytes + nt - 1) / nt + 63) & ~(size_t)63;
    auto piece = [=](int t) { const size_t o = std::min(bytes, (size_t)t * per), e = std::min(bytes, o + per); if (e > o) memcpy((char*)dst + o, (const char*)src + o, e - o); };
    int started = 0;
    for (int t = 1; t < nt; t++) { try { th.emplace_back(piece, t); started = t; } catch (...) { break; } } // (a thread that cannot be started: its piece is copied here)
    piece(0);
    for (int t = started + 1; t < nt; t++) piece(t);
    for (auto& t : th) t.join();
}

struct rt_ctx {
    int device = 0; hipStream_t stream = nullptr; int n_cu = 0; size_t hbm = 0; std::string name;
    std::string err;
    // geometry
    double* d_surf = nullptr; double* d_recs = nullptr; uint32_t* d_smeta = nullptr; double* d_lut = nullptr;
    double* d_cam = nullptr; // camera block of the last rt_trace_camera / rt_generate_camera_rays
    int64_t cam_first = 0;   // rt_trace_camera_range: pixel of the call's first ray
    double* d_prog = nullptr; int nprog_frames = 0, nprog_generic = 0; // frame program, then the all-generic program
    bool use_frames = true, frames_have_generic = false;
    bool has_rayleigh = false, has_unified = false; bool defer_unified = true;
    int S = 0, n_atoms = 0, n_code = 0, lut_n = 0;
    // persistent small buffers
    unsigned long long* d_counters = nullptr; // head, tail, rec_count + overflow(int)
    double* d_table = nullptr; size_t table_elems = 0; unsigned long long* d_steps = nullptr; int steps_K = 0;
    double* d_table_ext = nullptr; unsigned long long* d_steps_ext = nullptr; // a lane of run_trace_lanes tallies into its parent's table and counters (cleared by the parent)
    // results of the last trace
    int K = 0; int num_steps = 0; int64_t n_rays = 0;
    std::vector<double> table; std::vector<int64_t> n_live, n_scat;
    RawBuf rec; std::vector<int32_t> rec_step, rec_flags; std::vector<int64_t> rec_ray, rec_parent;
    std::vector<double> fin;
    double* d_fin_user = nullptr; int64_t fin_user_rows = 0; // rt_set_final_rows_device: final rows stay in the caller's device buffer
    std::vector<std::pair<void*, size_t>> pool; // cached device buffers (see DevBuf)
    rt_record_sink sink = nullptr; void* sink_user = nullptr; // records leave chunk by chunk instead of accumulating
    // record streaming: two pinned host stages, a copy stream; the D2H copy of chunk c runs under the sink call of chunk c-1 and the kernels of chunk c+1
    hipStream_t copy_stream = nullptr; hipEvent_t ev_ready[2] = { nullptr, nullptr }, ev_copied[2] = { nullptr, nullptr };
    char* pin[2] = { nullptr, nullptr }; size_t pin_bytes[2] = { 0, 0 };
    // host inputs of large runs: pieces of upload_chunk rays are staged in two pinned buffers by host threads, one being filled
    // while the other is copied to the device
    char* in_pin[2] = { nullptr, nullptr }; size_t in_pin_bytes[2] = { 0, 0 }; hipEvent_t ev_up[2] = { nullptr, nullptr };
    int64_t upload_chunk = (int64_t)1 << 21;
    bool stream_overlap = true; size_t stream_stage_bytes = (size_t)2 << 30, accumulate_stage_bytes = (size_t)256 << 20;
    double kernel_ms = 0; int n_launches = 0;
    hipEvent_t lane_ev[4][8] = {};                                 // the lanes' turn events (run_trace_lanes), made on first use
    hipEvent_t evb[3 * 16] = {};                                   // per-step timing events of a batch of level steps (run_trace)
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };    // timing events of run_trace / rt_match_endpoints: created once, so that no error path can leak one
    double ms_kind[3] = { 0, 0, 0 }; int n_kind[3] = { 0, 0, 0 }; // k_trace, k_level_hit, k_level_phys
    unsigned long long* d_lvl = nullptr; int lvl_K = 0;           // live rays per step of the level-synchronous path
    // options
    int n_pass_retries = 0; // ray-splitting passes of the last trace that overflowed and were traced again in smaller ones (rt_result_pass_retries)
    size_t workspace_bytes = (size_t)32 << 30; int64_t nodice_chunk = 1 << 20; int refill_min = 4; int blocks_per_cu = 0; size_t lds_budget = 32 * 1024;
    int level_steps = 1 << 20; int64_t level_min_rays = -1; /* below that many live rays the persistent kernel takes over; -1: by run mode, see level_min_rays_of() */ int hit_blocks_per_cu = 0, phys_blocks_per_cu = 0, hit_split = 128, hit_split_lanes = 12, seg_per_cu = 8; bool seg_auto = true; int sync_every = 8; int64_t sync_rays_max = (int64_t)1 << 24;
    int match_max_level = 10; // rt_match_endpoints: deepest octree level (0: one uniform grid of cell size max_dist)
    // further lanes of large tally-only runs (run_trace_lanes; two by default): contexts of their own that borrow this one's
    // geometry.  The launches of the lanes overlap - which is what they are for - so per-kernel times then include the company
    int lanes = 2; int64_t lanes_min_rays = (int64_t)1 << 21; rt_ctx* lane[3] = { nullptr, nullptr, nullptr }; bool is_lane = false, raw_tally = false;
    // the lanes take turns with the nearest-hit launch (run_trace_lanes): what each has recorded so far, and the events
    struct LanePair { int L; std::atomic<int> rec[4]; std::atomic<int> done[4]; hipEvent_t ev[4][8]; }* pair = nullptr;
    int pair_id = 0, pair_seq = 0, lanes_alternate = 1;
};

// ---- rt_trace_device_noalloc: its queues, hits, lists and counters live here instead of in hipMalloc'ed memory.
// One copy per GPU the library runs on, held until the process ends.  The name below is a host-side stand-in, not a
// device address: host code uses g_scratch_ptr[device] (hipGetSymbolAddress, taken in rt_init with that device
// current), kernels get that pointer as an argument.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
constexpr int MAX_BATCH = 16; // level steps launched between two host synchronisations (rt_ctx::evb)
static unsigned char* g_scratch_ptr[MAX_DEVICES];
static std::mutex g_scratch_mutex;

// rt_init has no context to leave its message in: rt_last_error(NULL) returns the message of the calling thread's last
// failed rt_init (one context per (GPU, host thread) is the header's contract, so threads may call rt_init side by side)
static thread_local std::string g_init_err;
#define CK(call)                                                                                                           \
    do {                                                                                                                   \
        hipError_t e_ = (call);                                                                                            \
        if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return -1; }                \
    } while (0)

extern "C" int rt_init(int device, rt_ctx** out) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_init_err = "rt_init: no HIP device available (this library has no CPU path)"; return -1; }
    if (device < 0 || device >= n) { g_init_err = "rt_init: bad device index"; return -1; }
    rt_ctx* ctx = new rt_ctx; ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) { g_init_err = "rt_init: cannot create stream"; delete ctx; return -1; }
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, device);
    ctx->n_cu = pr.multiProcessorCount; ctx->hbm = pr.totalGlobalMem; ctx->name = pr.gcnArchName;
    if (hipMalloc(&ctx->d_counters, 256) != hipSuccess) { g_init_err = "rt_init: hipMalloc failed"; delete ctx; return -1; }
    for (int i = 0; i < 4; i++) if (hipEventCreate(&ctx->ev[i]) != hipSuccess) { g_init_err = "rt_init: cannot create events"; rt_free(ctx); return -1; }
    for (int i = 0; i < 3 * MAX_BATCH; i++) if (hipEventCreate(&ctx->evb[i]) != hipSuccess) { g_init_err = "rt_init: cannot create events"; rt_free(ctx); return -1; }
    if (device < MAX_DEVICES) { // this GPU's copy of `scratch` (each GPU holds its own, at its own address)
        std::lock_guard<std::mutex> lk(g_scratch_mutex);
        if (!g_scratch_ptr[device]) {
            void* sp = nullptr;
            hipError_t se = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
            if (se != hipSuccess || !sp) { g_init_err = std::string("rt_init: hipGetSymbolAddress(scratch): ") + hipGetErrorString(se); rt_free(ctx); return -1; }
            g_scratch_ptr[device] = (unsigned char*)sp;
        }
    }
    const char* ws = getenv("RT_WORKSPACE_GB"); if (ws && atof(ws) > 0) ctx->workspace_bytes = (size_t)(atof(ws) * (double)(1ull << 30));
    const char* rf = getenv("RT_REFILL_MIN"); if (rf && atoi(rf) > 0) ctx->refill_min = atoi(rf);
    const char* bc = getenv("RT_BLOCKS_PER_CU"); if (bc && atoi(bc) > 0) ctx->blocks_per_cu = atoi(bc);
    const char* nc = getenv("RT_NODICE_CHUNK"); if (nc && atoll(nc) > 0) ctx->nodice_chunk = atoll(nc);
    const char* nf = getenv("RT_NO_FRAMES"); if (nf && atoi(nf) > 0) ctx->use_frames = false;
    const char* ls = getenv("RT_LEVEL_STEPS"); if (ls && atoi(ls) >= 0) ctx->level_steps = atoi(ls);
    const char* lm = getenv("RT_LEVEL_MIN_RAYS"); if (lm) ctx->level_min_rays = atoll(lm) >= 0 ? atoll(lm) : -1;
    const char* hs = getenv("RT_HIT_SPLIT"); if (hs && atoi(hs) > 0) ctx->hit_split = atoi(hs);
    const char* ln = getenv("RT_LANES"); if (ln && atoi(ln) >= 1 && atoi(ln) <= 4) ctx->lanes = atoi(ln);
    const char* la = getenv("RT_LANES_ALTERNATE"); if (la) ctx->lanes_alternate = atoi(la) != 0;
    const char* lmr = getenv("RT_LANES_MIN_RAYS"); if (lmr && atoll(lmr) >= 0) ctx->lanes_min_rays = atoll(lmr);
    const char* sm = getenv("RT_SYNC_RAYS_MAX"); if (sm && atoll(sm) > 0) ctx->sync_rays_max = atoll(sm);
    const char* sc = getenv("RT_SEG_PER_CU"); if (sc && atoi(sc) > 0 && atoi(sc) <= 64) { ctx->seg_per_cu = atoi(sc); ctx->seg_auto = false; }
    const char* hb = getenv("RT_HIT_BLOCKS_PER_CU"); if (hb && atoi(hb) >= 0) ctx->hit_blocks_per_cu = atoi(hb);
    const char* pb = getenv("RT_PHYS_BLOCKS_PER_CU"); if (pb && atoi(pb) >= 0) ctx->phys_blocks_per_cu = atoi(pb);
    const char* so = getenv("RT_STREAM_OVERLAP"); if (so) ctx->stream_overlap = atoi(so) != 0;
    const char* sg = getenv("RT_STREAM_STAGE_GB"); if (sg && atof(sg) > 0) ctx->stream_stage_bytes = (size_t)(atof(sg) * (double)(1ull << 30));
    *out = ctx; return 0;
}

extern "C" void rt_free(rt_ctx* ctx) {
    if (!ctx) return;
    for (auto& l : ctx->lane) if (l) { rt_free(l); l = nullptr; } // (their geometry pointers are null between calls: they are this context's)
    hipSetDevice(ctx->device);
    hipFree(ctx->d_surf); hipFree(ctx->d_recs); hipFree(ctx->d_smeta); hipFree(ctx->d_prog); hipFree(ctx->d_cam); hipFree(ctx->d_lut); hipFree(ctx->d_counters); hipFree(ctx->d_table); hipFree(ctx->d_steps); hipFree(ctx->d_lvl);
    for (auto& pb : ctx->pool) hipFree(pb.first);
    for (int s = 0; s < 2; s++) { if (ctx->pin[s]) hipHostFree(ctx->pin[s]); if (ctx->ev_ready[s]) hipEventDestroy(ctx->ev_ready[s]); if (ctx->ev_copied[s]) hipEventDestroy(ctx->ev_copied[s]); }
    for (int s = 0; s < 2; s++) { if (ctx->in_pin[s]) hipHostFree(ctx->in_pin[s]); if (ctx->ev_up[s]) hipEventDestroy(ctx->ev_up[s]); }
    for (int i = 0; i < 4; i++) if (ctx->ev[i]) hipEventDestroy(ctx->ev[i]);
    for (int i = 0; i < 3 * MAX_BATCH; i++) if (ctx->evb[i]) hipEventDestroy(ctx->evb[i]);
    for (auto& row : ctx->lane_ev) for (auto& e : row) if (e) { hipEventDestroy(e); e = nullptr; }
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char* rt_last_error(rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_init_err.c_str(); }

extern "C" int rt_device_info(rt_ctx* ctx, int32_t* n_cu, int64_t* hbm_bytes, char* name, int32_t name_len) {
    if (n_cu) *n_cu = ctx->n_cu; if (hbm_bytes) *hbm_bytes = (int64_t)ctx->hbm;
    if (name && name_len > 0) { strncpy(name, ctx->name.c_str(), (size_t)name_len - 1); name[name_len - 1] = 0; }
    return 0;
}

extern "C" int rt_set_option(rt_ctx* ctx, const char* key, double value) {
    std::string k(key);
    if (k == "workspace_gb") ctx->workspace_bytes = (size_t)(value * (double)(1ull << 30));
    else if (k == "refill_min") ctx->refill_min = (int)value;
    else if (k == "blocks_per_cu") ctx->blocks_per_cu = (int)value;
    else if (k == "nodice_chunk") ctx->nodice_chunk = (int64_t)value;
    else if (k == "lds_budget_kb") ctx->lds_budget = (size_t)(value * 1024);
    else if (k == "release_workspace") {
        for (rt_ctx* l : ctx->lane) if (l) rt_set_option(l, "release_workspace", 0);
        hipSetDevice(ctx->device); for (auto& pb : ctx->pool) hipFree(pb.first); ctx->pool.clear();
        for (int s = 0; s < 2; s++) { if (ctx->pin[s]) hipHostFree(ctx->pin[s]); ctx->pin[s] = nullptr; ctx->pin_bytes[s] = 0; }
        for (int s = 0; s < 2; s++) { if (ctx->in_pin[s]) hipHostFree(ctx->in_pin[s]); ctx->in_pin[s] = nullptr; ctx->in_pin_bytes[s] = 0; }
    }
    else if (k == "upload_chunk") ctx->upload_chunk = (int64_t)value; // rays per piece; 0: one hipMemcpy from pageable memory
    else if (k == "stream_overlap") ctx->stream_overlap = value != 0;
    else if (k == "accumulate_stage_gb") { if (!(value > 0)) { ctx->err = "rt_set_option: accumulate_stage_gb must be positive"; return -1; } ctx->accumulate_stage_bytes = (size_t)(value * (double)(1ull << 30)); }
    else if (k == "stream_stage_gb") { if (!(value > 0)) { ctx->err = "rt_set_option: stream_stage_gb must be positive"; return -1; } ctx->stream_stage_bytes = (size_t)(value * (double)(1ull << 30)); }
    else if (k == "frames") ctx->use_frames = value != 0;
    else if (k == "level_steps") ctx->level_steps = (int)value;
    else if (k == "level_min_rays") ctx->level_min_rays = value >= 0 ? (int64_t)value : -1; // (negative: back to the defaults by run mode)
    else if (k == "hit_blocks_per_cu") ctx->hit_blocks_per_cu = (int)value;
    else if (k == "hit_split") ctx->hit_split = (int)value;
    else if (k == "hit_split_lanes") ctx->hit_split_lanes = std::max(1, (int)value);
    else if (k == "seg_per_cu") { if (!(value >= 1 && value <= 64)) { ctx->err = "rt_set_option: seg_per_cu must be in 1..64"; return -1; } ctx->seg_per_cu = (int)value; ctx->seg_auto = false; }
    else if (k == "phys_blocks_per_cu") ctx->phys_blocks_per_cu = (int)value;
    else if (k == "defer_unified") ctx->defer_unified = value != 0;
    else if (k == "lanes") { if (!(value >= 1 && value <= 4)) { ctx->err = "rt_set_option: lanes must be 1 to 4"; return -1; } ctx->lanes = (int)value; }
    else if (k == "lanes_min_rays") ctx->lanes_min_rays = (int64_t)value;
    else if (k == "lanes_alternate") ctx->lanes_alternate = value != 0;
    else if (k == "sync_every") { if (!(value >= 1 && value <= MAX_BATCH)) { ctx->err = "rt_set_option: sync_every must be in 1..16"; return -1; } ctx->sync_every = (int)value; }
    else if (k == "sync_rays_max") ctx->sync_rays_max = (int64_t)value;
    else if (k == "match_max_level") ctx->match_max_level = std::min(10, std::max(0, (int)value));
    else { ctx->err = "rt_set_option: unknown key " + k; return -1; }
    return 0;
}

extern "C" int rt_set_surfaces(rt_ctx* ctx, const double* table, int32_t n_surf, const double* atoms, int32_t n_atoms, const uint32_t* code, int32_t n_code) {
    if (n_surf <= 0 || n_surf > RT_MAX_SURF) { ctx->err = "rt_set_surfaces: need 1..256 surfaces"; return -1; }
    if (n_atoms < 0 || n_code < 0) { ctx->err = "rt_set_surfaces: negative size"; return -1; }
    // validate what the kernel will index with
    for (int s = 0; s < n_surf; s++) {
        const double* r = table + (size_t)s * ROW;
        int shape = (int)r[0], off = (int)r[27], len = (int)r[28];
        if (shape < 0 || shape > 4) { ctx->err = "Geometry has unrecognized shape"; return -1; }
        if (off < 0 || len < 0 || off + len > n_code) { ctx->err = "rt_set_surfaces: bounds program out of range"; return -1; }
        int depth = 0;
        for (int i = off; i < off + len; i++) {
            uint32_t op = code[i] >> 24, arg = code[i] & 0xFFFFFFu;
            if (op == 1) { if ((int)arg >= n_atoms) { ctx->err = "rt_set_surfaces: atom index out of range"; return -1; } depth++; }
            else if (op == 5 || op == 6) depth++;
            else if (op == 2 || op == 3 || op == 7) depth--;
            else if (op != 4) { ctx->err = "rt_set_surfaces: bad opcode"; return -1; }
            if (depth < 1 || depth > 32) { ctx->err = "rt_set_surfaces: bounds program stack out of range"; return -1; }
        }
    }
    for (int i = 0; i < n_atoms; i++) {
        int kind = (int)atoms[(size_t)i * ATOM], cmp = (int)atoms[(size_t)i * ATOM + 1];
        if (kind < 0 || kind > 3 || cmp < 0 || cmp > 5) { ctx->err = "rt_set_surfaces: bad atom"; return -1; }
        if (kind == 3) { int ax = (int)atoms[(size_t)i * ATOM + 3]; if (ax < 0 || ax > 2) { ctx->err = "rt_set_surfaces: bad coordinate atom"; return -1; } }
    }
    CK(hipSetDevice(ctx->device));
    hipFree(ctx->d_surf); hipFree(ctx->d_recs); hipFree(ctx->d_smeta); hipFree(ctx->d_prog);
    ctx->d_surf = ctx->d_recs = ctx->d_prog = nullptr; ctx->d_smeta = nullptr;
    std::vector<double> recs, prog, prog_generic; std::vector<uint32_t> smeta;
    build_device_tables(table, n_surf, atoms, n_atoms, code, recs, smeta);
    build_program(table, n_surf, atoms, code, true, recs, smeta, prog);
    build_program(table, n_surf, atoms, code, false, recs, smeta, prog_generic);
    finish_device_tables(recs);
    ctx->frames_have_generic = false;
    for (size_t i = 0; i + PREC < prog.size(); i += PREC) { uint64_t m; memcpy(&m, &prog[i], 8); if ((m & 15u) == PK_GENERIC) ctx->frames_have_generic = true; }
    ctx->nprog_frames = (int)(prog.size() / PREC) - 1; ctx->nprog_generic = (int)(prog_generic.size() / PREC) - 1;
    prog.insert(prog.end(), prog_generic.begin(), prog_generic.end());
    CK(hipMalloc(&ctx->d_prog, sizeof(double) * prog.size()));
    CK(hipMemcpy(ctx->d_prog, prog.data(), sizeof(double) * prog.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&ctx->d_surf, sizeof(double) * (size_t)n_surf * ROW));
    CK(hipMalloc(&ctx->d_recs, sizeof(double) * recs.size()));
    CK(hipMalloc(&ctx->d_smeta, 4 * smeta.size()));
    CK(hipMemcpy(ctx->d_surf, table, sizeof(double) * (size_t)n_surf * ROW, hipMemcpyHostToDevice));
    CK(hipMemcpy(ctx->d_recs, recs.data(), sizeof(double) * recs.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(ctx->d_smeta, smeta.data(), 4 * smeta.size(), hipMemcpyHostToDevice));
    ctx->has_rayleigh = false; ctx->has_unified = false;
    for (int s = 0; s < n_surf; s++) if ((int)table[(size_t)s * ROW + 1] == ST_UNIFIED) ctx->has_unified = true;
    for (int s = 0; s < n_surf; s++) if (table[(size_t)s * ROW + 20] < INFINITY || table[(size_t)s * ROW + 21] < INFINITY) ctx->has_rayleigh = true;
    ctx->S = n_surf; ctx->n_atoms = n_atoms; ctx->n_code = n_code;
    return 0;
}

extern "C" int rt_set_final_rows_device(rt_ctx* ctx, double* d_rows, int64_t n_rows) {
    if (d_rows && n_rows <= 0) { ctx->err = "rt_set_final_rows_device: no rows"; return -1; }
    ctx->d_fin_user = d_rows; ctx->fin_user_rows = d_rows ? n_rows : 0; return 0;
}
extern "C" int rt_set_record_sink(rt_ctx* ctx, rt_record_sink sink, void* user) { ctx->sink = sink; ctx->sink_user = user; return 0; }

extern "C" int rt_set_rayleigh_lut(rt_ctx* ctx, const double* lut, int32_t n) {
    if (n < 2) { ctx->err = "rt_set_rayleigh_lut: need at least 2 entries"; return -1; }
    CK(hipSetDevice(ctx->device));
    hipFree(ctx->d_lut); ctx->d_lut = nullptr;
    CK(hipMalloc(&ctx->d_lut, sizeof(double) * (size_t)n));
    CK(hipMemcpy(ctx->d_lut, lut, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    ctx->lut_n = n; return 0;
}

namespace {

// Device buffers of a trace come from a per-context pool and go back to it: hipMalloc/hipFree of the multi-GB ray
// queues cost milliseconds per call.  The pool is emptied by rt_free() or rt_set_option("release_workspace", 1).
struct DevBuf {
    void* p = nullptr; size_t bytes = 0; rt_ctx* owner = nullptr;
    ~DevBuf() { if (!p) return; if (owner) owner->pool.push_back({ p, bytes }); else hipFree(p); }
};
hipError_t dev_alloc(rt_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 8;
    size_t best = (size_t)-1; int bi = -1;
    for (size_t i = 0; i < ctx->pool.size(); i++) if (ctx->pool[i].second >= bytes && ctx->pool[i].second < best) { best = ctx->pool[i].second; bi = (int)i; }
    if (bi >= 0 && best <= 2 * bytes + (1u << 20)) { b.p = ctx->pool[bi].first; b.bytes = best; b.owner = ctx; ctx->pool.erase(ctx->pool.begin() + bi); return hipSuccess; }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) { // make room: drop what is cached and try again
        for (auto& pb : ctx->pool) hipFree(pb.first);
        ctx->pool.clear(); (void)hipGetLastError();
        e = hipMalloc(&b.p, bytes);
    }
    if (e == hipSuccess) { b.bytes = bytes; b.owner = ctx; } else b.p = nullptr;
    return e;
}

int alloc_queue(rt_ctx* ctx, Queue& q, DevBuf& bd, DevBuf& bi, DevBuf& bm, int64_t cap) {
    q.cap = cap;
    CK(dev_alloc(ctx, bd, sizeof(double) * QD * (size_t)cap)); CK(dev_alloc(ctx, bi, sizeof(int64_t) * QI * (size_t)cap)); CK(dev_alloc(ctx, bm, sizeof(int32_t) * QM * (size_t)cap));
    q.d = (double*)bd.p; q.i = (int64_t*)bi.p; q.m = (int32_t*)bm.p;
    return 0;
}

// dynamic LDS of k_level_phys: surface rows | the step's tally row | 8 counters
static size_t lds_phys_bytes(int S) { return (size_t)S * ROW * 8 + (size_t)5 * S * 2 * 8 + 32; }

// path hash of every ray of step k when each interaction leaves one child (lean_child(): path_hash(h, k, 0) from h = 0)
static uint32_t lean_hash_of(int k) { uint32_t h = 0; for (int j = 0; j < k; j++) h = path_hash(h, j, 0); return h; }

// absorption_table [K][5][S][2] and the per-step counters [2][K] of a trace on the device: room for K steps, cleared
int prepare_tallies(rt_ctx* ctx, int K) {
    const size_t telems = (size_t)K * 5 * ctx->S * 2;
    if (ctx->table_elems < telems) { hipFree(ctx->d_table); ctx->d_table = nullptr; ctx->table_elems = 0; CK(hipMalloc(&ctx->d_table, telems * 8)); ctx->table_elems = telems; }
    if (ctx->steps_K < K) { hipFree(ctx->d_steps); ctx->d_steps = nullptr; ctx->steps_K = 0; CK(hipMalloc(&ctx->d_steps, sizeof(unsigned long long) * 2 * (size_t)K)); ctx->steps_K = K; }
    CK(hipMemsetAsync(ctx->d_table, 0, telems * 8, ctx->stream));
    CK(hipMemsetAsync(ctx->d_steps, 0, sizeof(unsigned long long) * 2 * (size_t)K, ctx->stream));
    return 0;
}
// ... to the host, with the reference's step bookkeeping (RayTracer2.py:266-270, 405-407, 542)
int fetch_tallies(rt_ctx* ctx, int K) {
    const int S = ctx->S; const size_t telems = (size_t)K * 5 * S * 2;
    ctx->table.assign(telems, 0.0); std::vector<unsigned long long> st(2 * (size_t)K);
    CK(hipMemcpyAsync(ctx->table.data(), ctx->d_table, telems * 8, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipMemcpyAsync(st.data(), ctx->d_steps, sizeof(unsigned long long) * 2 * (size_t)K, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    ctx->n_live.assign(K, 0); ctx->n_scat.assign(K, 0);
    int ns = 0;
    for (int k = 0; k < K; k++) { ctx->n_live[k] = (int64_t)st[k]; ctx->n_scat[k] = (int64_t)st[K + k]; if (st[k] > 0) ns = k + 1; }
    for (int k = 0; k < ns; k++)
        if (ctx->n_scat[k] == 0) std::fill(ctx->table.begin() + (size_t)k * 5 * S * 2, ctx->table.begin() + (size_t)(k + 1) * 5 * S * 2, 0.0); // `continue` before the tallies
    ctx->num_steps = ns;
    return 0;
}

// Live rays below which the persistent kernel is faster than the level-synchronous pair, by run mode (one lane, jar geometry,
// profiles/r3_level_threshold.txt): tally-only single-child runs have the lean persistent instances (2 waves per SIMD, whole
// register file): 786 432; single-child runs that keep final rows: 524 288 (long chains - the jar with reflective cones,
// K = 100 - cross there; the image pipeline, whose chains are three interactions long, is level or a little faster on the
// persistent kernel up to 500 k pixels, tools/bench_image.py); with
// records the host's share decides and the persistent kernel stays ahead at every size
// measured: 786 432; ray-splitting runs: 0 - the level path is 2 to 4.5 times faster from 256 rays up.
static int64_t level_min_rays_of(const rt_ctx* ctx, bool single, bool full, bool wantfin) {
    if (ctx->level_min_rays >= 0) return ctx->level_min_rays;
    if (!single) return 0;
    return (wantfin && !full) ? ((int64_t)1 << 19) : ((int64_t)3 << 18);
}
// mode 0: h_start/h_rays host arrays or d_soa device SoA; mode 2: point source
int run_trace(rt_ctx* ctx, int mode, const double* h_start, const double* h_rays, const double* d_soa, const double* origin, int64_t n, const rt_params* prm, int64_t soa_stride = 0) {
    if (!ctx->d_surf) { ctx->err = "rt_trace: no surfaces set"; return -1; }
    if (n <= 0) { ctx->err = "Improper input to RayTracer2"; return -1; }
    const int K = prm->max_scatters;
    if (K < 1 || K > 4094) { ctx->err = "rt_trace: max_scatters must be in 1..4094"; return -1; }
    if (prm->flags & ~(RT_FLAG_FULL_OUTPUT | RT_FLAG_SINGLECHILD | RT_FLAG_FINAL_ROWS)) { ctx->err = "rt_trace: unknown bits in rt_params.flags"; return -1; }
    if (prm->compat & ~(RT_COMPAT_PY_INDEX | RT_COMPAT_PY_ROOT2)) { ctx->err = "rt_trace: unknown bits in rt_params.compat"; return -1; }
    CK(hipSetDevice(ctx->device));
    const int S = ctx->S;
    const bool full = prm->flags & RT_FLAG_FULL_OUTPUT, single = prm->flags & RT_FLAG_SINGLECHILD, wantfin = prm->flags & RT_FLAG_FINAL_ROWS;
    const bool has_rayleigh = ctx->has_rayleigh;
    if (has_rayleigh && !ctx->d_lut) { ctx->err = "rt_trace: geometry has finite Rayleigh lengths but no azimuth table was set (rt_set_rayleigh_lut)"; return -1; }

    // ---- per-call device state
    size_t telems = (size_t)K * 5 * S * 2;
    if (!ctx->d_table_ext && prepare_tallies(ctx, K)) return -1; // (a lane: its parent has done this before the lanes started)
    double* const d_table = ctx->d_table_ext ? ctx->d_table_ext : ctx->d_table; unsigned long long* const d_steps = ctx->d_table_ext ? ctx->d_steps_ext : ctx->d_steps;
    ctx->K = K; ctx->n_rays = n; ctx->kernel_ms = 0; ctx->n_launches = 0; ctx->n_pass_retries = 0;
    for (int i = 0; i < 3; i++) { ctx->ms_kind[i] = 0; ctx->n_kind[i] = 0; }
    // level-synchronous path: single-child runs with at most one child per ray and step
    const bool level = ctx->level_steps > 0 && (!single || !(prm->follow_threshold[1] < 0));
    if (level && ctx->lvl_K < K) { hipFree(ctx->d_lvl); ctx->d_lvl = nullptr; CK(hipMalloc(&ctx->d_lvl, sizeof(unsigned long long) * ((size_t)K + 2))); ctx->lvl_K = K; }
    ctx->rec.clear(); ctx->rec_step.clear(); ctx->rec_flags.clear(); ctx->rec_ray.clear(); ctx->rec_parent.clear();
    const bool fin_dev = wantfin && ctx->d_fin_user != nullptr;
    if (fin_dev && n > ctx->fin_user_rows) { ctx->err = "rt_trace: the device buffer for the final rows is smaller than the number of rays"; return -1; }
    ctx->fin.clear(); if (wantfin && !fin_dev) ctx->fin.resize((size_t)n * FIN);

    // ---- LDS budget: surface rows + as many absorption_table rows as fit in lds_budget (the first steps carry almost
    // all interactions; deeper rows go to global atomics)
    size_t fixed = (size_t)(HDR_DOUBLES + TQ_DOUBLES) * 8 + (size_t)S * ROW * 8 + (size_t)2 * K * 4;
    const size_t lds_max = ctx->lds_budget;
    size_t rowb = (size_t)5 * S * 2 * 8;
    int rows_lds = fixed + rowb <= lds_max ? (int)std::min<size_t>((size_t)K, (lds_max - fixed) / rowb) : 0;
    size_t lds = fixed + (size_t)rows_lds * rowb;
    if (lds > 160 * 1024) { ctx->err = "rt_trace: geometry does not fit LDS"; return -1; }
    // kernel instance: exact on single-child; records/final rows compiled in only when one of them is requested
    const bool out = full || wantfin;
    void (*kern)(const KArgs) = single ? (out ? k_trace<FLAG_SINGLE | FLAG_FULL | FLAG_FINAL> : k_trace<FLAG_SINGLE>) : (out ? k_trace<FLAG_FULL | FLAG_FINAL> : k_trace<0>);
    const bool frames_only = ctx->use_frames && !(prm->compat & RT_COMPAT_PY_ROOT2) && !ctx->frames_have_generic;
    if (single && !out && !(prm->follow_threshold[1] < 0)) {
        if (ctx->has_unified) kern = frames_only ? k_trace<FLAG_SINGLE, true, false, 1> : k_trace<FLAG_SINGLE, true, true, 1>;
        else kern = frames_only ? k_trace<FLAG_SINGLE, true, false, 2> : k_trace<FLAG_SINGLE, true, true, 2>;
    }
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int bpc = ctx->blocks_per_cu;
    if (bpc <= 0) { int nb = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, BLOCK, lds) != hipSuccess || nb < 1) nb = 1; bpc = nb > 8 ? 8 : nb; }
    // ---- chunking
    const size_t W = ctx->workspace_bytes;
    const size_t rec_bytes = (size_t)REC * 8 + 4 + 8 + 8 + 4;
    int64_t chunk = n, rec_cap = 0, q_cap = 0;
    // records that stream to a sink: the chunk's records + two packed copies of them (one in flight to the host, one being filled), each at most one stage
    // Records that accumulate for rt_result_records take the same way in single-child runs (smaller stages: pinning memory
    // costs ~0.25 s per GiB once per context); ray-splitting runs cannot bound the records of a pass and keep one buffer.
    const bool piped = full && ctx->stream_overlap && (ctx->sink || single);
    const size_t Wrec = piped ? std::min(W / 3, ctx->sink ? ctx->stream_stage_bytes : ctx->accumulate_stage_bytes) : W;
    if (single) {
        if (full) { chunk = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)(Wrec / (rec_bytes * (size_t)K)))); rec_cap = chunk * K; }
        if (mode == 0 && !d_soa) chunk = std::min<int64_t>(chunk, std::max<int64_t>(1, (int64_t)(W / (13 * 8 * 2))));
        if (wantfin) chunk = std::min<int64_t>(chunk, std::max<int64_t>(1, (int64_t)(W / (FIN * 8))));
        if (prm->follow_threshold[1] < 0) q_cap = chunk * K; // degenerate: reflected child followed as well
        if (level) { // two ray queues (ping-pong) + the hit of every live ray
            const size_t per_ray = 2 * ((size_t)QD * 8 + QI * 8 + QM * 4) + 12 + 8; // + hit (12 B) + the two ray lists of the physics launches (4 B each)
            chunk = std::min<int64_t>(chunk, std::max<int64_t>(1, (int64_t)(W / per_ray)));
            q_cap = chunk + (int64_t)ctx->n_cu * ctx->seg_per_cu; // segments round the capacity up
        }
    } else {
        chunk = std::min<int64_t>(n, ctx->nodice_chunk);
        // every interaction above the thresholds spawns a reflected child: the live count can grow by 2 per step (13 live
        // rays per initial ray at the peak of the jar geometry with black reflectors, K = 18; far more with reflective
        // ones).  The queues take the whole workspace (at most 1024 rays per ray of the pass), so that a smaller pass has
        // more room per ray; an overflow is reported (-3) and the caller retries with smaller passes.
        const size_t per_q = 2 * ((size_t)QD * 8 + QI * 8 + QM * 4) + 12 + 8;
        q_cap = std::max<int64_t>(std::min<int64_t>(1024 * chunk, (int64_t)(W / per_q)), std::max<int64_t>(4 * chunk, 1 << 16));
        q_cap += (int64_t)ctx->n_cu * ctx->seg_per_cu; // segments of the level path round the capacity up
        if (full) rec_cap = std::max<int64_t>((int64_t)(Wrec / rec_bytes), 1 << 16), rec_cap = std::min<int64_t>(rec_cap, (int64_t)1 << 31);
    }
    // host inputs of more than two upload pieces come in through two pinned stages (see the chunk loop)
    const bool piped_in = mode == 0 && !d_soa && ctx->stream_overlap && ctx->upload_chunk > 0 && std::min(n, chunk) > 2 * ctx->upload_chunk;
    if (full && rec_cap < chunk) { rec_cap = chunk; }

    DevBuf b_hl, b_hso, b_seg, b_defer;
    const int nseg_max = ctx->n_cu * ctx->seg_per_cu;
    // geometry with UNIFIED surfaces: the rays that end on them get a physics launch of their own (k_level_phys PASS 1 + 2)
    const bool defer = level && ctx->has_unified && ctx->defer_unified; // (a launch of their own for bulk scatters alone did not pay: SBC light collection 6.1e9 -> 5.6e9)
    if (level) { CK(dev_alloc(ctx, b_hl, 8 * (size_t)q_cap)); CK(dev_alloc(ctx, b_hso, 4 * (size_t)q_cap)); CK(dev_alloc(ctx, b_seg, sizeof(uint32_t) * 4 * (size_t)nseg_max)); }
    if (defer) CK(dev_alloc(ctx, b_defer, 2 * 4 * (size_t)q_cap));
    DevBuf b_in, b_hs, b_hr, b_q0d, b_q0i, b_q0m, b_q1d, b_q1i, b_q1m, b_rec, b_rstep, b_rray, b_rpar, b_rflag, b_fin, b_rows;
    Queue q[2] = { { nullptr, nullptr, nullptr, 0 }, { nullptr, nullptr, nullptr, 0 } };
    if (q_cap > 0) { if (alloc_queue(ctx, q[0], b_q0d, b_q0i, b_q0m, q_cap)) return -1; if (alloc_queue(ctx, q[1], b_q1d, b_q1i, b_q1m, q_cap)) return -1; }
    if (mode == 0 && !d_soa) { CK(dev_alloc(ctx, b_in, 13 * 8 * (size_t)chunk)); CK(dev_alloc(ctx, b_hs, 3 * 8 * (size_t)chunk)); CK(dev_alloc(ctx, b_hr, 10 * 8 * (size_t)chunk)); }
    if (piped_in) for (int s = 0; s < 2; s++) {
        if (!ctx->ev_up[s]) CK(hipEventCreateWithFlags(&ctx->ev_up[s], hipEventDisableTiming));
        const size_t need = 13 * 8 * (size_t)ctx->upload_chunk;
        if (ctx->in_pin_bytes[s] < need) {
            if (ctx->in_pin[s]) { CK(hipHostFree(ctx->in_pin[s])); ctx->in_pin[s] = nullptr; ctx->in_pin_bytes[s] = 0; }
            CK(hipHostMalloc((void**)&ctx->in_pin[s], need, hipHostMallocDefault)); ctx->in_pin_bytes[s] = need;
        }
    }
    if (full) {
        CK(dev_alloc(ctx, b_rec, (size_t)REC * 8 * (size_t)rec_cap)); CK(dev_alloc(ctx, b_rstep, 4 * (size_t)rec_cap)); CK(dev_alloc(ctx, b_rray, 8 * (size_t)rec_cap));
        CK(dev_alloc(ctx, b_rpar, 8 * (size_t)rec_cap)); CK(dev_alloc(ctx, b_rflag, 4 * (size_t)rec_cap));
    }
    if (wantfin && !fin_dev) CK(dev_alloc(ctx, b_fin, (size_t)FIN * 8 * (size_t)chunk));
    struct Stage { DevBuf dev; int64_t nrec = 0, first = 0, m = 0; bool pending = false; } stg[2];
    // every way out of this function (the many early error returns included) waits for both streams before the DevBufs
    // above and the stages go back to the pool: a later call must not be handed memory a kernel or a copy still uses
    struct Quiesce { rt_ctx* c; ~Quiesce() { (void)hipStreamSynchronize(c->stream); if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream); } } quiesce{ ctx };
    if (piped) {
        const size_t stage_bytes = rec_bytes * (size_t)rec_cap;
        if (!ctx->copy_stream) CK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (int s = 0; s < 2; s++) {
            if (!ctx->ev_ready[s]) CK(hipEventCreateWithFlags(&ctx->ev_ready[s], hipEventDisableTiming));
            if (!ctx->ev_copied[s]) CK(hipEventCreateWithFlags(&ctx->ev_copied[s], hipEventDisableTiming));
            CK(dev_alloc(ctx, stg[s].dev, stage_bytes));
            if (ctx->pin_bytes[s] < stage_bytes) {
                if (ctx->pin[s]) { CK(hipHostFree(ctx->pin[s])); ctx->pin[s] = nullptr; ctx->pin_bytes[s] = 0; }
                CK(hipHostMalloc((void**)&ctx->pin[s], stage_bytes, hipHostMallocDefault)); ctx->pin_bytes[s] = stage_bytes;
            }
        }
    }
    // hands the records of the chunk in stage s to the sink, once their copy has landed
    auto drain = [&](int s) -> int {
        if (!stg[s].pending) return 0;
        stg[s].pending = false;
        if (hipEventSynchronize(ctx->ev_copied[s]) != hipSuccess) { ctx->err = "rt_trace: record copy failed"; return -1; }
        const int64_t nr = stg[s].nrec; char* h = ctx->pin[s];
        int64_t* h_ray = (int64_t*)(h + (size_t)nr * REC * 8); int64_t* h_par = h_ray + nr; int32_t* h_step = (int32_t*)(h_par + nr); int32_t* h_fl = h_step + nr;
        if (ctx->sink) {
            if (ctx->sink(ctx->sink_user, nr, (const double*)h, h_step, h_ray, h_par, h_fl, stg[s].first, stg[s].m) != 0) { ctx->err = "rt_trace: the record sink asked to stop"; return -4; }
            return 0;
        }
        const size_t base = ctx->rec_step.size();
        if (!ctx->rec.resize((base + (size_t)nr) * REC)) { ctx->err = "rt_trace: out of host memory for the records"; return -1; }
        parallel_copy(ctx->rec.data() + base * REC, h, (size_t)nr * REC * 8);
        ctx->rec_step.insert(ctx->rec_step.end(), h_step, h_step + nr); ctx->rec_flags.insert(ctx->rec_flags.end(), h_fl, h_fl + nr);
        ctx->rec_ray.insert(ctx->rec_ray.end(), h_ray, h_ray + nr);
        ctx->rec_parent.resize(base + (size_t)nr);
        for (int64_t i = 0; i < nr; i++) ctx->rec_parent[base + (size_t)i] = h_par[i] >= 0 ? h_par[i] + (int64_t)base : h_par[i]; // parents are chunk-local
        return 0;
    };

    KArgs a; memset(&a, 0, sizeof a);
    a.surf = ctx->d_surf; a.S = S; a.recs = ctx->d_recs; a.smeta = ctx->d_smeta;
    a.lut = ctx->d_lut; a.lut_prec = ctx->lut_n > 0 ? ctx->lut_n - 1 : 1;
    if (ctx->use_frames && !(prm->compat & RT_COMPAT_PY_ROOT2)) { a.prog = ctx->d_prog; a.nprog = ctx->nprog_frames; }
    else { a.prog = ctx->d_prog + (size_t)(ctx->nprog_frames + 1) * PREC; a.nprog = ctx->nprog_generic; }
    a.P.K = K; a.P.flags = prm->flags; a.P.compat = prm->compat; a.P.maxiter = prm->unified_maxiter > 0 ? prm->unified_maxiter : 10000;
    a.P.base = (prm->compat & RT_COMPAT_PY_INDEX) ? 0 : 1; a.P.min_travel = prm->min_travel_length; a.P.thrA = prm->follow_threshold[0]; a.P.thrB = prm->follow_threshold[1];
    a.P.tir = prm->tir_handling; a.P.key0 = (uint32_t)prm->seed; a.P.key1 = (uint32_t)(prm->seed >> 32);
    a.table = d_table; a.n_live = d_steps; a.n_scat = d_steps + K;
    a.prof = ctx->d_counters + 8; a.head = ctx->d_counters; a.tail = ctx->d_counters + 1; a.rec_count = ctx->d_counters + 2; a.overflow = (int*)(ctx->d_counters + 3);
    a.rec = (double*)b_rec.p; a.rec_cap = rec_cap; a.rec_step = (int32_t*)b_rstep.p; a.rec_ray = (int64_t*)b_rray.p; a.rec_parent = (int64_t*)b_rpar.p; a.rec_flags = (int32_t*)b_rflag.p;
    a.fin = (double*)b_fin.p; a.rows_lds = rows_lds; a.refill_min = ctx->refill_min;
    if (origin) { a.ox = origin[0]; a.oy = origin[1]; a.oz = origin[2]; }
    const int64_t stride = prm->ray_id_stride > 0 ? prm->ray_id_stride : 1;
    a.id_stride = stride;
    a.cam = ctx->d_cam; a.cam_base = prm->ray_id_base - ctx->cam_first; // mode 3: pixel = ray id - cam_base
    a.hit_l = (double*)b_hl.p; a.hit_so = (int32_t*)b_hso.p;
    // k_level_phys variants: 3 = geometry without UNIFIED surfaces; 1 + 2 = those rays in a launch of their own; 0 = one launch
    const bool nu = !ctx->has_unified;
    void (*kphys)(const KArgs); void (*kphys2)(const KArgs) = nullptr;
#define RT_PICK(CF_) { if (defer) { kphys = k_level_phys<CF_, 1>; kphys2 = k_level_phys<CF_, 2>; } else if (nu) kphys = k_level_phys<CF_, 3>; else kphys = k_level_phys<CF_, 0>; }
    if (!single) { if (out) RT_PICK(FLAG_FULL | FLAG_FINAL) else RT_PICK(0) }
    else { if (out) RT_PICK(FLAG_SINGLE | FLAG_FULL | FLAG_FINAL) else RT_PICK(FLAG_SINGLE) }
#undef RT_PICK
    a.defer = (uint32_t*)b_defer.p; a.dense = defer ? (uint32_t*)b_defer.p + (size_t)q_cap : nullptr; a.defer_n = level ? (uint32_t*)b_seg.p + 2 * (size_t)nseg_max : nullptr;
    void (*kinit)(const KArgs) = out ? k_level_init<FLAG_FULL | FLAG_FINAL> : k_level_init<0>;
    const bool prog_generic = !(ctx->use_frames && !(prm->compat & RT_COMPAT_PY_ROOT2)) || ctx->frames_have_generic;
    void (*khit)(const HitArgs) = prog_generic ? k_level_hit<true> : k_level_hit<false>;
    HitArgs ha; memset(&ha, 0, sizeof ha);
    ha.prog = a.prog; ha.nprog = a.nprog; ha.recs = ctx->d_recs; ha.surf = ctx->d_surf; ha.smeta = ctx->d_smeta; ha.S = S;
    ha.min_travel = a.P.min_travel; ha.base = a.P.base; ha.compat = a.P.compat; ha.hit_l = (double*)b_hl.p; ha.hit_so = (int32_t*)b_hso.p;
    ha.lean = a.lean = (single && !out) ? 1 : 0; // one decision for the writers (k_level_init, k_level_phys) and the readers (k_level_hit, k_level_phys, k_level_compact)
    const size_t lds_hit = (size_t)TQ_DOUBLES * 8, lds_phys = lds_phys_bytes(S);
    if (level) {
        if (lds_phys > 160 * 1024) { ctx->err = "rt_trace: geometry does not fit LDS"; return -1; }
        CK(hipFuncSetAttribute((const void*)kphys, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_phys));
        if (defer) CK(hipFuncSetAttribute((const void*)kphys2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_phys));
    }
    const int hit_bpc = ctx->hit_blocks_per_cu > 0 ? ctx->hit_blocks_per_cu : 4096, phys_bpc = ctx->phys_blocks_per_cu > 0 ? ctx->phys_blocks_per_cu : RT_PHYS_WAVES;

    hipEvent_t ev0 = ctx->ev[0], ev1 = ctx->ev[1], ev2 = ctx->ev[2];
    int rc = 0;
    auto do_pass = [&](const int64_t off, const int64_t m) -> int {
        int rc = 0;
        CK(hipMemsetAsync(ctx->d_counters, 0, 64, ctx->stream));
        a.mode = mode; a.n_in = m; a.ray_id_base = prm->ray_id_base + off * stride; a.slot_base = a.ray_id_base;
        if (fin_dev) a.fin = ctx->d_fin_user + (size_t)off * FIN;
        if (mode == 0) {
            if (d_soa) { a.in_d = d_soa + off; a.in_stride = soa_stride > 0 ? soa_stride : n; }
            else if (piped_in) {
                // the caller's pageable arrays go through two pinned stages, upload_chunk rays at a time: host threads fill one
                // stage while the other is on its way to the device (a plain hipMemcpy of pageable memory stages on one thread)
                int64_t piece = 0;
                for (int64_t o = 0; o < m && rc == 0; o += ctx->upload_chunk, piece++) {
                    const int s = (int)(piece & 1); const int64_t mm = std::min<int64_t>(ctx->upload_chunk, m - o);
                    if (piece >= 2) CK(hipEventSynchronize(ctx->ev_up[s]));
                    char* pin = ctx->in_pin[s];
                    parallel_copy(pin, h_start + (off + o) * 3, 24 * (size_t)mm); parallel_copy(pin + 24 * (size_t)mm, h_rays + (off + o) * 10, 80 * (size_t)mm);
                    CK(hipMemcpyAsync((char*)b_hs.p + 24 * (size_t)o, pin, 24 * (size_t)mm, hipMemcpyHostToDevice, ctx->stream));
                    CK(hipMemcpyAsync((char*)b_hr.p + 80 * (size_t)o, pin + 24 * (size_t)mm, 80 * (size_t)mm, hipMemcpyHostToDevice, ctx->stream));
                    CK(hipEventRecord(ctx->ev_up[s], ctx->stream));
                }
                hipLaunchKernelGGL(k_aos_to_soa, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)b_hs.p, (const double*)b_hr.p, m, (double*)b_in.p);
                a.in_d = (const double*)b_in.p; a.in_stride = m;
            } else {
                CK(hipMemcpyAsync(b_hs.p, h_start + off * 3, 3 * 8 * (size_t)m, hipMemcpyHostToDevice, ctx->stream));
                CK(hipMemcpyAsync(b_hr.p, h_rays + off * 10, 10 * 8 * (size_t)m, hipMemcpyHostToDevice, ctx->stream));
                hipLaunchKernelGGL(k_aos_to_soa, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)b_hs.p, (const double*)b_hr.p, m, (double*)b_in.p);
                a.in_d = (const double*)b_in.p; a.in_stride = m;
            }
        }
        int cur = 0;
        int64_t live = m; int k0 = 0;
        if (level) {
            // segments: nseg blocks, each owns seg_cap consecutive queue slots through all level steps
            // one block of the physics kernel per segment: enough of them to fill the machine once the rays have multiplied
            // Single-child runs up to 16M rays: two segments per CU, which is one resident round of the physics kernel (2 blocks per
            // CU at its register budget) - measured against 8 per CU: +24 % at 1M rays, +12 % at 4M, +3 % at 16M, -2 % at 33M
            const int spc = (single && ctx->seg_auto && (m <= ((int64_t)1 << 24) || ctx->pair)) ? std::min(2, ctx->seg_per_cu) : ctx->seg_per_cu; // (beside another lane two stay better above 16M too: +3-4 % at 50M and 67M rays)
            const int nseg = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->n_cu * spc, ((single ? m : std::min<int64_t>(q_cap, 64 * m)) + 511) / 512));
            const int64_t seg_fill = (m + nseg - 1) / nseg;
            const int64_t seg_cap = single ? seg_fill : std::max<int64_t>(seg_fill, (q_cap - nseg) / nseg); // rays that split: room to grow
            a.seg_fill = seg_fill;
            uint32_t* seg[2] = { (uint32_t*)b_seg.p, (uint32_t*)b_seg.p + nseg_max };
            CK(hipMemsetAsync(ctx->d_lvl, 0, sizeof(unsigned long long) * ((size_t)K + 2), ctx->stream));
            CK(hipMemsetAsync(b_seg.p, 0, sizeof(uint32_t) * 2 * (size_t)nseg_max, ctx->stream));
            a.seg_cap = seg_cap; ha.seg_cap = seg_cap;
            const int64_t lmin = level_min_rays_of(ctx, single, full, wantfin);
            const bool go = K > 0 && ctx->level_steps > 0 && live >= lmin;
            if (go) { // initial rays -> queue 1 (step 0 writes queue 0)
                a.out = q[1]; a.seg_out = seg[1];
                hipLaunchKernelGGL(kinit, dim3((unsigned)((live + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, a);
                CK(hipGetLastError());
                a.mode = 1; a.in_d = q[1].d; a.in_i = q[1].i; a.in_m = q[1].m; a.in_stride = q[1].cap;
            }
            int sin = 1; // seg[sin]: counts of the input queue
            while (k0 < K && k0 < ctx->level_steps && live > 0 && live >= lmin) {
                // The host learns the live count by waiting for the step (the kernels themselves read the counts of their
                // segments from device memory).  Runs whose steps take milliseconds wait after every step; on smaller ones
                // the round trip is a tenth to a third of the step (profiles/r2_sweep.json), so `sync_every` steps are
                // launched back to back on the grid of the first - rays that die only leave tiles without work - and the
                // counts are read once.  Only where rays cannot multiply (single child) and nothing per step leaves the device.
                const int batch = (single && !full && live < ctx->sync_rays_max) ? std::max(1, std::min(ctx->sync_every, MAX_BATCH)) : 1;
                const int nb = (int)std::min<int64_t>(batch, std::min<int64_t>(K, ctx->level_steps) - k0);
                // Blocks of the nearest-hit launch per segment: one per tile of 256 rays (up to hit_split) when the launch has the
                // GPU to itself.  Beside another lane's physics launches a dozen longer-lived blocks per segment do better
                // (+2-3 % from 16M rays: blocks that end every few microseconds keep taking the registers a physics block
                // - four waves of 199 - is waiting for); powers of two are slow (16: -2 %, 64: -6 %).
                const int64_t tiles = (live / nseg + BLOCK - 1) / BLOCK;
                const int split = (int)std::max<int64_t>(1, std::min<int64_t>((ctx->pair && tiles >= 32) ? std::min(ctx->hit_split, ctx->hit_split_lanes) : ctx->hit_split, tiles));
                for (int j = 0; j < nb; j++) {
                    const int k = k0 + j;
                    a.step = k; a.cnt_in = nullptr; a.n_in = live; a.out = q[cur]; a.tail = ctx->d_lvl + k + 1; a.lean_hash = lean_hash_of(k);
                    a.seg_in = seg[sin]; a.seg_out = seg[sin ^ 1];
                    ha.in_d = a.in_d; ha.in_m = a.in_m; ha.in_stride = a.in_stride; ha.n_in = live; ha.seg_in = seg[sin]; ha.split = split;
                    if (ctx->pair) { // round robin: lane m's i-th nearest-hit launch starts when lane m-1's i-th has ended, lane 0's when the last lane's (i-1)-th has
                        const int me = ctx->pair_id, ot = me == 0 ? ctx->pair->L - 1 : me - 1, need = ctx->pair_seq - (me == 0 ? 1 : 0);
                        if (need >= 0) {
                            while (ctx->pair->rec[ot].load(std::memory_order_acquire) <= need && !ctx->pair->done[ot].load(std::memory_order_acquire)) std::this_thread::yield();
                            if (ctx->pair->rec[ot].load(std::memory_order_acquire) > need) CK(hipStreamWaitEvent(ctx->stream, ctx->pair->ev[ot][need & 7], 0));
                        }
                    }
                    CK(hipEventRecord(ctx->evb[3 * j], ctx->stream));
                    hipLaunchKernelGGL(khit, dim3((unsigned)(nseg * ha.split)), dim3(BLOCK), lds_hit, ctx->stream, ha);
                    CK(hipEventRecord(ctx->evb[3 * j + 1], ctx->stream));
                    if (ctx->pair) { CK(hipEventRecord(ctx->pair->ev[ctx->pair_id][ctx->pair_seq & 7], ctx->stream)); ctx->pair_seq++; ctx->pair->rec[ctx->pair_id].store(ctx->pair_seq, std::memory_order_release); }
                    hipLaunchKernelGGL(kphys, dim3((unsigned)nseg), dim3(BLOCK), lds_phys, ctx->stream, a);
                    if (defer) hipLaunchKernelGGL(kphys2, dim3((unsigned)nseg), dim3(BLOCK), lds_phys, ctx->stream, a);
                    CK(hipEventRecord(ctx->evb[3 * j + 2], ctx->stream));
                    CK(hipGetLastError());
                    a.in_d = q[cur].d; a.in_i = q[cur].i; a.in_m = q[cur].m; a.in_stride = q[cur].cap;
                    cur ^= 1; sin ^= 1;
                }
                unsigned long long nxt = 0, cnt[4];
                CK(hipMemcpyAsync(&nxt, ctx->d_lvl + k0 + nb, 8, hipMemcpyDeviceToHost, ctx->stream));
                CK(hipMemcpyAsync(cnt, ctx->d_counters, 32, hipMemcpyDeviceToHost, ctx->stream));
                CK(hipStreamSynchronize(ctx->stream));
                for (int j = 0; j < nb; j++) {
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, ctx->evb[3 * j], ctx->evb[3 * j + 1])); ctx->ms_kind[1] += ms; ctx->n_kind[1]++; ctx->kernel_ms += ms;
                    CK(hipEventElapsedTime(&ms, ctx->evb[3 * j + 1], ctx->evb[3 * j + 2])); ctx->ms_kind[2] += ms; ctx->n_kind[2]++; ctx->kernel_ms += ms;
                }
                ctx->n_launches += (defer ? 3 : 2) * nb;
                int ovf = (int)(cnt[3] & 0xFFFFFFFFull);
                if (ovf == 3) { ctx->err = "rt_trace: internal error: a ray left a physics step with a direction or polarisation reference that is not a unit vector"; rc = -1; break; }
                if (ovf) { ctx->err = ovf == 1 ? "rt_trace: record buffer overflow (raise workspace or lower nodice_chunk)" : "rt_trace: ray queue overflow (lower nodice_chunk)"; rc = -3; break; }
                live = (int64_t)nxt; k0 += nb;
            }
            if (rc) return rc;
            if (go && live > 0 && k0 < K) { // hand-over: the persistent kernel reads a dense queue
                a.seg_in = seg[sin]; a.out = q[cur]; a.tail = ctx->d_counters + 1; a.step = k0; a.lean_hash = lean_hash_of(k0);
                hipLaunchKernelGGL(k_level_compact, dim3((unsigned)nseg), dim3(BLOCK), 0, ctx->stream, a);
                CK(hipGetLastError());
                CK(hipMemsetAsync(ctx->d_counters + 1, 0, 8, ctx->stream));
                a.in_d = q[cur].d; a.in_i = q[cur].i; a.in_m = q[cur].m; a.in_stride = q[cur].cap;
                cur ^= 1;
            }
            if (rc) return rc;
            a.n_in = live; a.tail = ctx->d_counters + 1;
        }
        // persistent kernel: everything (ray-splitting runs, one launch per generation) or the tail of a level-synchronous run
        for (int gen = 0; gen <= K && live > 0 && k0 < K; gen++) {
            a.out = q[cur];
            int64_t blocks = std::min<int64_t>((a.n_in + BLOCK - 1) / BLOCK, (int64_t)ctx->n_cu * bpc);
            CK(hipEventRecord(ev0, ctx->stream));
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BLOCK), lds, ctx->stream, a);
            CK(hipEventRecord(ev1, ctx->stream));
            CK(hipGetLastError());
            unsigned long long cnt[4];
            CK(hipMemcpyAsync(cnt, ctx->d_counters, 32, hipMemcpyDeviceToHost, ctx->stream));
            CK(hipStreamSynchronize(ctx->stream));
            float ms = 0; CK(hipEventElapsedTime(&ms, ev0, ev1)); ctx->kernel_ms += ms; ctx->n_launches++;
            ctx->ms_kind[0] += ms; ctx->n_kind[0]++;
            int ovf = (int)(cnt[3] & 0xFFFFFFFFull);
            if (ovf == 3) { ctx->err = "rt_trace: internal error: a ray left a physics step with a direction or polarisation reference that is not a unit vector"; rc = -1; break; }
            if (ovf) { ctx->err = ovf == 1 ? "rt_trace: record buffer overflow (raise workspace or lower nodice_chunk)" : "rt_trace: ray queue overflow (lower nodice_chunk)"; rc = -3; break; }
            unsigned long long spawned = cnt[1];
            if (spawned == 0) break;
            // next generation reads what this one appended
            a.mode = 1; a.in_d = q[cur].d; a.in_i = q[cur].i; a.in_m = q[cur].m; a.in_stride = q[cur].cap; a.n_in = (int64_t)spawned;
            cur ^= 1;
            CK(hipMemsetAsync(ctx->d_counters, 0, 16, ctx->stream)); // head, tail (records keep counting)
        }
        if (rc) return rc;
        if (piped) {
            unsigned long long nrec = 0;
            CK(hipMemcpyAsync(&nrec, ctx->d_counters + 2, 8, hipMemcpyDeviceToHost, ctx->stream)); CK(hipStreamSynchronize(ctx->stream));
            const int s = (int)((off / chunk) & 1); // free: its last chunk was drained while the previous one was copied
            if (nrec) {
                hipLaunchKernelGGL(k_rec_pack, dim3((unsigned)((nrec * REC + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)b_rec.p, rec_cap, (int64_t)nrec,
                                   (const int32_t*)b_rstep.p, (const int32_t*)b_rflag.p, (const int64_t*)b_rray.p, (const int64_t*)b_rpar.p, (char*)stg[s].dev.p);
                CK(hipGetLastError());
                CK(hipEventRecord(ctx->ev_ready[s], ctx->stream));
                CK(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_ready[s], 0));
                CK(hipMemcpyAsync(ctx->pin[s], stg[s].dev.p, rec_bytes * (size_t)nrec, hipMemcpyDeviceToHost, ctx->copy_stream));
                CK(hipEventRecord(ctx->ev_copied[s], ctx->copy_stream));
                stg[s].nrec = (int64_t)nrec; stg[s].first = prm->ray_id_base + off * stride; stg[s].m = m; stg[s].pending = true;
            }
            rc = drain(s ^ 1); // the previous chunk goes to the sink while this one is copied; the next one is traced under the same copy
            if (rc) return rc;
        } else if (full) {
            unsigned long long nrec = 0;
            CK(hipMemcpy(&nrec, ctx->d_counters + 2, 8, hipMemcpyDeviceToHost));
            if (nrec) {
                if (ctx->sink) { ctx->rec.clear(); ctx->rec_step.clear(); ctx->rec_flags.clear(); ctx->rec_ray.clear(); ctx->rec_parent.clear(); }
                size_t base = ctx->rec_step.size();
                DevBuf rows; CK(dev_alloc(ctx, rows, (size_t)REC * 8 * nrec));
                hipLaunchKernelGGL(k_rec_to_rows, dim3((unsigned)((nrec * REC + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)b_rec.p, rec_cap, (int64_t)nrec, (double*)rows.p);
                if (!ctx->rec.resize((base + nrec) * REC)) { ctx->err = "rt_trace: out of host memory for the records"; return -1; }
                ctx->rec_step.resize(base + nrec); ctx->rec_flags.resize(base + nrec); ctx->rec_ray.resize(base + nrec); ctx->rec_parent.resize(base + nrec);
                CK(hipMemcpyAsync(ctx->rec.data() + base * REC, rows.p, (size_t)REC * 8 * nrec, hipMemcpyDeviceToHost, ctx->stream));
                CK(hipMemcpyAsync(ctx->rec_step.data() + base, b_rstep.p, 4 * nrec, hipMemcpyDeviceToHost, ctx->stream));
                CK(hipMemcpyAsync(ctx->rec_flags.data() + base, b_rflag.p, 4 * nrec, hipMemcpyDeviceToHost, ctx->stream));
                CK(hipMemcpyAsync(ctx->rec_ray.data() + base, b_rray.p, 8 * nrec, hipMemcpyDeviceToHost, ctx->stream));
                CK(hipMemcpyAsync(ctx->rec_parent.data() + base, b_rpar.p, 8 * nrec, hipMemcpyDeviceToHost, ctx->stream));
                CK(hipStreamSynchronize(ctx->stream));
                for (size_t i = base; i < base + nrec; i++) if (ctx->rec_parent[i] >= 0) ctx->rec_parent[i] += (int64_t)base;
                if (ctx->sink) { // streaming: this chunk's records (self-contained: parents are chunk-local) go to the caller now
                    int src = ctx->sink(ctx->sink_user, (int64_t)nrec, ctx->rec.data(), ctx->rec_step.data(), ctx->rec_ray.data(), ctx->rec_parent.data(), ctx->rec_flags.data(),
                                        prm->ray_id_base + off * stride, m);
                    ctx->rec.clear(); ctx->rec_step.clear(); ctx->rec_flags.clear(); ctx->rec_ray.clear(); ctx->rec_parent.clear();
                    if (src != 0) { ctx->err = "rt_trace: the record sink asked to stop"; return -4; }
                }
            }
        }
        if (wantfin && !fin_dev) CK(hipMemcpy(ctx->fin.data() + (size_t)off * FIN, b_fin.p, (size_t)FIN * 8 * (size_t)m, hipMemcpyDeviceToHost));
        return rc;
    };
    // Passes.  A ray-splitting pass cannot bound its queues and records in advance; one that overflows is traced again in
    // smaller passes from where it began, instead of failing the call and having the caller start over from ray 0: such a
    // pass tallies into a table of its own, which is added to the run's (this context's, or the one the lanes share) when
    // the pass has ended well and dropped when it has not - nothing else of a failed pass has left the device.
    const bool can_retry = !single && !piped;
    DevBuf b_ptab; double* p_table = nullptr; unsigned long long* p_steps = nullptr;
    if (can_retry) { CK(dev_alloc(ctx, b_ptab, telems * 8 + sizeof(unsigned long long) * 2 * (size_t)K)); p_table = (double*)b_ptab.p; p_steps = (unsigned long long*)((char*)b_ptab.p + telems * 8); }
    int shrunk = 0;
    for (int64_t off = 0; off < n && rc == 0;) {
        const int64_t m = std::min<int64_t>(chunk, n - off);
        if (can_retry) {
            CK(hipMemsetAsync(b_ptab.p, 0, telems * 8 + sizeof(unsigned long long) * 2 * (size_t)K, ctx->stream));
            a.table = p_table; a.n_live = p_steps; a.n_scat = p_steps + K;
        }
        rc = do_pass(off, m);
        if (rc == 0 && can_retry) {
            hipLaunchKernelGGL(k_table_add, dim3((unsigned)((telems + 2 * (size_t)K + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, d_table, (const double*)p_table, (int64_t)telems, d_steps, (const unsigned long long*)p_steps, (int64_t)(2 * K));
            CK(hipGetLastError());
        }
        if (rc == -3 && can_retry && shrunk < 8 && m > 256) {
            CK(hipStreamSynchronize(ctx->stream));
            chunk = std::max<int64_t>(256, m / 4); shrunk++; rc = 0; ctx->err.clear(); ctx->n_pass_retries++;
            continue;
        }
        off += m;
    }
    if (can_retry) CK(hipStreamSynchronize(ctx->stream)); // (the pass table goes back to the pool: the last addition has ended)
    if (piped) {
        for (int s = 0; s < 2 && rc == 0; s++) { const int t = (int)(((n + chunk - 1) / chunk + s) & 1); rc = drain(t); } // older stage first
        hipStreamSynchronize(ctx->copy_stream); // nothing of this call is in flight when its buffers go back to the pool
    }
    if (rc) return rc;
    if (ctx->raw_tally) { CK(hipStreamSynchronize(ctx->stream)); return 0; } // a lane of run_trace_lanes: the tallies of all lanes are in one table, fetched when all are done
    return fetch_tallies(ctx, K);
}

// Lanes.  The nearest-hit launch is bound by vector issue (0.72 of all cycles, no memory traffic to speak of), the physics
// launches by HBM traffic and the latency of FP64 dependency chains (half the issue slots idle): run beside each other
// they fill each other's gaps.  So a large tally-only run is traced as `lanes` (2) independent parts at the same time,
// each by a context of its own (stream, queues, counters, workspace; the geometry is shared) on a host thread of its own,
// and the lanes take turns with the nearest-hit launch - lane m's i-th starts when lane m-1's i-th has ended (events
// across the streams) - so that it runs beside the other lane's physics and not beside its twin: 33.5 M rays
// 5.58e9 -> 6.15e9 interactions/s (left to chance, without the turns: 5.90e9); +9 % down to 2 M rays.  Three and four
// lanes are slower than two (each launch then shares the registers of a SIMD with two others).  Rays never interact and
// are keyed by ray id, so the split changes no result.  Tally-only and device-final-row runs only, dice or ray splitting.
static int run_trace_lanes(rt_ctx* ctx, int mode, const double* h_start, const double* h_rays, const double* d_soa, const double* origin, int64_t n, const rt_params* prm) {
    const bool full = prm->flags & RT_FLAG_FULL_OUTPUT, single = prm->flags & RT_FLAG_SINGLECHILD, wantfin = prm->flags & RT_FLAG_FINAL_ROWS;
    for (auto* l : ctx->lane) if (l) l->n_pass_retries = 0;
    const bool eligible = ctx->lanes >= 2 && !ctx->is_lane && (mode == 0 || mode == 2) && !full && // (ray-splitting runs as well: 6.56e9 -> 7.50e9 interactions/s on the arbc_nodice workload)
                          !(prm->follow_threshold[1] < 0) && (!wantfin || ctx->d_fin_user) &&
                          ctx->level_steps > 0 && n >= ctx->lanes_min_rays && !ctx->sink && ctx->d_surf && prm->max_scatters >= 1 && prm->max_scatters <= 4094;
    if (!eligible) return run_trace(ctx, mode, h_start, h_rays, d_soa, origin, n, prm);
    const int L = std::min(ctx->lanes, 4);
    rt_ctx* lc[4] = { ctx, nullptr, nullptr, nullptr };
    for (int j = 1; j < L; j++) { // all of them first: a lane is lent the geometry only once nothing below can fail (rt_free frees what a context points at)
        if (ctx->lane[j - 1]) continue;
        rt_ctx* l = nullptr;
        if (rt_init(ctx->device, &l) != 0) { ctx->err = "rt_trace: cannot create another lane: " + g_init_err; return -1; }
        l->is_lane = true; ctx->lane[j - 1] = l;
    }
    for (int j = 1; j < L; j++) {
        rt_ctx* l = lc[j] = ctx->lane[j - 1];
        // the lane sees the parent's geometry and options as they are now
        l->d_surf = ctx->d_surf; l->d_recs = ctx->d_recs; l->d_smeta = ctx->d_smeta; l->d_prog = ctx->d_prog; l->d_lut = ctx->d_lut; l->d_cam = ctx->d_cam;
        l->nprog_frames = ctx->nprog_frames; l->nprog_generic = ctx->nprog_generic; l->use_frames = ctx->use_frames; l->frames_have_generic = ctx->frames_have_generic;
        l->has_rayleigh = ctx->has_rayleigh; l->has_unified = ctx->has_unified; l->defer_unified = ctx->defer_unified; l->S = ctx->S; l->lut_n = ctx->lut_n;
        l->nodice_chunk = ctx->nodice_chunk; l->refill_min = ctx->refill_min; l->blocks_per_cu = ctx->blocks_per_cu; l->lds_budget = ctx->lds_budget;
        l->level_steps = ctx->level_steps; l->level_min_rays = ctx->level_min_rays; l->hit_blocks_per_cu = ctx->hit_blocks_per_cu; l->phys_blocks_per_cu = ctx->phys_blocks_per_cu;
        l->hit_split = ctx->hit_split; l->hit_split_lanes = ctx->hit_split_lanes; l->seg_per_cu = ctx->seg_per_cu; l->seg_auto = ctx->seg_auto; l->sync_every = ctx->sync_every; l->sync_rays_max = ctx->sync_rays_max;
        l->upload_chunk = ctx->upload_chunk; l->stream_overlap = ctx->stream_overlap;
    }
    const size_t W = ctx->workspace_bytes;
    // whatever way this function is left, the lanes give back what they were lent (rt_free frees what a context points at)
    struct Unlend {
        rt_ctx* c; rt_ctx** lc; int L; size_t W;
        ~Unlend() {
            c->workspace_bytes = W;
            for (int j = 0; j < L; j++) { if (!lc[j]) continue; lc[j]->pair = nullptr; lc[j]->raw_tally = false; lc[j]->d_table_ext = nullptr; lc[j]->d_steps_ext = nullptr; }
            for (int j = 1; j < L; j++) { rt_ctx* l = lc[j]; if (!l) continue; l->d_surf = l->d_recs = l->d_prog = l->d_lut = l->d_cam = nullptr; l->d_smeta = nullptr; l->d_fin_user = nullptr; }
        }
    } unlend{ ctx, lc, L, W };
    const int64_t per = ((n + L - 1) / L + 255) & ~(int64_t)255; // rays of a lane (the last one takes what is left: n >= lanes_min_rays, so it is not empty)
    double* fin_user = ctx->d_fin_user; const int64_t fin_rows = ctx->fin_user_rows;
    rt_params pl[4]; int64_t first[4], cnt[4]; int rcs[4] = { 0, 0, 0, 0 };
    for (int j = 0; j < L; j++) {
        first[j] = per * j; cnt[j] = std::max<int64_t>(0, std::min<int64_t>(per, n - first[j]));
        pl[j] = *prm; pl[j].ray_id_base += first[j] * (prm->ray_id_stride > 0 ? prm->ray_id_stride : 1);
        lc[j]->workspace_bytes = W / L; lc[j]->raw_tally = true;
        if (j > 0) { if (fin_user) { lc[j]->d_fin_user = fin_user + (size_t)first[j] * FIN; lc[j]->fin_user_rows = fin_rows - first[j]; } else { lc[j]->d_fin_user = nullptr; lc[j]->fin_user_rows = 0; } }
    }
    // The nearest-hit launch is bound by vector issue, the physics launches by memory and dependency latency: the lanes
    // take turns with the former (events across the streams, round robin), so that one lane's nearest-hit launch runs
    // beside the others' physics and not beside its twin.
    rt_ctx::LanePair pair; bool paired = false; pair.L = L;
    for (int i = 0; i < 4; i++) { pair.rec[i].store(0); pair.done[i].store(0); }
    if (ctx->lanes_alternate) {
        paired = true;
        for (int i = 0; i < 8 * L && paired; i++) { // (kept in the context: no event is made or destroyed per call after the first)
            hipEvent_t& e = ctx->lane_ev[i / 8][i % 8];
            if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; paired = false; }
            pair.ev[i / 8][i % 8] = e;
        }
    }
    if (paired) for (int j = 0; j < L; j++) { lc[j]->pair = &pair; lc[j]->pair_id = j; lc[j]->pair_seq = 0; }
    // one table and one set of step counters for all lanes (the kernels add with atomics anyway): the tallies of the run are
    // in this context's device table when the lanes are done, as after a run on one lane (rt_result_table_device)
    const int K = prm->max_scatters, S = ctx->S; const size_t row = (size_t)5 * S * 2;
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "rt_trace: hipSetDevice failed"; return -1; }
    if (prepare_tallies(ctx, K)) return -1;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "rt_trace: clearing the tallies failed"; return -1; }
    for (int j = 0; j < L; j++) { lc[j]->d_table_ext = ctx->d_table; lc[j]->d_steps_ext = ctx->d_steps; }
    auto run_lane = [&](int j) {
        if (cnt[j] > 0) rcs[j] = run_trace(lc[j], mode, h_start ? h_start + 3 * first[j] : nullptr, h_rays ? h_rays + 10 * first[j] : nullptr, d_soa ? d_soa + first[j] : nullptr, origin, cnt[j], &pl[j], d_soa ? n : 0);
        pair.done[j].store(1, std::memory_order_release);
    };
    std::vector<std::thread> th;
    for (int j = 1; j < L; j++) th.emplace_back(run_lane, j);
    run_lane(0);
    for (auto& t : th) t.join();
    ctx->workspace_bytes = W; // (fetch_tallies below is this context's own work again; the rest is given back by `unlend`)
    for (int j = 0; j < L; j++) { lc[j]->pair = nullptr; lc[j]->raw_tally = false; lc[j]->d_table_ext = nullptr; lc[j]->d_steps_ext = nullptr; }
    if (rcs[0]) return rcs[0];
    for (int j = 1; j < L; j++) if (rcs[j]) { ctx->err = lc[j]->err; return rcs[j]; }
    (void)row;
    for (int j = 1; j < L; j++) {
        rt_ctx* l = lc[j]; if (cnt[j] <= 0) continue;
        ctx->kernel_ms += l->kernel_ms; ctx->n_launches += l->n_launches; // sums over the lanes: they overlap in time
        for (int i = 0; i < 3; i++) { ctx->ms_kind[i] += l->ms_kind[i]; ctx->n_kind[i] += l->n_kind[i]; }
    }
    ctx->K = K; ctx->n_rays = n;
    return fetch_tallies(ctx, K); // RayTracer2.py:405-407 on the whole
}

} // namespace

extern "C" int rt_trace(rt_ctx* ctx, const double* start, const double* rays, int64_t n, const rt_params* prm) {
    if (!start || !rays) { ctx->err = "Improper input to RayTracer2"; return -1; }
    return run_trace_lanes(ctx, 0, start, rays, nullptr, nullptr, n, prm);
}
extern "C" int rt_trace_device(rt_ctx* ctx, const double* d_soa, int64_t n, const rt_params* prm) {
    if (!d_soa) { ctx->err = "Improper input to RayTracer2"; return -1; }
    return run_trace_lanes(ctx, 0, nullptr, nullptr, d_soa, nullptr, n, prm);
}
extern "C" int rt_trace_point_source(rt_ctx* ctx, const double origin[3], int64_t n, const rt_params* prm) {
    return run_trace_lanes(ctx, 2, nullptr, nullptr, nullptr, origin, n, prm);
}

// ---------------------------------------------------------------------------------------------- no allocation
// rt_trace_device for callers that cannot allocate (include/rt_mi355x.h).  Tally-only single-child runs, traced by
// the level-synchronous kernels for all K steps with fixed grids (the kernels read every count they need from device
// memory, so the host never has to look at one): no hipMalloc / hipFree / event creation, no pooled buffer, no host
// vector - queues, hits, the physics launches' ray lists, segment counts and counters are carved out of this GPU's
// copy of `scratch`.  The array keeps the last call's contents: what must start at zero is cleared here.
namespace {
struct Carver {
    unsigned char* p; size_t used = 0, cap;
    void* take(size_t bytes) { void* r = p + used; used += (bytes + 255) & ~(size_t)255; return r; }
};
} // namespace
extern "C" int rt_trace_device_noalloc(rt_ctx* ctx, const double* d_soa, int64_t n, const rt_params* prm, double* d_table, uint64_t* d_step_counts) {
    if (!ctx->d_surf) { ctx->err = "rt_trace_device_noalloc: no surfaces set"; return -1; }
    if (!d_soa || !d_table || !d_step_counts || n <= 0) { ctx->err = "Improper input to RayTracer2"; return -1; }
    if (n > RT_NOALLOC_MAX_RAYS) { ctx->err = "rt_trace_device_noalloc: at most " + std::to_string((long long)RT_NOALLOC_MAX_RAYS) + " rays per call (the static scratch buffer holds no more); use rt_trace_device"; return -1; }
    const int K = prm->max_scatters, S = ctx->S;
    if (K < 1 || K > RT_NOALLOC_MAX_SCATTERS) { ctx->err = "rt_trace_device_noalloc: max_scatters must be in 1.." + std::to_string(RT_NOALLOC_MAX_SCATTERS); return -1; }
    if (prm->flags != RT_FLAG_SINGLECHILD || prm->follow_threshold[1] < 0) { ctx->err = "rt_trace_device_noalloc: tally-only single-child runs only (flags == RT_FLAG_SINGLECHILD, thresholds >= 0)"; return -1; }
    if (ctx->has_rayleigh && !ctx->d_lut) { ctx->err = "rt_trace: geometry has finite Rayleigh lengths but no azimuth table was set (rt_set_rayleigh_lut)"; return -1; }
    if (ctx->device >= MAX_DEVICES || !g_scratch_ptr[ctx->device]) { ctx->err = "rt_trace_device_noalloc: no scratch buffer on this device"; return -1; }
    CK(hipSetDevice(ctx->device));
    const size_t lds_phys = lds_phys_bytes(S);
    if (lds_phys > 160 * 1024) { ctx->err = "rt_trace: geometry does not fit LDS"; return -1; }

    // ---- segments as in run_trace(): nseg blocks, block b owns queue slots [b * seg_cap, (b + 1) * seg_cap) through all steps
    const int nseg = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->n_cu * ctx->seg_per_cu, (n + 511) / 512));
    const int64_t seg_cap = (n + nseg - 1) / nseg, q_cap = seg_cap * nseg;
    Carver C{ g_scratch_ptr[ctx->device], 0, (size_t)RT_NOALLOC_SCRATCH_BYTES };
    unsigned long long* counters = (unsigned long long*)C.take(256);
    unsigned long long* lvl = (unsigned long long*)C.take(sizeof(unsigned long long) * ((size_t)K + 2));
    uint32_t* seg = (uint32_t*)C.take(sizeof(uint32_t) * 4 * (size_t)nseg);
    const size_t zero_bytes = C.used;                        // counters, live counts, segment counts: must start at zero
    double* hit_l = (double*)C.take(8 * (size_t)q_cap); int32_t* hit_so = (int32_t*)C.take(4 * (size_t)q_cap);
    uint32_t* lists = (uint32_t*)C.take(2 * 4 * (size_t)q_cap);
    Queue q[2];
    for (int j = 0; j < 2; j++) { q[j].cap = q_cap; q[j].d = (double*)C.take(8 * QD * (size_t)q_cap); q[j].i = (int64_t*)C.take(8 * QI * (size_t)q_cap); q[j].m = (int32_t*)C.take(4 * QM * (size_t)q_cap); }
    if (C.used > C.cap) { ctx->err = "rt_trace_device_noalloc: scratch buffer too small for this call"; return -1; } // cannot happen within the limits above
    const size_t telems = (size_t)K * 5 * S * 2;
    CK(hipMemsetAsync(C.p, 0, zero_bytes, ctx->stream));
    CK(hipMemsetAsync(d_table, 0, telems * 8, ctx->stream));
    CK(hipMemsetAsync(d_step_counts, 0, sizeof(uint64_t) * 2 * (size_t)K, ctx->stream));

    KArgs a; memset(&a, 0, sizeof a);
    a.surf = ctx->d_surf; a.S = S; a.recs = ctx->d_recs; a.smeta = ctx->d_smeta; a.lut = ctx->d_lut; a.lut_prec = ctx->lut_n > 0 ? ctx->lut_n - 1 : 1;
    const bool frames = ctx->use_frames && !(prm->compat & RT_COMPAT_PY_ROOT2);
    if (frames) { a.prog = ctx->d_prog; a.nprog = ctx->nprog_frames; } else { a.prog = ctx->d_prog + (size_t)(ctx->nprog_frames + 1) * PREC; a.nprog = ctx->nprog_generic; }
    a.P.K = K; a.P.flags = prm->flags; a.P.compat = prm->compat; a.P.maxiter = prm->unified_maxiter > 0 ? prm->unified_maxiter : 10000;
    a.P.base = (prm->compat & RT_COMPAT_PY_INDEX) ? 0 : 1; a.P.min_travel = prm->min_travel_length; a.P.thrA = prm->follow_threshold[0]; a.P.thrB = prm->follow_threshold[1];
    a.P.tir = prm->tir_handling; a.P.key0 = (uint32_t)prm->seed; a.P.key1 = (uint32_t)(prm->seed >> 32);
    a.table = d_table; a.n_live = (unsigned long long*)d_step_counts; a.n_scat = (unsigned long long*)d_step_counts + K;
    a.head = counters; a.tail = counters + 1; a.rec_count = counters + 2; a.overflow = (int*)(counters + 3); a.prof = counters + 8;
    a.hit_l = hit_l; a.hit_so = hit_so; a.rows_lds = 1; a.refill_min = ctx->refill_min; a.lean = 1;
    const bool defer = ctx->has_unified && ctx->defer_unified;
    void (*kphys)(const KArgs) = defer ? k_level_phys<FLAG_SINGLE, 1> : (ctx->has_unified ? k_level_phys<FLAG_SINGLE, 0> : k_level_phys<FLAG_SINGLE, 3>);
    void (*kphys2)(const KArgs) = k_level_phys<FLAG_SINGLE, 2>;
    a.defer = lists; a.dense = lists + (size_t)q_cap; a.defer_n = seg + 2 * (size_t)nseg;
    void (*khit)(const HitArgs) = (!frames || ctx->frames_have_generic) ? k_level_hit<true> : k_level_hit<false>;
    CK(hipFuncSetAttribute((const void*)kphys, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_phys));
    if (defer) CK(hipFuncSetAttribute((const void*)kphys2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_phys));
    HitArgs ha; memset(&ha, 0, sizeof ha);
    ha.prog = a.prog; ha.nprog = a.nprog; ha.recs = ctx->d_recs; ha.surf = ctx->d_surf; ha.smeta = ctx->d_smeta; ha.S = S;
    ha.min_travel = a.P.min_travel; ha.base = a.P.base; ha.compat = a.P.compat; ha.hit_l = hit_l; ha.hit_so = hit_so; ha.lean = 1;
    ha.seg_cap = seg_cap; ha.split = (int)((seg_cap + BLOCK - 1) / BLOCK); // one 256-ray tile per block; tiles behind a segment's live count have nothing to do
    a.seg_cap = seg_cap; a.seg_fill = seg_cap;
    uint32_t* sg[2] = { seg, seg + nseg };

    // initial rays -> queue 1 (step 0 writes queue 0)
    a.mode = 0; a.in_d = d_soa; a.in_stride = n; a.n_in = n; a.ray_id_base = prm->ray_id_base; a.slot_base = prm->ray_id_base; a.id_stride = prm->ray_id_stride > 0 ? prm->ray_id_stride : 1;
    a.out = q[1]; a.seg_out = sg[1];
    hipLaunchKernelGGL(k_level_init<0>, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, a);
    a.mode = 1; a.in_d = q[1].d; a.in_i = q[1].i; a.in_m = q[1].m; a.in_stride = q_cap;
    int cur = 0, sin = 1;
    for (int k = 0; k < K; k++) {
        a.step = k; a.n_in = n; a.out = q[cur]; a.tail = lvl + k + 1; a.seg_in = sg[sin]; a.seg_out = sg[sin ^ 1]; a.lean_hash = lean_hash_of(k);
        ha.in_d = a.in_d; ha.in_m = a.in_m; ha.in_stride = q_cap; ha.n_in = n; ha.seg_in = sg[sin];
        hipLaunchKernelGGL(khit, dim3((unsigned)(nseg * ha.split)), dim3(BLOCK), (size_t)TQ_DOUBLES * 8, ctx->stream, ha);
        hipLaunchKernelGGL(kphys, dim3((unsigned)nseg), dim3(BLOCK), lds_phys, ctx->stream, a);
        if (defer) hipLaunchKernelGGL(kphys2, dim3((unsigned)nseg), dim3(BLOCK), lds_phys, ctx->stream, a);
        a.in_d = q[cur].d; a.in_i = q[cur].i; a.in_m = q[cur].m;
        cur ^= 1; sin ^= 1;
    }
    hipLaunchKernelGGL(k_level_finish, dim3((unsigned)K), dim3(BLOCK), 0, ctx->stream, d_table, (const unsigned long long*)d_step_counts + K, 5 * S * 2);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(ctx->stream));
    return 0;
}
namespace {
int upload_camera(rt_ctx* ctx, const rt_camera* cam, int64_t* n_pix) {
    if (!cam || cam->resolution[0] < 1 || cam->resolution[1] < 1 || cam->n_distortion < 0 || cam->n_distortion > 8 || cam->lens_type < 0 || cam->lens_type > 2 ||
        !(cam->focal_length != 0)) { ctx->err = "impropper input to GenerateRaysFromCamera"; return -1; } // GenerateRaysFromCamera.m:35 (sic)
    double c[CAM_DOUBLES]; for (int i = 0; i < CAM_DOUBLES; i++) c[i] = 0;
    c[0] = cam->resolution[0]; c[1] = cam->resolution[1]; c[2] = cam->lens_type; c[3] = cam->n_distortion;
    c[4] = cam->pixel_pitch[0]; c[5] = cam->pixel_pitch[1]; c[6] = cam->pixel_center[0]; c[7] = cam->pixel_center[1]; c[8] = cam->focal_length;
    for (int k = 0; k < cam->n_distortion; k++) c[9 + k] = cam->distortion[k];
    // M = M1 M2 M3, GenerateRaysFromCamera.m:74-78
    const double cy = cos(cam->yaw), sy = sin(cam->yaw), cp = cos(cam->pitch), sp = sin(cam->pitch), cr = cos(cam->roll), sr = sin(cam->roll);
    const double M1[9] = { cy, -sy, 0, sy, cy, 0, 0, 0, 1 }, M2[9] = { 1, 0, 0, 0, cp, -sp, 0, sp, cp }, M3[9] = { cr, 0, sr, 0, 1, 0, -sr, 0, cr };
    double T[9], M[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T[3 * i + j] = 0; for (int k = 0; k < 3; k++) T[3 * i + j] += M1[3 * i + k] * M2[3 * k + j]; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { M[3 * i + j] = 0; for (int k = 0; k < 3; k++) M[3 * i + j] += T[3 * i + k] * M3[3 * k + j]; }
    for (int i = 0; i < 9; i++) c[17 + i] = M[i];
    for (int i = 0; i < 3; i++) { c[26 + i] = cam->position[i]; c[29 + i] = cam->polref[i]; }
    for (int i = 0; i < 4; i++) c[32 + i] = cam->stokes[i];
    CK(hipSetDevice(ctx->device));
    if (!ctx->d_cam) CK(hipMalloc(&ctx->d_cam, sizeof c));
    CK(hipMemcpy(ctx->d_cam, c, sizeof c, hipMemcpyHostToDevice));
    *n_pix = (int64_t)cam->resolution[0] * cam->resolution[1];
    return 0;
}
} // namespace
extern "C" int rt_trace_camera(rt_ctx* ctx, const rt_camera* cam, const rt_params* prm) {
    int64_t n = 0;
    if (upload_camera(ctx, cam, &n)) return -1;
    if (prm->ray_id_stride > 1) { ctx->err = "rt_trace_camera: a stride needs a pixel range (rt_trace_camera_range)"; return -1; }
    ctx->cam_first = 0;
    return run_trace(ctx, 3, nullptr, nullptr, nullptr, nullptr, n, prm);
}
extern "C" int rt_trace_camera_range(rt_ctx* ctx, const rt_camera* cam, int64_t first, int64_t n, const rt_params* prm) {
    int64_t npix = 0;
    if (upload_camera(ctx, cam, &npix)) return -1;
    const int64_t stride = prm->ray_id_stride > 0 ? prm->ray_id_stride : 1;
    if (first < 0 || n < 1 || first + (n - 1) * stride >= npix) { ctx->err = "rt_trace_camera_range: pixel range outside the sensor"; return -1; }
    ctx->cam_first = first;
    const int rc = run_trace(ctx, 3, nullptr, nullptr, nullptr, nullptr, n, prm);
    ctx->cam_first = 0;
    return rc;
}
extern "C" int rt_generate_camera_rays(rt_ctx* ctx, const rt_camera* cam, int64_t first, int64_t n, double* start, double* rays, int32_t* pixels) {
    int64_t npix = 0;
    if (upload_camera(ctx, cam, &npix)) return -1;
    if (first < 0 || n < 1 || first + n > npix) { ctx->err = "rt_generate_camera_rays: pixel range outside the sensor"; return -1; }
    DevBuf s, r, px; CK(dev_alloc(ctx, s, 24 * (size_t)n)); CK(dev_alloc(ctx, r, 80 * (size_t)n)); CK(dev_alloc(ctx, px, 8 * (size_t)n));
    hipLaunchKernelGGL(k_gen_camera, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)ctx->d_cam, first, n, (double*)s.p, (double*)r.p, (int32_t*)px.p);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(start, s.p, 24 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipMemcpyAsync(rays, r.p, 80 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipMemcpyAsync(pixels, px.p, 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return 0;
}
extern "C" int rt_generate_point_source(rt_ctx* ctx, const double origin[3], int64_t n, uint64_t seed, int64_t ray_id_base, double* start, double* rays) {
    CK(hipSetDevice(ctx->device));
    DevBuf s, r; CK(dev_alloc(ctx, s, 24 * (size_t)n)); CK(dev_alloc(ctx, r, 80 * (size_t)n));
    KArgs a; memset(&a, 0, sizeof a);
    a.P.key0 = (uint32_t)seed; a.P.key1 = (uint32_t)(seed >> 32); a.ray_id_base = ray_id_base; a.id_stride = 1; a.ox = origin[0]; a.oy = origin[1]; a.oz = origin[2];
    hipLaunchKernelGGL(k_gen_point, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, n, (double*)s.p, (double*)r.p);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(start, s.p, 24 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipMemcpyAsync(rays, r.p, 80 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// diagnostic builds (-DRT_PROFILE) only, 16 counters: cycle sums {refill, find_hit, physics, tally+output}, solver calls, wave passes,
// live lanes, lanes wanting a torus solve, candidate-block cycles, candidate blocks
extern "C" int rt_debug_profile(rt_ctx* ctx, unsigned long long* out8, int reset) {
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    if (out8 && hipMemcpy(out8, ctx->d_counters + 8, 128, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (reset && hipMemset(ctx->d_counters + 8, 0, 128) != hipSuccess) return -1;
    return 0;
}
extern "C" int rt_result_num_steps(rt_ctx* ctx, int32_t* num_steps) { *num_steps = ctx->num_steps; return 0; }
extern "C" int rt_result_pass_retries(rt_ctx* ctx, int32_t* n) { int t = ctx->n_pass_retries; for (auto* l : ctx->lane) if (l) t += l->n_pass_retries; *n = t; return 0; }
extern "C" int rt_result_table(rt_ctx* ctx, double* table) { memcpy(table, ctx->table.data(), sizeof(double) * (size_t)ctx->num_steps * 5 * ctx->S * 2); return 0; }
extern "C" int rt_result_table_device(rt_ctx* ctx, double* d_table, uint64_t* d_step_counts) {
    if (!d_table || ctx->K < 1 || !ctx->d_table) { ctx->err = "rt_result_table_device: no trace to report, or no destination"; return -1; }
    const int K = ctx->K; const size_t telems = (size_t)K * 5 * ctx->S * 2;
    CK(hipSetDevice(ctx->device));
    CK(hipMemcpyAsync(d_table, ctx->d_table, telems * 8, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_level_finish, dim3((unsigned)K), dim3(BLOCK), 0, ctx->stream, d_table, (const unsigned long long*)ctx->d_steps + K, 5 * ctx->S * 2); // RayTracer2.py:405-407
    CK(hipGetLastError());
    if (d_step_counts) CK(hipMemcpyAsync(d_step_counts, ctx->d_steps, sizeof(uint64_t) * 2 * (size_t)K, hipMemcpyDeviceToDevice, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return 0;
}
extern "C" int rt_result_step_counts(rt_ctx* ctx, int64_t* n_live, int64_t* n_scat) {
    memcpy(n_live, ctx->n_live.data(), 8 * (size_t)ctx->K); memcpy(n_scat, ctx->n_scat.data(), 8 * (size_t)ctx->K); return 0;
}
// ---- host work of the drop-in call, off the interpreter (python/RayTracer2.py:221-222 and :531-540 for 10^7 rows)
namespace {
template <class F> void parallel_ranges(int64_t n, int64_t grain, F f) { // f(thread, first, last) over [0, n) on up to 16 host threads
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())), n / std::max<int64_t>(grain, 1)));
    if (nt <= 1) { f(0, (int64_t)0, n); return; }
    std::vector<std::thread> th; th.reserve((size_t)nt); const int64_t per = (n + nt - 1) / nt;
    auto lo = [&](int t) { return std::min<int64_t>(n, t * per); };
    int started = 0;
    for (int t = 1; t < nt; t++) { try { th.emplace_back(f, t, lo(t), lo(t + 1)); started = t; } catch (...) { break; } } // (a thread that cannot be started: its range is done here)
    f(0, lo(0), lo(1));
    for (int t = started + 1; t < nt; t++) f(t, lo(t), lo(t + 1));
    for (auto& t : th) t.join();
}
int parallel_threads(int64_t n, int64_t grain) { return (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())), n / std::max<int64_t>(grain, 1))); }
// entries >= 0 of tab[0..m) in index order appended to out; returns how many.  (two passes: counts per thread, then the copy)
int64_t compact_table(const int64_t* tab, int64_t m, int64_t* out) {
    const int nt = parallel_threads(m, 1 << 16);
    std::vector<int64_t> cnt((size_t)nt + 1, 0);
    parallel_ranges(m, 1 << 16, [&](int t, int64_t a, int64_t b) { int64_t c = 0; for (int64_t i = a; i < b; i++) c += tab[i] >= 0; cnt[(size_t)t + 1] = c; });
    for (int t = 0; t < nt; t++) cnt[(size_t)t + 1] += cnt[(size_t)t];
    parallel_ranges(m, 1 << 16, [&](int t, int64_t a, int64_t b) { int64_t o = cnt[(size_t)t]; for (int64_t i = a; i < b; i++) if (tab[i] >= 0) out[o++] = tab[i]; });
    return cnt[(size_t)nt];
}
} // namespace
// Records order[0..n) of a row table split into fields: field f = columns [col0[f], col0[f] + ncols[f]) as a dense
// [n][ncols[f]] array at outs[f].  One pass over the rows, on several host threads for large n.  Returns 0, -1 (bad
// column range) or -2 (record index out of range).
extern "C" int rt_split_record_fields(const double* rows, int64_t n_rows, const int64_t* order, int64_t n, int32_t nfields, const int32_t* col0, const int32_t* ncols,
                                      double* const* outs) {
    for (int f = 0; f < nfields; f++) if (col0[f] < 0 || ncols[f] < 1 || col0[f] + ncols[f] > REC) return -1;
    std::atomic<int> bad{ 0 };
    auto work = [&](int64_t j0, int64_t j1) {
        for (int64_t j = j0; j < j1; j++) {
            const int64_t i = order ? order[j] : j;
            if (i < 0 || i >= n_rows) { bad = 1; continue; }
            const double* row = rows + (size_t)i * REC;
            for (int f = 0; f < nfields; f++) { double* o = outs[f] + (size_t)j * ncols[f]; const double* r = row + col0[f]; for (int c = 0; c < ncols[f]; c++) o[c] = r[c]; }
        }
    };
    parallel_ranges(n, 65536, [&](int, int64_t a, int64_t b) { work(a, b); });
    return bad ? -2 : 0;
}
// The caller's rays normalised in place as the reference does it to the caller's array (python/RayTracer2.py:221-222:
// columns 0-2 and 3-5 divided by |sqrt(sum of squares)|, the same operations in the same order as the numpy expression,
// so the same bits), on several host threads.
extern "C" int rt_normalize_rays(double* rays, int64_t n, double* max_s0) try {
    if (!rays || n < 0) return -1;
    const int nt = parallel_threads(n, 1 << 15);
    std::vector<double> mx((size_t)nt, -INFINITY); std::vector<char> nan((size_t)nt, 0);
    parallel_ranges(n, 1 << 15, [&](int t, int64_t a, int64_t b) {
        for (int64_t i = a; i < b; i++) {
            double* r = rays + 10 * i;
            if (r[6] != r[6]) nan[(size_t)t] = 1; else if (r[6] > mx[(size_t)t]) mx[(size_t)t] = r[6];
            for (int c = 0; c < 6; c += 3) {
#pragma clang fp contract(off)
                const double s = fabs(sqrt((r[c] * r[c] + r[c + 1] * r[c + 1]) + r[c + 2] * r[c + 2]));
                r[c] = r[c] / s; r[c + 1] = r[c + 1] / s; r[c + 2] = r[c + 2] / s;
            }
        }
    });
    if (max_s0) { double m = -INFINITY; bool any_nan = false; for (int t = 0; t < nt; t++) { m = std::max(m, mx[(size_t)t]); any_nan |= nan[(size_t)t] != 0; } *max_s0 = any_nan ? NAN : m; } // numpy.max: NaN if any
    return 0;
} catch (...) { return -1; } // (only the two small vectors above can throw, before anything is written; the caller then normalises with numpy)
// Row order of the reference for the records of a trace (python/RayTracer2.py:531-540: step 0 in input order; step k + 1 =
// [refracted children in the order of their parents ; reflected children in the order of their parents]) from the ordering
// metadata of rt_result_records.  order[0..n): record indices, step after step; counts[k]: records of step k.  Tables
// instead of sorts: a record of step k + 1 is the only child of its kind of its parent, so its place follows from the
// parent's place by one scatter and one compaction - O(n) on several threads where numpy's lexsort per step took longer than
// the trace.  Returns 0, or -2 if the metadata is not of that form (the caller then sorts: RayTracer2.order_records).
extern "C" int rt_order_records(const int32_t* step, const int64_t* ray_id, const int64_t* parent, const int32_t* flags, int64_t n, int32_t num_steps,
                                int64_t* order, int64_t* counts) try {
    if (n < 0 || num_steps < 0 || (n > 0 && (!step || !ray_id || !parent || !flags || !order)) || (num_steps > 0 && !counts)) return -1;
    for (int k = 0; k < num_steps; k++) counts[k] = 0;
    if (n == 0 || num_steps == 0) return 0;
    // records of every step, in record order
    std::atomic<int> bad{ 0 };
    const int nt = parallel_threads(n, 1 << 16);
    std::vector<int64_t> cnt((size_t)nt * num_steps, 0); // per thread and step
    parallel_ranges(n, 1 << 16, [&](int t, int64_t a, int64_t b) {
        int64_t* c = cnt.data() + (size_t)t * num_steps;
        for (int64_t r = a; r < b; r++) { if (step[r] < 0) { bad = 1; continue; } if (step[r] < num_steps) c[step[r]]++; }
    });
    if (bad) return -2;
    std::vector<int64_t> start((size_t)num_steps + 1, 0);
    { int64_t at_ = 0; for (int k = 0; k < num_steps; k++) { start[(size_t)k] = at_; for (int t = 0; t < nt; t++) { const int64_t c = cnt[(size_t)t * num_steps + k]; cnt[(size_t)t * num_steps + k] = at_; at_ += c; } } start[(size_t)num_steps] = at_; }
    std::unique_ptr<int64_t[]> by_step_buf(new int64_t[(size_t)start[(size_t)num_steps] + 1]); int64_t* by_step = by_step_buf.get();
    parallel_ranges(n, 1 << 16, [&](int t, int64_t a, int64_t b) {
        int64_t* c = cnt.data() + (size_t)t * num_steps;
        for (int64_t r = a; r < b; r++) if (step[r] < num_steps) by_step[c[step[r]]++] = r;
    });
    std::unique_ptr<int64_t[]> pos_buf(new int64_t[(size_t)n]); int64_t* pos = pos_buf.get(); // place of a record within its step (written before it is read: every parent is placed a step earlier)
    parallel_ranges(n, 1 << 16, [&](int, int64_t a, int64_t b) { for (int64_t r = a; r < b; r++) pos[r] = -1; });
    int64_t at = 0;
    // step 0: by ray id (one record per ray): a table over the id range
    {
        const int64_t m = start[1];
        if (m > 0) {
            int64_t lo = ray_id[by_step[0]], hi = lo;
            { const int nt0 = parallel_threads(m, 1 << 16); std::vector<int64_t> los((size_t)nt0, lo), his((size_t)nt0, hi);
              parallel_ranges(m, 1 << 16, [&](int t, int64_t a, int64_t b) { int64_t l_ = lo, h_ = hi; for (int64_t j = a; j < b; j++) { const int64_t v = ray_id[by_step[j]]; l_ = std::min(l_, v); h_ = std::max(h_, v); } los[(size_t)t] = l_; his[(size_t)t] = h_; });
              for (int t = 0; t < nt0; t++) { lo = std::min(lo, los[(size_t)t]); hi = std::max(hi, his[(size_t)t]); } }
            const int64_t span = hi - lo + 1;
            if (span > 64 * m + (1 << 20)) return -2; // ids too sparse for a table
            std::unique_ptr<int64_t[]> tab_buf(new int64_t[(size_t)span]); int64_t* tab_ = tab_buf.get();
            parallel_ranges(span, 1 << 16, [&](int, int64_t a, int64_t b) { for (int64_t i = a; i < b; i++) tab_[i] = -1; });
            struct { int64_t* p; int64_t& operator[](size_t i) { return p[i]; } int64_t* data() { return p; } } tab{ tab_ };
            parallel_ranges(m, 1 << 16, [&](int, int64_t a, int64_t b) { for (int64_t j = a; j < b; j++) { const int64_t r = by_step[j]; int64_t& e = tab[(size_t)(ray_id[r] - lo)]; if (e >= 0) bad = 1; e = r; } });
            if (bad) return -2; // two records of one ray in step 0
            const int64_t got = compact_table(tab.data(), span, order);
            if (got != m) return -2;
            parallel_ranges(m, 1 << 16, [&](int, int64_t a, int64_t b) { for (int64_t j = a; j < b; j++) pos[order[j]] = j; });
        }
        counts[0] = m; at = m;
    }
    for (int k = 1; k < num_steps; k++) {
        const int64_t m = start[(size_t)k + 1] - start[(size_t)k], mp = counts[k - 1];
        const int64_t* rec = by_step + start[(size_t)k];
        if (m > 0) {
            if (mp <= 0) return -2;
            std::unique_ptr<int64_t[]> tab_buf(new int64_t[(size_t)(2 * mp)]); int64_t* tab_ = tab_buf.get(); // [born refracted: by the parent's place | born reflected: by the parent's place]
            parallel_ranges(2 * mp, 1 << 16, [&](int, int64_t a, int64_t b) { for (int64_t i = a; i < b; i++) tab_[i] = -1; });
            struct { int64_t* p; int64_t& operator[](size_t i) { return p[i]; } int64_t* data() { return p; } } tab{ tab_ };
            parallel_ranges(m, 1 << 16, [&](int, int64_t a, int64_t b) {
                for (int64_t j = a; j < b; j++) {
                    const int64_t r = rec[j], p = parent[r];
                    if (p < 0 || p >= n || step[p] != k - 1 || pos[p] < 0) { bad = 1; continue; }
                    int64_t& e = tab[(size_t)(((flags[r] & 4) ? mp : 0) + pos[p])];
                    if (e >= 0) bad = 1; // a second child of the same kind: not the reference's tree
                    e = r;
                }
            });
            if (bad) return -2;
            if (compact_table(tab.data(), 2 * mp, order + at) != m) return -2;
            int64_t* o = order + at;
            parallel_ranges(m, 1 << 16, [&](int, int64_t a, int64_t b) { for (int64_t j = a; j < b; j++) pos[o[j]] = j; });
        }
        counts[k] = m; at += m;
    }
    return 0;
} catch (...) { return -1; } // (no memory for the tables, no thread: the caller sorts instead)
extern "C" int rt_result_record_fields(rt_ctx* ctx, const int64_t* order, int64_t n, int32_t nfields, const int32_t* col0, const int32_t* ncols, double* const* outs) {
    const int rc = rt_split_record_fields(ctx->rec.data(), (int64_t)ctx->rec_step.size(), order, n, nfields, col0, ncols, outs);
    if (rc == -1) ctx->err = "rt_result_record_fields: bad column range";
    if (rc == -2) ctx->err = "rt_result_record_fields: record index out of range";
    return rc ? -1 : 0;
}
extern "C" int rt_result_num_records(rt_ctx* ctx, int64_t* n) { *n = (int64_t)ctx->rec_step.size(); return 0; }
extern "C" int rt_result_records(rt_ctx* ctx, double* rec, int32_t* step, int64_t* ray_id, int64_t* parent, int32_t* flags) {
    size_t n = ctx->rec_step.size();
    if (rec) memcpy(rec, ctx->rec.data(), n * REC * 8);
    if (step) memcpy(step, ctx->rec_step.data(), n * 4);
    if (ray_id) memcpy(ray_id, ctx->rec_ray.data(), n * 8);
    if (parent) memcpy(parent, ctx->rec_parent.data(), n * 8);
    if (flags) memcpy(flags, ctx->rec_flags.data(), n * 4);
    return 0;
}
extern "C" int rt_result_final(rt_ctx* ctx, double* rows) {
    if (ctx->fin.empty()) { ctx->err = "rt_result_final: trace was run without RT_FLAG_FINAL_ROWS"; return -1; }
    memcpy(rows, ctx->fin.data(), ctx->fin.size() * 8); return 0;
}
extern "C" int rt_result_timing_by_kernel(rt_ctx* ctx, double ms[3], int32_t launches[3]) {
    for (int i = 0; i < 3; i++) { if (ms) ms[i] = ctx->ms_kind[i]; if (launches) launches[i] = ctx->n_kind[i]; }
    return 0;
}
extern "C" int rt_result_timing(rt_ctx* ctx, double* kernel_ms, int32_t* n_launches) { if (kernel_ms) *kernel_ms = ctx->kernel_ms; if (n_launches) *n_launches = ctx->n_launches; return 0; }

#include "rt_match.hpp"
